"""Where does the data-parallel step spend its extra time at world size 1?  (torchrun --nproc-per-node 1 tools/dp_overhead.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from coati_amd.engine import Engine, ModelConfig
from coati_amd.synthetic import make_batch
from coati_amd import distributed as D
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
DEV = "cuda:0"
GRANDE = dict(n_layer_e3gnn=5, n_layer_xformer=16, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256,
              n_head=16, n_seq=250, n_tok=10322)
eng = Engine(ModelConfig(**GRANDE), DEV)
g = torch.Generator(device="cpu").manual_seed(0)
with torch.no_grad():
    for name, (off, shape) in eng.layout.items():
        v = eng.view(name)
        if len(shape) == 2:
            v.copy_((torch.randn(shape, generator=g) * (0.02 if "tok_emb" not in name else 1.0)).to(DEV))
        elif ".ln_" in name and name.endswith("weight") or name.endswith("clip.0.weight"):
            v.fill_(1.0)
        else:
            v.zero_()
eng.refresh_shadows()
batch, up = make_batch(1024, 80, 16, GRANDE["n_tok"], seed=1)
batch = {k: v.to(DEV) for k, v in batch.items()}; up = up.to(DEV)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
plain = timeit(lambda: eng.train_step(batch, up, lr=5e-4))
dp = timeit(lambda: D.distributed_train_step(eng, batch, up, lr=5e-4))
real_ar = dist.all_reduce
class _W:
    def wait(self): pass
dist.all_reduce = lambda *a, **k: _W()
dp_noar = timeit(lambda: D.distributed_train_step(eng, batch, up, lr=5e-4))
dist.all_reduce = real_ar
print(f"plain {plain:.3f} ms  dp {dp:.3f} ms  dp without the 4 all-reduces {dp_noar:.3f} ms")
dist.destroy_process_group()
