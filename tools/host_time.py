"""Host enqueue time of one training step against its GPU time: does the host stay ahead of the device?
    python tools/host_time.py            (one GPU, grande_closed B = 1024, packed rows)
Per step: wall time of the Python call Engine.train_step() right after a device synchronise (nothing queued: the call cannot block
on a full queue) = the host's cost of enqueueing the step; then the synchronise that follows = what is left of the GPU's work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from coati_amd.engine import Engine, ModelConfig
from coati_amd.synthetic import make_batch

dev = torch.device("cuda:0")
eng = Engine(ModelConfig(**bench.GRANDE), dev)
g = torch.Generator(device="cpu").manual_seed(0)
with torch.no_grad():
    for name, (off, shape) in eng.layout.items():
        v = eng.view(name)
        if len(shape) == 2:
            v.copy_((torch.randn(shape, generator=g) * (0.02 if "tok_emb" not in name else 1.0)).to(dev))
        elif ".ln_" in name and name.endswith("weight") or name.endswith("clip.0.weight"):
            v.fill_(1.0)
        else:
            v.zero_()
eng.refresh_shadows()
batch_cpu, up_cpu = make_batch(1024, 80, 16, bench.GRANDE["n_tok"], seed=1234, with_rows=True)
batch = {k: (v if k == "rows" else v.to(dev)) for k, v in batch_cpu.items()}
up = up_cpu.to(dev)
for _ in range(5):
    eng.train_step(batch, up, lr=5e-4)
torch.cuda.synchronize()
host, total = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train_step(batch, up, lr=5e-4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"host enqueue of one step: median {host[10]:.2f} ms (min {host[0]:.2f}, max {host[-1]:.2f}); step end to end from an idle device: median {total[10]:.2f} ms")
t0 = time.perf_counter()
for _ in range(50):
    eng.train_step(batch, up, lr=5e-4)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"50 steps back to back: host loop {1e3 * (t1 - t0) / 50:.2f} ms/step, with the final synchronise {1e3 * (t2 - t0) / 50:.2f} ms/step")
