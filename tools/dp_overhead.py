"""Where does the data-parallel step's overhead at world size 1 (RCCL, nothing on the wire) come from?
    COATI_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29633 tools/dp_overhead.py
V0 plain Engine.train_step | V1 staged backward, no exchange step, no gradient collectives | V2 + exchange step | V3 + gradient collectives"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from coati_amd.engine import Engine, ModelConfig
from coati_amd.synthetic import make_batch
from coati_amd import distributed as D

dist.init_process_group("nccl", **({"device_id": torch.device("cuda:0")} if os.environ.get("DP_DEVICE_ID") else {}))
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
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
os.environ.setdefault("COATI_DP_SPLIT", "0")
variants = {
    "V0 plain train_step": lambda: eng.train_step(batch, up, lr=5e-4),
    "V1 staged, no exchange, no grad collectives": lambda: D.distributed_train_step(eng, batch, up, lr=5e-4, do_clip=False, reduce_grads=False),
    "V2 + exchange step": lambda: D.distributed_train_step(eng, batch, up, lr=5e-4, reduce_grads=False),
    "V3 + gradient collectives": lambda: D.distributed_train_step(eng, batch, up, lr=5e-4),
    "V0b plain, InfoNCE off": lambda: eng.train_step(batch, up, lr=5e-4, do_clip=False) if "do_clip" in eng.train_step.__code__.co_varnames else eng.train_step(batch, up, lr=5e-4),
}
for rep in range(2):
    for name, fn in variants.items():
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        print(f"{name:48s} {1e3 * (time.perf_counter() - t0) / 30:7.3f} ms/step", flush=True)
# the same with the bench's per-site events switched on (prof_select): bench.py times its steps that way
eng.prof_select("fc1_dgrad,qkv_dgrad,lmhead_dgrad", keep_overlap=True)
for name in ("V0 plain train_step", "V3 + gradient collectives"):
    fn = variants[name]
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name + ' [site events on]':48s} {1e3 * (time.perf_counter() - t0) / 20:7.3f} ms/step", flush=True)
    eng.prof_collect()
eng.prof_select(-1)
dist.destroy_process_group()
