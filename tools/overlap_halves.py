"""Experiment: does running two HALF batches concurrently (two engines, two streams) beat one full batch?  If kernels of
different character (latency-bound attention, HBM-bound GEMM epilogues, MFMA phases) overlap across the two streams, the pair
finishes sooner than one B = 1024 step.   python tools/overlap_halves.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coati_amd.engine import Engine, ModelConfig
from coati_amd.synthetic import make_batch
dev = torch.device("cuda:0")
GRANDE = dict(n_layer_e3gnn=5, n_layer_xformer=16, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256, n_head=16, n_seq=250, n_tok=10322)

def mk(B, seed):
    eng = Engine(ModelConfig(**GRANDE), dev)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for name, (off, shape) in eng.layout.items():
            v = eng.view(name)
            if len(shape) == 2: v.copy_((torch.randn(shape, generator=g) * (0.02 if "tok_emb" not in name else 1.0)).to(dev))
            elif ".ln_" in name and name.endswith("weight") or name.endswith("clip.0.weight"): v.fill_(1.0)
            else: v.zero_()
    eng.refresh_shadows()
    b, up = make_batch(B, 80, 16, GRANDE["n_tok"], seed=seed, with_rows=True)
    db = {k: (v if k == "rows" else v.to(dev)) for k, v in b.items()}
    return eng, db, up.to(dev)

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

e1, b1, u1 = mk(1024, 1234)
t_full = timed(lambda: e1.train_step(b1, u1, lr=5e-4))
print(f"one engine, B = 1024: {t_full:.2f} ms/step")
del e1
ea, ba, ua = mk(512, 1)
eb, bb, ub = mk(512, 2)
t_half = timed(lambda: ea.train_step(ba, ua, lr=5e-4))
print(f"one engine, B = 512: {t_half:.2f} ms/step (x2 = {2 * t_half:.2f})")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def pair():
    with torch.cuda.stream(sa): ea.train_step(ba, ua, lr=5e-4)
    with torch.cuda.stream(sb): eb.train_step(bb, ub, lr=5e-4)
t_pair = timed(pair)
print(f"two engines, B = 512 each, two streams concurrently: {t_pair:.2f} ms per pair (= 1024 molecules)")
