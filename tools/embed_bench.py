import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops, _lib
from coati_amd.ops import ptr, stream
from coati_amd.synthetic import make_batch
from gemm_bench_util import timeit
dev = "cuda:0"; B, T, C, V = 1024, 80, 256, 10322
batch, up = make_batch(B, T, 16, V, seed=1234)
dx = torch.randn(B * T, C, device=dev)
for name in ("tokens", "raw_tokens"):
    idx = batch[name].to(dev)
    Tn = idx.shape[1]
    dxx = dx[: B * Tn].contiguous()
    dt = torch.zeros(V, C, device=dev); dinj = torch.zeros(B, C, device=dev)
    print(name, Tn, "synthetic  %.1f us" % timeit(lambda: _lib.call('coati_embed_bwd', ptr(idx), ptr(dxx), ptr(dt), ptr(dinj), 7, B, Tn, C, V, stream())))
    rnd = torch.randint(16, V, idx.shape, device=dev)
    print(name, "random ids %.1f us" % timeit(lambda: _lib.call('coati_embed_bwd', ptr(rnd), ptr(dxx), ptr(dt), ptr(dinj), 7, B, Tn, C, V, stream())))
    same = torch.full_like(idx, 5)
    print(name, "all same   %.1f us" % timeit(lambda: _lib.call('coati_embed_bwd', ptr(same), ptr(dxx), ptr(dt), ptr(dinj), 7, B, Tn, C, V, stream())))
    z = torch.zeros_like(dxx)
    print(name, "zero grads %.1f us" % timeit(lambda: _lib.call('coati_embed_bwd', ptr(rnd), ptr(z), ptr(dt), ptr(dinj), 7, B, Tn, C, V, stream())))
