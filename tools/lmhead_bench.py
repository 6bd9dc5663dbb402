"""lm_head partial cross-entropy and its gradient at the bench's decoder-pass size: quick A/B timing (COATI_T32=1 selects the
32-row-slab transposed kernel).   python tools/lmhead_bench.py [M] [V]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops
from gemm_bench_util import timeit, row

dev, K = "cuda:0", 256
M = int(sys.argv[1]) if len(sys.argv) > 1 else 51265
V = int(sys.argv[2]) if len(sys.argv) > 2 else 10322
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(dev).bfloat16()
W = (torch.randn(V, K, generator=g) * 0.2).to(dev).bfloat16()
tgt = torch.randint(0, V, (M,), generator=g).to(dev)
lse, scal = ops.ce_fwd(a, W, tgt)
row(f"lm_head partial CE  M={M} V={V}", timeit(lambda: ops.ce_fwd(a, W, tgt)), 2.0 * M * V * K, M * K * 2 + V * K * 2)
row(f"lm_head dlogits     M={M} V={V}", timeit(lambda: ops.ce_bwd(a, W, tgt, lse, scal)), 2.0 * M * V * K, M * K * 2 + V * K * 2 + M * V * 2)
