"""Row-block GEMM shapes only (K=256, N>=512): quick A/B timing on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops
from gemm_bench_util import timeit, row

dev, M, K = "cuda:0", 81920, 256
torch.manual_seed(0)
A = torch.randn(M, K, device=dev).bfloat16()
cos, sin = ops.rope_tables(250, 16, device=dev)
for N in (768, 1024):
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    row(f"nt bf16   N={N}", timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_BF16, out=o16)), fl, M*K*2 + M*N*2)
    row(f"nt gelu   N={N}", timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_GELU, out=o16)), fl, M*K*2 + M*N*4)
W = (torch.randn(768, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(768, device=dev)
row("qkv + rope N=768", timeit(lambda: ops.gemm_qkv_rope(A, W, bias, 80, cos, sin)), 2.0*M*768*K, M*K*2 + M*768*2)
