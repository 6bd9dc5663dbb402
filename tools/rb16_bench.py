"""16-row-slab row-block GEMM (gemm_rb16.hip) at a packed batch size: quick A/B timing on the GPU box.
   python tools/rb16_bench.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops
from gemm_bench_util import timeit, row

dev, K = "cuda:0", 256
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
torch.manual_seed(0)
A = torch.randn(M, K, device=dev).bfloat16()
cos, sin = ops.rope_tables(250, 16, device=dev)
for N in (768, 1024):
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row(f"nt bf16   N={N}", timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_BF16, out=o16)), 2.0 * M * N * K, M*K*2 + M*N*2)
W = (torch.randn(1024, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(1024, device=dev)
row("fc1 gelu + gelu' N=1024", timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_GELU_GRAD)), 2.0*M*1024*K, M*K*2 + M*1024*3)
xq = torch.randint(0, 255, (M, 1024), device=dev, dtype=torch.uint8)
o16 = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
row("fc2 dgrad x gelu' N=1024", timeit(lambda: ops.gemm_nt(A, W, None, ops.EPI_MUL_AUX, aux_in=xq, out=o16)), 2.0*M*1024*K, M*K*2 + M*1024*3)
W = (torch.randn(768, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(768, device=dev)
row("qkv + rope N=768", timeit(lambda: ops.gemm_qkv_rope(A, W, bias, 80, cos, sin)), 2.0*M*768*K, M*K*2 + M*768*2)
