"""Ring GEMM (N = 256) time against the number of rounds of row blocks over the 256 persistent workgroups: how much does the
partial second round of a packed batch (391 blocks of 128 rows = 1.53 rounds) cost?  python tools/ring_rounds.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops
from gemm_bench_util import timeit, row

dev = "cuda:0"
torch.manual_seed(0)
for K in (1024, 256):
    W = (torch.randn(256, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(256, device=dev)
    rows_env = "auto"
    for M in (32768, 36000, 40960, 50000, 65536, 81920):
        A = torch.randn(M, K, device=dev).bfloat16()
        res = torch.randn(M, 256, device=dev)
        o16 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
        row(f"K={K} bf16 out   M={M} blocks of {rows_env}", timeit(lambda: ops.gemm_nt(A, W, None, ops.EPI_BF16, out=o16)), 2.0 * M * 256 * K, M * K * 2 + M * 512)
        row(f"K={K} + residual M={M} blocks of {rows_env}", timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_RES_F32, aux_in=res)), 2.0 * M * 256 * K, M * K * 2 + M * 2048)
