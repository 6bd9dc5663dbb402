"""Tile-quantisation check of the tiled NT GEMM: time per row at M = 65536 (2.0 rounds of 128-row tiles), 81920 (2.5), 98304 (3.0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops
from gemm_bench_util import timeit
dev = "cuda:0"
torch.manual_seed(0)
for (N, K) in [(256, 1024), (256, 768)]:
    for M in (65536, 81920, 98304):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        o32 = torch.empty(M, N, device=dev)
        t1 = timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_RES_F32, aux_in=res, out=o32))
        t2 = timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_BF16, out=o16))
        print(f"N={N} K={K} M={M}: res {t1*1e6:7.1f} us ({t1*1e9/M:.3f} ns/row, {(M*K*2+M*N*8)/t1/1e12:.2f} TB/s)   bf16 {t2*1e6:7.1f} us ({t2*1e9/M:.3f} ns/row, {(M*K*2+M*N*2)/t2/1e12:.2f} TB/s)")
