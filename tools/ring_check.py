"""Correctness + timing of the N = 256 ring GEMM against torch (bf16-rounded operands, f32 accumulate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops
from gemm_bench_util import timeit
dev = "cuda:0"
torch.manual_seed(0)
worst = 0.0
for (M, K) in [(81920, 1024), (81920, 768), (74451, 1024), (40000, 512), (20481, 768)]:
    N = 256
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    o32 = torch.empty(M, N, device=dev)
    ops.gemm_nt(A, W, bias, ops.EPI_RES_F32, aux_in=res, out=o32)
    ops.gemm_nt(A, W, None, ops.EPI_BF16, out=o16)
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t()
    e1 = ((o32 - (ref + bias + res)).abs().max() / ref.abs().max()).item()
    e2 = ((o16.float() - ref).abs().max() / ref.abs().max()).item()
    t1 = min(timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_RES_F32, aux_in=res, out=o32)) for _ in range(5))
    t2 = min(timeit(lambda: ops.gemm_nt(A, W, None, ops.EPI_BF16, out=o16)) for _ in range(5))
    print(f"M={M} K={K}: res err {e1:.2e} {t1:7.1f} us {(M*K*2+M*N*8)/t1/1e6:.2f} TB/s | bf16 err {e2:.2e} {t2:7.1f} us {(M*K*2+M*N*2)/t2/1e6:.2f} TB/s")
    worst = max(worst, e1, e2 / 4)
assert worst < 2e-3, worst
