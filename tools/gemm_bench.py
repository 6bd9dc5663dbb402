"""Micro-benchmark of the GEMM family at the shapes of the grande step (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops

dev = "cuda:0"
M = 81920

from gemm_bench_util import timeit, row

torch.manual_seed(0)
for (N, K) in [(768, 256), (256, 256), (1024, 256), (256, 1024), (256, 768), (1024, 1024)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    A32 = torch.randn(M, K, device=dev)
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    pre = torch.randn(M, N, device=dev).bfloat16()
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    o32 = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    row(f"nt bf16   N={N} K={K}", timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_BF16, out=o16)), fl, M*K*2 + M*N*2)
    row(f"nt f32out N={N} K={K}", timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_F32, out=o32)), fl, M*K*2 + M*N*4)
    row(f"nt res    N={N} K={K}", timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_RES_F32, aux_in=res, out=o32)), fl, M*K*2 + M*N*8)
    row(f"nt gelu   N={N} K={K}", timeit(lambda: ops.gemm_nt(A, W, bias, ops.EPI_GELU, out=o16)), fl, M*K*2 + M*N*4)
    row(f"nt dgelu  N={N} K={K} (A f32)", timeit(lambda: ops.gemm_nt(A32, W, None, ops.EPI_DGELU, aux_in=pre, out=o16)), fl, M*K*4 + M*N*4)
    row(f"nt bf16   N={N} K={K} (A f32)", timeit(lambda: ops.gemm_nt(A32, W, None, ops.EPI_BF16, out=o16)), fl, M*K*4 + M*N*2)
for (N, K, f32) in [(768, 256, False), (256, 256, False), (1024, 256, False), (256, 1024, False), (256, 256, True), (256, 1024, True)]:
    A = torch.randn(M, N, device=dev)
    A = A if f32 else A.bfloat16()
    X = torch.randn(M, K, device=dev).bfloat16()
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    fl = 2.0 * M * N * K
    row(f"wgrad N={N} K={K} f32A={f32} +bias", timeit(lambda: ops.wgrad(A, X, dW, db)), fl, M*N*(4 if f32 else 2) + M*K*2)
    row(f"wgrad N={N} K={K} f32A={f32}", timeit(lambda: ops.wgrad(A, X, dW, None)), fl, M*N*(4 if f32 else 2) + M*K*2)
# attention + layernorm
B, T, nh = 1024, 80, 16
qkv = torch.randn(B*T, 768, device=dev).bfloat16()
cos, sin = ops.rope_tables(250, 16, device=dev)
y, lse = ops.attn_fwd(qkv, B, T, nh)
dy = torch.randn(B*T, 256, device=dev).bfloat16()
row("attn fwd", timeit(lambda: ops.attn_fwd(qkv, B, T, nh)), 4.0*B*T*T*256, B*T*1024*2)
row("attn bwd", timeit(lambda: ops.attn_bwd(qkv, y, dy, lse, B, T, nh, cos, sin)), 10.0*B*T*T*256, B*T*2560*2)
x = torch.randn(B*T, 256, device=dev); g = torch.ones(256, device=dev); b = torch.zeros(256, device=dev)
y16, _, mean, rstd = ops.layernorm_fwd(x, g, b)
row("ln fwd", timeit(lambda: ops.layernorm_fwd(x, g, b)), 0, B*T*256*6)
row("ln bwd", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres=x)), 0, B*T*256*14)
row("ln bwd (no dgamma/dbeta)", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres=x, want_affine_grads=False)), 0, B*T*256*14)
row("ln bwd (no dres)", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g)), 0, B*T*256*10)
