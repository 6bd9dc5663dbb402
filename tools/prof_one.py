"""Runs a few launches of selected kernels at the step's shapes (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coati_amd import ops
dev = "cuda:0"
M = 81920
which = sys.argv[1:] or ["nt768", "gelu1024", "wgrad768", "wgrad1024", "attn"]
torch.manual_seed(0)
def mk(N, K):
    return (torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) * 0.05).bfloat16(), torch.randn(N, device=dev))
for w in which:
    if w == "nt768":
        A, W, b = mk(768, 256); o = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
        for _ in range(5): ops.gemm_nt(A, W, b, ops.EPI_BF16, out=o)
    if w == "gelu1024":
        A, W, b = mk(1024, 256); o = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
        for _ in range(5): ops.gemm_nt(A, W, b, ops.EPI_GELU, out=o)
    if w == "nt256k1024":
        A, W, b = mk(256, 1024); o = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
        for _ in range(5): ops.gemm_nt(A, W, b, ops.EPI_BF16, out=o)
    if w in ("wgrad768", "wgrad1024"):
        N = 768 if w == "wgrad768" else 1024
        A = torch.randn(M, N, device=dev).bfloat16(); X = torch.randn(M, 256, device=dev).bfloat16()
        dW = torch.zeros(N, 256, device=dev); db = torch.zeros(N, device=dev)
        for _ in range(5): ops.wgrad(A, X, dW, db)
    if w == "attn":
        B, T, nh = 1024, 80, 16
        qkv = torch.randn(B*T, 768, device=dev).bfloat16(); cos, sin = ops.rope_tables(250, 16, device=dev)
        dy = torch.randn(B*T, 256, device=dev).bfloat16()
        for _ in range(5):
            y, lse = ops.attn_fwd(qkv, B, T, nh)
            ops.attn_bwd(qkv, y, dy, lse, B, T, nh, cos, sin)
torch.cuda.synchronize()
