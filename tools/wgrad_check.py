"""Correctness sweep of coati_wgrad (LDS-DMA kernel sizes) against torch fp32 on bf16-rounded operands."""
import sys, torch
sys.path.insert(0, ".")
from coati_amd import ops
dev = "cuda"
torch.manual_seed(0)
worst = 0.0
for (M, N, K) in [(74451, 256, 256), (81920, 768, 256), (74451, 768, 256), (40001, 1024, 256), (74451, 256, 1024), (65536 + 19, 264, 136),
                  (300000, 256, 256), (262144, 256, 512), (20000, 256, 256), (16384, 256, 256)]:
    A = torch.randn(M, N, device=dev).bfloat16()
    X = torch.randn(M, K, device=dev).bfloat16()
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    ops.wgrad(A, X, dW, db)
    torch.cuda.synchronize()
    ref = A.float().t() @ X.float()
    rb = A.float().sum(0)
    ew = ((dW - ref).abs().max() / ref.abs().max()).item()
    eb = ((db - rb).abs().max() / rb.abs().max()).item()
    dW2 = torch.zeros(N, K, device=dev)
    ops.wgrad(A, X, dW2, None)
    ew2 = ((dW2 - ref).abs().max() / ref.abs().max()).item()
    print(f"M={M} N={N} K={K}: dW {ew:.2e} dW(no bias) {ew2:.2e} db {eb:.2e}")
    worst = max(worst, ew, ew2, eb)
print("worst", worst)
assert worst < 1e-3
