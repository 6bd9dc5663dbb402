"""The four transformer weight-gradient launches of one layer at the bench shapes (M = 1024 x 80 rows), 5 rounds: the
`xf_wgrad` site in isolation, for rocprofv3 PMC passes (HBM bytes per launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coati_amd import ops
dev, M = "cuda:0", 81920
torch.manual_seed(0)
shapes = [(768, 256), (256, 256), (1024, 256), (256, 1024)]   # (N, K): qkv, proj, fc1, fc2
ops_ = []
for N, K in shapes:
    A = torch.randn(M, N, device=dev).bfloat16(); X = torch.randn(M, K, device=dev).bfloat16()
    ops_.append((A, X, torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)))
for _ in range(5):
    for A, X, dW, db in ops_:
        ops.wgrad(A, X, dW, db)
torch.cuda.synchronize()
