"""Where does a 16-row-slab GEMM launch (gemm_rb16_kernel, K = 256, 50 000 rows) spend its time?  Probe builds of the library with
parts of the tile loop removed (results are wrong, the timing is what is read):
    for n in 0 1 2 3 4 7; do COATI_AMD_CXXFLAGS=-DR16_ABLATE=$n python -m coati_amd.build --force; python tools/rb16_ablate.py $n; done
bits: 1 = no weight stream behind the first tile, 2 = no stores in the write-out, 4 = one LDS operand read per tile instead of 8."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops
from gemm_bench_util import timeit, row

dev, M, K = "cuda:0", 50000, 256
torch.manual_seed(0)
tag = sys.argv[1] if len(sys.argv) > 1 else "?"
A = torch.randn(M, K, device=dev).bfloat16()
for N in (1024, 768):
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row(f"[ablate {tag}] N={N} bf16 out", timeit(lambda: ops.gemm_nt(A, W, b, ops.EPI_BF16, out=o16), reps=50), 2.0 * M * N * K, M * K * 2 + M * N * 2)
    if N == 1024:
        row(f"[ablate {tag}] N={N} NewGELU + codes", timeit(lambda: ops.gemm_nt(A, W, b, ops.EPI_GELU_GRAD, out=o16), reps=50), 2.0 * M * N * K, M * K * 2 + M * N * 3)
