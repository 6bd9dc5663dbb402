"""Where does a one-round ring GEMM launch (gemm_ring1_kernel, N = 256, 50 000 rows) spend its time?  Probe builds of the library
with parts of the k loop's operand streams removed (results are wrong, the timing is what is read):
    for n in 0 1 2 3; do COATI_AMD_CXXFLAGS=-DRG_ABLATE=$n python -m coati_amd.build --force; python tools/ring_ablate.py $n; done
0 = the product kernel, 1 = no weight stream behind the prologue, 2 = no A stream, 3 = neither (barriers + LDS reads + MFMAs only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops
from gemm_bench_util import timeit, row

dev, M = "cuda:0", 50000
torch.manual_seed(0)
tag = sys.argv[1] if len(sys.argv) > 1 else "?"
for K in (1024, 768, 256):
    W = (torch.randn(256, K, device=dev) * 0.05).bfloat16()
    A = torch.randn(M, K, device=dev).bfloat16()
    res = torch.randn(M, 256, device=dev)
    o16 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
    row(f"[ablate {tag}] K={K} bf16 out", timeit(lambda: ops.gemm_nt(A, W, None, ops.EPI_BF16, out=o16), reps=50), 2.0 * M * 256 * K, M * K * 2 + M * 512)
    row(f"[ablate {tag}] K={K} + residual", timeit(lambda: ops.gemm_nt(A, W, None, ops.EPI_RES_F32, aux_in=res), reps=50), 2.0 * M * 256 * K, M * K * 2 + M * 2048)
