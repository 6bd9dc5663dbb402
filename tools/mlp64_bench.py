"""The fused MLP forward (csrc/mlp64.hip) against the two launches it replaces, at a packed-batch size.   python tools/mlp64_bench.py [M] [cold]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coati_amd import ops, _lib
dev = "cuda:0"
args = [a for a in sys.argv[1:] if a != "cold"]
cold = "cold" in sys.argv[1:]
M = int(args[0]) if args else 50000
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 256, generator=g).to(dev)
gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(dev); beta = (0.1 * torch.randn(256, generator=g)).to(dev)
W1 = (torch.randn(1024, 256, generator=g) * 0.06).to(dev).bfloat16(); b1 = (0.1 * torch.randn(1024, generator=g)).to(dev)
W2 = (torch.randn(256, 1024, generator=g) * 0.03).to(dev).bfloat16(); b2 = (0.1 * torch.randn(256, generator=g)).to(dev)
a2 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
gg = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16); codes = torch.empty(M, 1024, device=dev, dtype=torch.uint8); out = torch.empty(M, 256, device=dev)
p = ops.ptr
junk = torch.empty(1536 * 1024 * 1024 // 4, device=dev) if cold else None
def fused():
    _lib.call("coati_mlp_fwd", p(x), p(gamma), p(beta), p(a2), p(mean), p(rstd), p(W1), p(b1), p(W2), p(b2), p(gg), p(codes), p(out), M, ops.stream())
A = torch.randn(M, 256, generator=g).to(dev).bfloat16()
def two():
    h2, x8 = ops.gemm_nt(A, W1, b1, ops.EPI_GELU_GRAD)       # (without the LayerNorm in the operand load: + ~ 5 us in the step)
    ops.gemm_nt(h2, W2, b2, ops.EPI_RES_F32, aux_in=x, out=out)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        if cold: junk.add_(1.0)
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / n * 1e3
print(f"M = {M}{' (cold caches)' if cold else ''}")
print(f"  fused MLP forward (coati_mlp_fwd)          {timeit(fused):7.1f} us")
print(f"  two launches (FC1 + NewGELU', FC2 + res)   {timeit(two):7.1f} us")
