"""Phase breakdown of the fused MLP forward (csrc/mlp64.hip) from shader-clock stamps.  Needs a probe build:
   COATI_AMD_CXXFLAGS=-DCOATI_M64_TRACE COATI_AMD_REBUILD=1 python tools/probes/m64_trace.py [M]
(add -DM64_ABLATE=bits for the timing-only ablations; rebuild without the flags afterwards)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coati_amd import ops, _lib
import numpy as np
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 256, generator=g).to(dev)
gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(dev); beta = (0.1 * torch.randn(256, generator=g)).to(dev)
W1 = (torch.randn(1024, 256, generator=g) * 0.06).to(dev).bfloat16(); b1 = (0.1 * torch.randn(1024, generator=g)).to(dev)
W2 = (torch.randn(256, 1024, generator=g) * 0.03).to(dev).bfloat16(); b2 = (0.1 * torch.randn(256, generator=g)).to(dev)
a2 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
gg = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16); codes = torch.empty(M, 1024, device=dev, dtype=torch.uint8); out = torch.empty(M, 256, device=dev)
p = ops.ptr
def f():
    _lib.call("coati_mlp_fwd", p(x), p(gamma), p(beta), p(a2), p(mean), p(rstd), p(W1), p(b1), p(W2), p(b2), p(gg), p(codes), p(out), M, ops.stream())
for _ in range(3): f()
torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 10 * 1e3
print(f"fused MLP forward, M = {M}: {us:.1f} us/launch")
buf = (ctypes.c_uint64 * (16 * 4 * 8))()
if hasattr(lib, "coati_m64_trace_read") and lib.coati_m64_trace_read(buf) == 0:
    a = np.array(buf, dtype=np.float64).reshape(16, 4, 8)
    tot = a[:, :, 7].mean()
    names = ["slab load + LayerNorm", "tile barrier", "FC1 + NewGELU + stores + FC2", "wait for the next tile's DMA", "final write-out"]
    print(f"  shader-clock ticks per wave (first 16 workgroups): whole kernel {tot:.0f}")
    for i, n in enumerate(names):
        print(f"    {n:30s} {a[:, :, i].mean():9.0f}  {100 * a[:, :, i].mean() / tot:5.1f}%   min {a[:, :, i].min():7.0f} max {a[:, :, i].max():7.0f}")
