"""Phase breakdown of the one-round ring GEMM (gemm_ring1_kernel, csrc/gemm_ring.hip) from shader-clock stamps, at a packed-batch size.
Needs a probe build:
   COATI_AMD_CXXFLAGS=-DCOATI_RB_TRACE COATI_AMD_REBUILD=1 python tools/probes/ring1_trace.py [M]
(rebuild without the flag afterwards).  `cold`: a 1.5-GB buffer is rewritten between the timed launches, as in the training step."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coati_amd import ops, _lib
import numpy as np
dev = "cuda:0"
args = [a for a in sys.argv[1:] if a != "cold"]
cold = "cold" in sys.argv[1:]
M = int(args[0]) if args else 50000
lib = _lib.lib()
names = ["before the loop", "wait (vmcnt + barrier)", "MFMA + DMA issue", "write-out (LayerNorm bwd)", "dgamma / dbeta", "chained product"]
junk = torch.empty(1536 * 1024 * 1024 // 4, device=dev) if cold else None
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    if cold:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            junk.add_(1.0)
            a.record(); f(); b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / n * 1e3
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def report(tag, f, mb):
    us = timeit(f)
    print(f"{tag}: {us:.1f} us/launch  ({mb:.0f} MB of operands: {mb / us:.2f} TB/s)")
    buf = (ctypes.c_uint64 * (16 * 16 * 8))()
    if not hasattr(lib, "coati_rg_trace_read") or lib.coati_rg_trace_read(buf) != 0:
        print("  (no trace: not a probe build)"); return
    a = np.array(buf, dtype=np.float64).reshape(16, 16, 8)[:, :, :6]
    a = a[:, a.sum((0, 2)) > 0, :]
    tot = a.sum(-1).mean()
    print(f"  shader-clock ticks per wave (first 16 workgroups, {a.shape[1]} waves): {tot:.0f}")
    for i, n in enumerate(names):
        if a[:, :, i].sum() > 0:
            print(f"    {n:28s} {a[:, :, i].mean():9.0f}  {100 * a[:, :, i].mean() / tot:5.1f}%   min {a[:, :, i].min():7.0f} max {a[:, :, i].max():7.0f}")
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 256, generator=g).to(dev)
mean = x.mean(-1); rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt()
gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(dev)
dres = torch.randn(M, 256, generator=g).to(dev)
for K, chain, tag in ((1024, True, "FC1 input gradient + ln_2 backward + c_proj input gradient (K = 1024, chained)"),
                      (768, False, "QKV input gradient + ln_1 backward (K = 768)")):
    dY = torch.randn(M, K, generator=g).to(dev).bfloat16()
    WT = (torch.randn(256, K, generator=g) * 0.05).to(dev).bfloat16()
    Wc = (torch.randn(256, 256, generator=g) * 0.05).to(dev).bfloat16() if chain else None
    mb = (M * K * 2 + M * 256 * (4 + 4 + 4 + 2) + (M * 256 * 2 if chain else 0)) / 1e6
    report(tag, lambda: ops.gemm_lnbwd(dY, WT, x, mean, rstd, gamma, dres, True, Wc), mb)
G = torch.randn(M, 1024, generator=g).to(dev).bfloat16()
W2 = (torch.randn(256, 1024, generator=g) * 0.05).to(dev).bfloat16()
b2 = torch.randn(256, generator=g).to(dev)
out = torch.empty(M, 256, device=dev)
names[3] = "write-out"
report("FC2 + residual (K = 1024, f32 out)", lambda: ops.gemm_nt(G, W2, b2, ops.EPI_RES_F32, aux_in=x, out=out), (M * 1024 * 2 + M * 256 * 8) / 1e6)
Y = torch.randn(M, 256, generator=g).to(dev).bfloat16()
Wp = (torch.randn(256, 256, generator=g) * 0.05).to(dev).bfloat16()
report("proj + residual (K = 256, f32 out)", lambda: ops.gemm_nt(Y, Wp, b2, ops.EPI_RES_F32, aux_in=x, out=out), (M * 256 * 2 + M * 256 * 8) / 1e6)
