"""Phase breakdown of the row-block GEMM (gemm_rb256_kernel) from shader-clock stamps.  Needs a probe build:
   COATI_AMD_CXXFLAGS=-DCOATI_RB_TRACE COATI_AMD_REBUILD=1 python tools/probes/rb_trace.py
(rebuild without the flag afterwards).  Prints, for the FC1 forward shape (LayerNorm-fused, GELU + derivative) and the lm_head
partial-CE shape, the average cycles per wave spent in: prologue | MFMA phases | vmcnt waits | epilogues | barriers."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coati_amd import ops, _lib
import numpy as np
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 81920     # 81920 = the padded batch; ~50000 = a packed one
lib = _lib.lib()
def dump(tag, us):
    buf = (ctypes.c_uint64 * (16 * 12 * 8))()     # [wg][wave <= 12][8]
    assert lib.coati_rb_trace_read(buf) == 0
    a = np.array(buf, dtype=np.float64).reshape(16, 12, 8)[:, :, :5]
    a = a[:, a.sum((0, 2)) > 0, :]                  # waves that ran (10, or 8 + 4)
    tot = a.sum(-1).mean()
    names = ["prologue", "mfma", "vmwait", "epilogue", "barrier"]
    print(f"{tag}: {us:.1f} us/launch; cycles per wave {tot:.0f} = " + "  ".join(f"{n} {a[:, :, i].mean():.0f} ({100 * a[:, :, i].mean() / tot:.0f}%)" for i, n in enumerate(names)))
    print("   per wave (wg 0): " + " | ".join(" ".join(f"{a[0, w, i]:.0f}" for i in range(5)) for w in (0, 4, a.shape[1] - 1)))
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
g = torch.Generator().manual_seed(0)
A = torch.randn(M, 256, generator=g).to(dev).bfloat16()
W1 = (torch.randn(1024, 256, generator=g) * 0.05).to(dev).bfloat16()
b1 = torch.randn(1024, generator=g).to(dev)
f = lambda: ops.gemm_nt(A, W1, b1, ops.EPI_GELU_GRAD)
us = timeit(f); dump("FC1 + GELU/GELU' (bf16 A)", us)
W3 = (torch.randn(768, 256, generator=g) * 0.05).to(dev).bfloat16()
f = lambda: ops.gemm_nt(A, W3, None, ops.EPI_BF16)
us = timeit(f); dump("plain bf16 N=768", us)


def dump_ring(tag, us):
    buf = (ctypes.c_uint64 * (16 * 16 * 8))()
    assert lib.coati_rg_trace_read(buf) == 0
    a = np.array(buf, dtype=np.float64).reshape(16, 16, 8)[:, :, :4]
    a = a[:, a.sum((0, 2)) > 0, :]                  # waves that ran (10, or 8 in the 128-row form)
    tot = a.sum(-1).mean()
    names = ["before loop", "wait (vmcnt + barrier)", "MFMA + DMA issue", "write-out"]
    print(f"{tag}: {us:.1f} us/launch; cycles per wave {tot:.0f} = " + "  ".join(f"{n} {a[:, :, i].mean():.0f} ({100 * a[:, :, i].mean() / tot:.0f}%)" for i, n in enumerate(names)))
    print("   per wave (wg 0): " + " | ".join(" ".join(f"{a[0, w, i]:.0f}" for i in range(4)) for w in (0, 1, 4, a.shape[1] - 1)))
G = torch.randn(M, 1024, generator=g).to(dev).bfloat16()
W2 = (torch.randn(256, 1024, generator=g) * 0.05).to(dev).bfloat16()
b2 = torch.randn(256, generator=g).to(dev)
X = torch.randn(M, 256, generator=g).to(dev)
out = torch.empty(M, 256, device=dev)
f = lambda: ops.gemm_nt(G, W2, b2, ops.EPI_RES_F32, aux_in=X, out=out)
us = timeit(f); dump_ring("ring: FC2 + residual (K = 1024, f32 out)", us)
out16 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
f = lambda: ops.gemm_nt(G, W2, None, ops.EPI_BF16, out=out16)
us = timeit(f); dump_ring("ring: FC1 input gradient (K = 1024, bf16 out)", us)
Y = torch.randn(M, 256, generator=g).to(dev).bfloat16()
Wp = (torch.randn(256, 256, generator=g) * 0.05).to(dev).bfloat16()
f = lambda: ops.gemm_nt(Y, Wp, b2, ops.EPI_RES_F32, aux_in=X, out=out)
us = timeit(f); dump_ring("ring: proj + residual (K = 256, f32 out)", us)
