"""Phase breakdown of the 32-row-slab transposed GEMM (csrc/gemm_t32.hip) from shader-clock stamps.  Needs a probe build:
   COATI_T32=1 COATI_AMD_CXXFLAGS=-DCOATI_T32_TRACE COATI_AMD_REBUILD=1 python tools/probes/t32_trace.py [M]
(rebuild without the flag afterwards)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coati_amd import ops, _lib
import numpy as np
dev, K = "cuda:0", 256
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
lib = _lib.lib()
torch.manual_seed(0)
A = torch.randn(M, K, device=dev).bfloat16()
names = ["slab load", "tile barrier", "mfma loop", "wait next tile", "write-out", "-", "-", "whole kernel"]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def report(tag, f):
    us = timeit(f)
    buf = (ctypes.c_uint64 * (16 * 8 * 8))()
    rc = lib.coati_t32_trace_read(buf)
    print(f"{tag}: {us:.1f} us/launch")
    if rc != 0:
        print("  (no trace: not a probe build)"); return
    a = np.array(buf, dtype=np.float64).reshape(16, 8, 8)
    a = a[:, a[0, :, 7] > 0, :]
    tot = a[:, :, 7].mean()
    print(f"  shader-clock ticks per wave (first 16 workgroups, {a.shape[1]} waves): whole kernel {tot:.0f}  -> {us / tot * 1e3:.2f} ns per tick if the launch were one wave long")
    for i, n in enumerate(names[:5]):
        print(f"    {n:16s} {a[:, :, i].mean():9.0f}  {100 * a[:, :, i].mean() / tot:5.1f}%   min {a[:, :, i].min():7.0f} max {a[:, :, i].max():7.0f}")
for N in (768, 1024):
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    report(f"nt bf16 N={N}", lambda: ops.gemm_nt(A, W, bias, ops.EPI_BF16, out=o16))
W = (torch.randn(1024, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(1024, device=dev)
report("fc1 gelu + gelu' N=1024", lambda: ops.gemm_nt(A, W, bias, ops.EPI_GELU_GRAD))
