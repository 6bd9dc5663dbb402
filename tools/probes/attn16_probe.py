"""Stand-alone timing of the 16-row-granular attention kernels (coati_amd/csrc/attention16.hip) on the training batch's length
distribution, with probe variants of the kernel built from the same source under -D switches.

  python tools/probes/attn16_probe.py --build            # here (hipcc cross-compiles): tools/probes/a16_<variant>.so
  python tools/probes/attn16_probe.py --run [variants]   # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "coati_amd", "csrc")
VARIANTS = {
    "base": [],
    "nocompute": ["-DA16_PROBE_NOCOMPUTE"],
    "nostore": ["-DA16_PROBE_NOSTORE"],
    "lds8k": ["-DA16_PROBE_LDS_EXTRA=8192"],      # one workgroup fewer per CU
    "lds16k": ["-DA16_PROBE_LDS_EXTRA=16384"],
}
WRAP = r'''
#include <cstdarg>
#include <cstdio>
#include "kernels.h"
void coati_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
extern "C" int a16_fwd(const bf16_t* qkv, bf16_t* y, float* lse, int B, int T, int nh, const int* off, void* s) {
  return launch_attn16_fwd(qkv, y, lse, B, T, nh, (hipStream_t)s, off);
}
extern "C" int a16_bwd(const bf16_t* qkv, const bf16_t* y, const bf16_t* dy, const float* lse, bf16_t* dqkv, const float* c, const float* sn,
                       int B, int T, int nh, const int* off, void* s) {
  return launch_attn16_bwd(qkv, y, dy, lse, dqkv, c, sn, B, T, nh, (hipStream_t)s, off);
}
'''


def build(names):
    wrap = os.path.join(HERE, "a16_wrap.cpp")
    open(wrap, "w").write(WRAP)
    for n in names:
        flags = VARIANTS.get(n) or [f for f in os.environ.get("A16_FLAGS_" + n, "").split()]
        out = os.path.join(HERE, f"a16_{n}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-mllvm",
               "-amdgpu-mfma-vgpr-form=1", "-I", CSRC] + flags + ["-x", "hip", os.path.join(CSRC, "attention16.hip"), wrap, "-o", out]
        subprocess.run(cmd, check=True)
        print("built", out)


def run(names, seq=80, B=1024, nh=16, reps=30, nbuf=4):
    import torch
    sys.path.insert(0, ROOT)
    from coati_amd.ops import rope_tables
    dev = "cuda:0"
    g = torch.Generator().manual_seed(1234)
    res = {}
    for which, extra in (("pass1", 2), ("pass2", 4)):
        lens = (torch.randint(16, seq - 4 + 1, (B,), generator=g) + extra).to(torch.int32)
        lens[0] = seq - 4 + extra
        off = torch.zeros(B + 1, dtype=torch.int32)
        off[1:] = torch.cumsum(lens, 0)
        M = int(off[-1])
        T = int(lens.max())
        C = nh * 16
        bufs = []
        for i in range(nbuf):
            qkv = (torch.randn(M, 3 * C, generator=g) * 1.0).bfloat16().to(dev)
            dy = torch.randn(M, C, generator=g).bfloat16().to(dev)
            bufs.append((qkv, torch.empty(M, C, device=dev, dtype=torch.bfloat16), torch.empty(B, nh, T, device=dev), dy, torch.empty_like(qkv)))
        offd = off.to(dev)
        cos, sin = rope_tables(256, 16, device=dev)
        nblk = sum(int((int(l) + 15) // 16) * (int((int(l) + 15) // 16) + 1) // 2 for l in lens)
        useful = sum(int(l) * (int(l) + 1) // 2 for l in lens)
        fwd_bytes = M * (3 * C + C) * 2 + B * nh * 4 * M / B
        bwd_bytes = M * (3 * C + C + C + 3 * C) * 2 + M * nh * 4
        for n in names:
            lib = ctypes.CDLL(os.path.join(HERE, f"a16_{n}.so"))
            P = ctypes.c_void_p
            st = P(torch.cuda.current_stream().cuda_stream)

            def fwd(i):
                q, y, lse, dy, dq = bufs[i % nbuf]
                rc = lib.a16_fwd(P(q.data_ptr()), P(y.data_ptr()), P(lse.data_ptr()), B, T, nh, P(offd.data_ptr()), st)
                assert rc == 0

            def bwd(i):
                q, y, lse, dy, dq = bufs[i % nbuf]
                rc = lib.a16_bwd(P(q.data_ptr()), P(y.data_ptr()), P(dy.data_ptr()), P(lse.data_ptr()), P(dq.data_ptr()), P(cos.data_ptr()),
                                 P(sin.data_ptr()), B, T, nh, P(offd.data_ptr()), st)
                assert rc == 0
            for name, f, nbytes in (("fwd", fwd, fwd_bytes), ("bwd", bwd, bwd_bytes)):
                for i in range(nbuf):
                    f(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    f(i)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1000.0 / reps
                res[(which, n, name)] = us
                print(f"{which} rows {M} T {T} {n:>12s} {name}: {us:7.2f} us  {nbytes / us / 1e6:6.2f} TB/s   (useful/computed scores {useful / (256.0 * nblk):.3f})", flush=True)
    return res


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(VARIANTS)
    if "--build" in sys.argv:
        build(names)
    if "--run" in sys.argv:
        run(names)
