"""Builds libcoati_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No JIT cache, no torch extension:
the library has a plain C ABI (include/coati_hip.h) and is loaded with ctypes."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcoati_hip_x.so" if os.environ.get("COATI_AMD_EXPERIMENTAL") == "1" else "libcoati_hip.so")
OBJDIR = os.path.join(HERE, "build")
HIP_UNITS = ["gemm.hip", "gemm_rb.hip", "gemm_rb16.hip", "gemm_ring.hip", "gemm_mx8.hip", "norm.hip", "attention.hip", "attention16.hip", "embed.hip", "gnn.hip", "loss.hip", "optim.hip", "batch.hip", "decode.hip"]
CPP_UNITS = ["engine.cpp", "capi.cpp", "tokenizer.cpp", "comm.cpp"]
# -amdgpu-mfma-vgpr-form: MFMA results in VGPRs (gfx950 has one unified register file).  Where the compiler picked the AGPR form
# (the attention forward kernels) 13 % of the instructions were v_accvgpr_read / write moves in a VALU-bound kernel
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + os.environ.get("COATI_AMD_CXXFLAGS", "").split()   # (probe builds: -DCOATI_RB_TRACE)


# COATI_AMD_EXPERIMENTAL=1: the three kernels of round 5 that are parity-green and SLOWER than what they would replace (the attention half of a
# block as one launch, the MLP half as one launch, the K = 256 products on 32-row slabs: DESIGN.md section 3a / 9, profiles/r05_*_trace.txt) are
# compiled in (csrc/experimental/, -DCOATI_EXPERIMENTAL) with their operators and their COATI_ATTN_BLOCK / COATI_MLP_FUSED / COATI_T32 switches;
# the default library neither holds nor exports them
EXPERIMENTAL = os.environ.get("COATI_AMD_EXPERIMENTAL") == "1"
EXPERIMENTAL_UNITS = ["experimental/attn_block.hip", "experimental/mlp64.hip", "experimental/gemm_t32.hip"]
if EXPERIMENTAL:
    HIP_UNITS = HIP_UNITS + EXPERIMENTAL_UNITS
    FLAGS = FLAGS + ["-DCOATI_EXPERIMENTAL"]
# mlp64.hip: one wave per SIMD with 256 accumulation registers next to 256 ordinary ones -- it needs the AGPR form of the MFMA results
UNIT_DROP_FLAGS = {"experimental/mlp64.hip": ("-mllvm", "-amdgpu-mfma-vgpr-form=1")}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _sources():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not os.path.isdir(os.path.join(CSRC, f))]
    if EXPERIMENTAL:
        deps += [os.path.join(CSRC, "experimental", f) for f in os.listdir(os.path.join(CSRC, "experimental"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "coati_hip.h"))
    return deps


def _stamp_path():
    return LIB + ".flags"


def needs_build():
    if not os.path.exists(LIB):
        return True
    # the compiler flags the library was built with (probe builds pass -D switches through COATI_AMD_CXXFLAGS): a library left behind by a
    # probe build must not pass for the product (round 6: a -DR16_TURNS=3 build survived an un-forced rebuild and produced NaN at 9-10 waves)
    try:
        with open(_stamp_path()) as f:
            if f.read() != " ".join(FLAGS):
                return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _sources())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    # several ranks of one node import the package at the same time (torch.distributed.run): one of them builds, the others
    # wait on the lock and then find the library up to date
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = _hipcc()

    def compile_one(unit):
        src = os.path.join(CSRC, unit)
        obj = os.path.join(OBJDIR, unit.replace("/", "_") + (".x.o" if EXPERIMENTAL else ".o"))
        flags = [f for f in FLAGS if f not in UNIT_DROP_FLAGS.get(unit, ())]
        cmd = [hipcc] + flags + (["-x", "hip"] if unit.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {unit}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, HIP_UNITS + CPP_UNITS))
    # link next to the target and rename: a rank that arrives while another one is linking (the unlocked fast path above)
    # must never dlopen a half-written library
    tmp = LIB + f".tmp{os.getpid()}"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lpthread", "-ldl", "-o", tmp], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    os.replace(tmp, LIB)
    with open(_stamp_path(), "w") as f:
        f.write(" ".join(FLAGS))
    if verbose:
        print(f"[coati_amd] built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
