"""Phase breakdown of the fused attention-half kernel (csrc/attn_block.hip) from shader-clock stamps.  Needs a probe build:
   COATI_AMD_CXXFLAGS=-DCOATI_AB_TRACE COATI_AMD_REBUILD=1 python tools/probes/ab_trace.py
(rebuild without the flag afterwards).  Workload: B = 1024 sequences of U{16..76} tokens (the bench's packed decoder pass)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coati_amd import ops, _lib
import numpy as np
dev = "cuda:0"
B, T, C = 1024, 80, 256
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
lens = torch.randint(16, 77, (B,), generator=g)
lens[0] = T
M = int(lens.sum())
keep = torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)
src = keep.view(-1).nonzero().squeeze(1).to(dev, torch.int32)
off = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)]).to(dev, torch.int32)
x = torch.randn(M, C, generator=g).to(dev)
ln_g = torch.ones(C, device=dev); ln_b = torch.zeros(C, device=dev)
Wqkv = (torch.randn(3 * C, C, generator=g) * 0.08).to(dev).bfloat16(); bqkv = torch.zeros(3 * C, device=dev)
Wproj = (torch.randn(C, C, generator=g) * 0.06).to(dev).bfloat16(); bproj = torch.zeros(C, device=dev)
cos, sin = ops.rope_tables(250, 16, device=dev)
grp = ops.attn_groups(off, B, T)
xmid = torch.zeros(M, C, device=dev); a1 = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
mean = torch.zeros(M, device=dev); rstd = torch.zeros(M, device=dev)
qkv = torch.zeros(M, 3 * C, device=dev, dtype=torch.bfloat16); y = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(B, 16, T, device=dev)
p = ops.ptr
def f():
    _lib.call("coati_attn_block_fwd", p(x), p(xmid), p(ln_g), p(ln_b), p(mean), p(rstd), p(a1), p(Wqkv), p(bqkv), p(Wproj), p(bproj),
              p(qkv), p(y), p(lse), p(cos), p(sin), p(src), p(grp), T, M, ops.stream())
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
us = timeit(f)
ng = int(grp[0])
if len(sys.argv) > 1 and sys.argv[1] == "cold":
    # every launch on cold caches, as in the training step (87 GB of traffic between two launches of a layer): a 1.5-GB buffer is
    # rewritten between the timed launches (L2 4 MB x 8, Infinity Cache 256 MB)
    junk = torch.empty(1536 * 1024 * 1024 // 4, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        junk.add_(1.0)
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    us = sum(a.elapsed_time(b) for a, b in ev) / len(ev) * 1e3
    print("(cold caches between launches)")
print(f"attn_block_fwd: {M} rows, {ng} groups, {us:.1f} us/launch")
buf = (ctypes.c_uint64 * (16 * 8 * 8))()
if lib.coati_ab_trace_read(buf) == 0:
    a = np.array(buf, dtype=np.float64).reshape(16, 8, 8)
    names = ["layernorm+slab", "stage wait+barrier", "gemm1 q/k/v", "attention", "gemm2 proj", "image copies", "write-out", "other barriers"]
    tot = a.sum(-1).mean()
    print(f"cycles per wave (first 16 workgroups, {100e6 and ''}all their groups): {tot:.0f}")
    for i, n in enumerate(names):
        print(f"  {n:20s} {a[:, :, i].mean():9.0f}  {100 * a[:, :, i].mean() / tot:5.1f}%   dma waves {a[:, :4, i].mean():9.0f}  copy waves {a[:, 4:, i].mean():9.0f}")
    print("  wg 0 per wave totals: " + " ".join(f"{a[0, w].sum():.0f}" for w in range(8)))
else:
    print("(no trace: build with -DCOATI_AB_TRACE)")
