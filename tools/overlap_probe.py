"""Does running the weight gradients on a second stream next to the dgrad chain buy anything?  One transformer-layer's
worth of backward kernels, serial on one stream vs wgrads on a side stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coati_amd import ops
dev = "cuda:0"
M, C = 81920, 256
B, T, nh = 1024, 80, 16
torch.manual_seed(0)
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
dx16, g, hpre, a2, y, a1 = bf(M, C), bf(M, 4 * C), bf(M, 4 * C), bf(M, C), bf(M, C), bf(M, C)
w_fc2T, w_fc1T, w_projT, w_attnT = bf(4 * C, C) * 0.05, bf(C, 4 * C) * 0.05, bf(C, C) * 0.05, bf(C, 3 * C) * 0.05
dh4, da, dyb, dqkv = bf(M, 4 * C), bf(M, C), bf(M, C), bf(M, 3 * C)
qkv = bf(M, 3 * C); lse = torch.randn(B, nh, T, device=dev)
cos, sin = ops.rope_tables(250, 16, device=dev)
x = torch.randn(M, C, device=dev); mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev); gam = torch.ones(C, device=dev)
dxf = torch.randn(M, C, device=dev)
G = [torch.zeros(C, 4 * C, device=dev), torch.zeros(4 * C, C, device=dev), torch.zeros(C, C, device=dev), torch.zeros(3 * C, C, device=dev)]
Gb = [torch.zeros(C, device=dev), torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.zeros(3 * C, device=dev)]
side = torch.cuda.Stream()

def chain():
    ops.gemm_nt(dx16, w_fc2T, None, ops.EPI_DGELU, aux_in=hpre, out=dh4)
    ops.gemm_nt(dh4, w_fc1T, None, ops.EPI_BF16, out=da)
    ops.layernorm_bwd(da, x, mean, rstd, gam, dres=dxf)
    ops.gemm_nt(dx16, w_projT, None, ops.EPI_BF16, out=dyb)
    ops.attn_bwd(qkv, y, dyb, lse, B, T, nh, cos, sin)
    ops.gemm_nt(dqkv, w_attnT, None, ops.EPI_BF16, out=da)
    ops.layernorm_bwd(da, x, mean, rstd, gam, dres=dxf)

def wgrads():
    ops.wgrad(dx16, g, G[0], Gb[0])
    ops.wgrad(dh4, a2, G[1], Gb[1])
    ops.wgrad(dx16, y, G[2], Gb[2])
    ops.wgrad(dqkv, a1, G[3], Gb[3])

def serial():
    chain(); wgrads()

def overlapped():
    ev = torch.cuda.Event(); ev.record()
    chain()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        wgrads()
    torch.cuda.current_stream().wait_stream(side)

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

tc, tw = timeit(chain), timeit(wgrads)
print(f"chain {tc:.0f} us  wgrads {tw:.0f} us  serial {timeit(serial):.0f} us  overlapped {timeit(overlapped):.0f} us")
