"""Chained MLP kernel (gemm_mlp.hip) against the two-launch paths it replaces, at the grande step's shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from coati_amd import ops, _lib
from coati_amd.ops import ptr, stream
from gemm_bench_util import timeit, row

dev = "cuda:0"
M, C, Hd = int(os.environ.get("M", 81920)), 256, 1024
torch.manual_seed(0)
x = torch.randn(M, C, device=dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
W1 = (torch.randn(Hd, C, device=dev) * 0.05).bfloat16(); b1 = torch.randn(Hd, device=dev) * 0.1
W2 = (torch.randn(C, Hd, device=dev) * 0.05).bfloat16(); b2 = torch.randn(C, device=dev) * 0.1
W2T = W2.t().contiguous(); W1T = W1.t().contiguous()
a = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
g = torch.empty(M, Hd, device=dev, dtype=torch.bfloat16); dg = torch.empty(M, Hd, device=dev, dtype=torch.uint8); dh = torch.empty_like(g)
W1p = ops.mlp_permute_w1(W1)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
out = torch.empty(M, C, device=dev); dA = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
dY = torch.randn(M, C, device=dev).bfloat16()
fl = 2.0 * 2 * M * C * Hd


def chain_fwd():
    _lib.call("coati_mlp_fwd", ptr(x), C, ptr(gamma), ptr(beta), ptr(W1p), C, ptr(b1), ptr(W2), Hd, ptr(b2), M, C, Hd, ptr(a), C,
              ptr(mean), ptr(rstd), ptr(g), ptr(dg), Hd, ptr(out), C, stream())


def pair_fwd():
    _lib.call("coati_mlp_fwd_paired", ptr(x), C, ptr(gamma), ptr(beta), ptr(W1), C, ptr(b1), ptr(W2), Hd, ptr(b2), M, C, Hd, ptr(a), C,
              ptr(mean), ptr(rstd), ptr(g), ptr(dg), Hd, ptr(out), C, stream())


def chain_bwd():
    _lib.call("coati_mlp_dgrad", ptr(dY), C, ptr(W2T), C, ptr(W1T), Hd, ptr(dg), M, C, Hd, ptr(dh), Hd, ptr(dA), C, stream())


y16 = ops.layernorm_fwd(x, gamma, beta)[0]


def split_fwd():     # (LayerNorm unfused here; the engine's row-block FC1 fuses it)
    gg, dd = ops.gemm_nt(y16, W1, b1, ops.EPI_GELU_GRAD)
    ops.gemm_nt(gg, W2, b2, ops.EPI_RES_F32, aux_in=x, out=out)


def split_bwd():
    d4 = ops.gemm_nt(dY, W2T, None, ops.EPI_MUL_AUX, aux_in=dg)     # dg: 8-bit codes (written by chain_fwd / split_fwd)
    ops.gemm_nt(d4, W1T, None, ops.EPI_BF16, out=dA)


row("mlp chain fwd (x, a, g, dg u8, out: 4608 B/row)", timeit(chain_fwd), fl, M * 4608.0)
row("mlp paired-wave fwd (same 4608 B/row)", timeit(pair_fwd), fl, M * 4608.0)
row("mlp chain dgrad (dY, dg u8, dh, dA: 4096 B/row)", timeit(chain_bwd), fl, M * 4096.0)
row("two launches fwd (FC1+GELU', FC2+res)", timeit(split_fwd), fl, M * 7680.0)
row("two launches dgrad (FC2 dgrad x GELU', FC1 dgrad)", timeit(split_bwd), fl, M * 6144.0)
