"""The MLP half of a block as one launch (csrc/mlp64.hip, coati_mlp_fwd): out = x + c_proj(NewGELU(c_fc(ln_2(x)))) and the tensors the
backward reads (ln_2(x) bf16, mean / rstd, NewGELU output bf16, NewGELU' codes), against fp32 torch on the operands the kernel rounds
and against the two launches it replaces (basic_transformer.py:12-28, 103-123, 165-169)."""
import math

import pytest
import torch

from tests.gpu_util import check, log

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_experimental_build():
    """the kernel under test lives in csrc/experimental/ and is only compiled under COATI_AMD_EXPERIMENTAL=1 (build.py)"""
    from coati_amd import _lib
    if not _lib.has_experimental():
        pytest.skip("csrc/experimental/ is not in this library: build and run with COATI_AMD_EXPERIMENTAL=1")
DEV = "cuda:0"


def rbf(t):
    return t.bfloat16().float()


def new_gelu(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("M", [24577, 50000, 65536, 49217])
def test_mlp_fwd_vs_torch_and_two_launches(M):
    from coati_amd import ops
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, 256, generator=g) * 1.5 + 0.3).to(DEV)
    gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(256, generator=g)).to(DEV)
    W1 = rbf(torch.randn(1024, 256, generator=g) * 0.06).to(DEV)
    b1 = (0.1 * torch.randn(1024, generator=g)).to(DEV)
    W2 = rbf(torch.randn(256, 1024, generator=g) * 0.03).to(DEV)
    b2 = (0.1 * torch.randn(256, generator=g)).to(DEV)
    out, a2, mean, rstd, gg, codes = ops.mlp_fwd(x, gamma, beta, W1.bfloat16(), b1, W2.bfloat16(), b2)
    torch.cuda.synchronize()
    # fp32 torch, rounding where the kernel rounds (a2 and g are bf16 operands of the two products)
    mu = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    a_ref = (x - mu) * torch.rsqrt(var + 1e-5) * gamma + beta
    h = rbf(a_ref) @ W1.t() + b1
    g_ref = new_gelu(h)
    hv = h.detach().clone().requires_grad_(True)
    new_gelu(hv).sum().backward()
    out_ref = x + rbf(g_ref) @ W2.t() + b2
    check(f"mlp64 M={M} ln_2 output", a2.float(), a_ref, 8e-3)
    check(f"mlp64 M={M} mean", mean, mu.squeeze(-1), 2e-6)
    check(f"mlp64 M={M} rstd", rstd, torch.rsqrt(var + 1e-5).squeeze(-1), 2e-5)
    check(f"mlp64 M={M} NewGELU output", gg.float(), g_ref, 8e-3)
    check(f"mlp64 M={M} NewGELU' codes", ops.dq8(codes), hv.grad, 4e-3)
    check(f"mlp64 M={M} out", out, out_ref, 2e-3)
    # the two launches it replaces (same operands, same roundings; the products' summation orders differ)
    h2, x8 = ops.gemm_nt(rbf(a_ref).bfloat16(), W1.bfloat16(), b1, ops.EPI_GELU_GRAD)
    out2 = ops.gemm_nt(h2, W2.bfloat16(), b2, ops.EPI_RES_F32, aux_in=x)
    check(f"mlp64 M={M} out vs two launches", out, out2, 2e-3)
    log(f"mlp64 M={M}: codes differing from the two-launch path in {(x8 != codes).float().mean().item():.2e} of the elements")
    assert (x8.int() - codes.int()).abs().max().item() <= 1
