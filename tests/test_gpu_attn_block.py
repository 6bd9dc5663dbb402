"""The attention half of a RotaryBlock as one launch (csrc/attn_block.hip) against a plain fp32 torch statement of
basic_transformer.py:126-154, 171-172 that rounds to bf16 where the kernel does (ln_1 output, q / k / v, y), and against the
three-launch path it replaces."""
import pytest
import torch

from gpu_util import check, log, rbf

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_experimental_build():
    """the kernel under test lives in csrc/experimental/ and is only compiled under COATI_AMD_EXPERIMENTAL=1 (build.py)"""
    from coati_amd import _lib
    if not _lib.has_experimental():
        pytest.skip("csrc/experimental/ is not in this library: build and run with COATI_AMD_EXPERIMENTAL=1")
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from coati_amd import ops as o
    return o


def test_lane_half_swap_semantics(ops):
    """v_permlane32_swap_b32 vdst, src0 exchanges vdst[32..63] with src0[0..31] (ab_regroup relies on it)"""
    from coati_amd import _lib
    out = torch.zeros(128, device=DEV, dtype=torch.int32)
    _lib.call("coati_ab_probe_swap", ops.ptr(out), ops.stream())
    o = out.cpu().tolist()
    lanes = list(range(64))
    # first result = the new vdst: lanes < 32 keep a (= lane), lanes >= 32 receive b (= 100 + lane) of lane - 32
    assert o[:64] == [l if l < 32 else 100 + (l - 32) for l in lanes], o[:64]
    # second result = the new src0: lanes < 32 receive a of lane + 32, lanes >= 32 keep b
    assert o[64:] == [l + 32 if l < 32 else 100 + l for l in lanes], o[64:]


def _rotate(x, cos, sin):
    # RotaryEmbedding.rotary_embed (basic_transformer.py:83-100): x [rows, nh, 16], cos / sin [rows, 16]
    x1, x2 = x[..., :8], x[..., 8:]
    rot = torch.cat((-x2, x1), dim=-1)
    return x * cos.unsqueeze(1) + rot * sin.unsqueeze(1)


def _reference(x, ln_g, ln_b, Wqkv, bqkv, Wproj, bproj, cos, sin, lens, t_of_row):
    C, nh, hs = 256, 16, 16
    a1 = rbf(torch.nn.functional.layer_norm(x, (C,), ln_g, ln_b, 1e-5))
    mean = x.mean(1)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
    qkv = a1 @ rbf(Wqkv).t() + bqkv
    q, k, v = qkv.split(C, dim=1)
    c, s = cos[t_of_row], sin[t_of_row]
    q = rbf(_rotate(q.reshape(-1, nh, hs), c, s)).reshape(-1, C)
    k = rbf(_rotate(k.reshape(-1, nh, hs), c, s)).reshape(-1, C)
    v = rbf(v)
    y = torch.zeros_like(q)
    lse = []
    o = 0
    for n in lens.tolist():
        qs, ks, vs = (t[o:o + n].reshape(n, nh, hs).transpose(0, 1) for t in (q, k, v))
        att = (qs @ ks.transpose(1, 2)) * 0.25
        att = att.masked_fill(torch.triu(torch.ones(n, n, dtype=torch.bool), 1), float("-inf"))
        lse.append(torch.logsumexp(att, dim=-1))   # [nh, n]
        y[o:o + n] = (torch.softmax(att, dim=-1) @ vs).transpose(0, 1).reshape(n, C)
        o += n
    y = rbf(y)
    xmid = x + y @ rbf(Wproj).t() + bproj
    return a1, mean, rstd, torch.cat([q, k, v], 1), y, lse, xmid


@pytest.mark.parametrize("B,T,packed,seed", [(7, 80, True, 1), (40, 80, True, 2), (3, 128, True, 3), (5, 48, False, 4), (700, 80, True, 5), (64, 17, True, 6)])
def test_attn_block_fwd_vs_torch(ops, B, T, packed, seed):
    g = torch.Generator().manual_seed(seed)
    C = 256
    if packed:
        lens = torch.randint(1, T + 1, (B,), generator=g)
        lens[0] = T
        if B > 2:
            lens[1] = 1
            lens[2] = min(32, T)
    else:
        lens = torch.full((B,), T)
    M = int(lens.sum())
    keep = torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)
    src = keep.view(-1).nonzero().squeeze(1)            # slot b * T + t of every row
    t_of_row = src % T
    x = torch.randn(M, C, generator=g) * 1.5 + 0.1 * torch.randn(M, 1, generator=g)
    ln_g = 1.0 + 0.1 * torch.randn(C, generator=g)
    ln_b = 0.1 * torch.randn(C, generator=g)
    Wqkv = torch.randn(3 * C, C, generator=g) * 0.08
    bqkv = 0.1 * torch.randn(3 * C, generator=g)
    Wproj = torch.randn(C, C, generator=g) * 0.06
    bproj = 0.1 * torch.randn(C, generator=g)
    cos, sin = ops.rope_tables(250, 16, device="cpu")
    a1_r, mean_r, rstd_r, qkv_r, y_r, lse_r, xmid_r = _reference(x, ln_g, ln_b, Wqkv, bqkv, Wproj, bproj, cos, sin, lens, t_of_row)

    off = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)]).to(DEV, torch.int32)
    d = lambda t: t.to(DEV).contiguous()
    xmid, a1, mean, rstd, qkv, y, lse, grp = ops.attn_block_fwd(
        d(x), d(ln_g), d(ln_b), d(Wqkv).bfloat16(), d(bqkv), d(Wproj).bfloat16(), d(bproj), d(cos), d(sin), B, T,
        off=off if packed else None, row_src=d(src).to(torch.int32) if packed else None)
    torch.cuda.synchronize()
    # the work list: whole sequences, <= 128 rows per group, every row covered once
    gl = grp.cpu().tolist()
    ng = gl[0]
    bounds = gl[1:2 + ng]
    assert bounds[0] == 0 and bounds[-1] == M, (bounds[:4], bounds[-1], M)
    offs = set(off.cpu().tolist())
    assert all(b in offs for b in bounds) and all(0 < b1 - b0 <= 128 for b0, b1 in zip(bounds, bounds[1:]))
    tag = f"attn_block B{B} T{T} {'packed' if packed else 'padded'}"
    log(f"{tag}: {M} rows in {ng} groups")
    check(f"{tag} mean", mean, mean_r, 2e-5)
    check(f"{tag} rstd", rstd, rstd_r, 2e-5)
    check(f"{tag} a1", a1.float(), a1_r, 8e-3)
    check(f"{tag} q", qkv[:, :C].float(), qkv_r[:, :C], 1.2e-2)
    check(f"{tag} k", qkv[:, C:2 * C].float(), qkv_r[:, C:2 * C], 1.2e-2)
    check(f"{tag} v", qkv[:, 2 * C:].float(), qkv_r[:, 2 * C:], 1.2e-2)
    check(f"{tag} y", y.float(), y_r, 2e-2)
    o = 0
    lse_c = lse.cpu()
    worst = 0.0
    for b, n in enumerate(lens.tolist()):
        worst = max(worst, float((lse_c[b, :, :n] - lse_r[b]).abs().max()))
    log(f"{tag} lse max abs err {worst:.3e}")
    assert worst < 3e-2
    check(f"{tag} xmid", xmid, xmid_r, 6e-3)


def test_attn_block_fwd_vs_three_launches(ops):
    """the same weights and rows through gemm_qkv_rope (padded layout) + attn_fwd + the residual product: what the engine ran before"""
    B, T, C = 33, 80, 256
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B * T, C, generator=g)
    ln_g = 1.0 + 0.1 * torch.randn(C, generator=g)
    ln_b = 0.1 * torch.randn(C, generator=g)
    Wqkv = (torch.randn(3 * C, C, generator=g) * 0.08)
    bqkv = 0.1 * torch.randn(3 * C, generator=g)
    Wproj = (torch.randn(C, C, generator=g) * 0.06)
    bproj = 0.1 * torch.randn(C, generator=g)
    cos, sin = ops.rope_tables(250, 16, device=DEV)
    d = lambda t: t.to(DEV).contiguous()
    xmid, a1, mean, rstd, qkv, y, lse, _ = ops.attn_block_fwd(d(x), d(ln_g), d(ln_b), d(Wqkv).bfloat16(), d(bqkv), d(Wproj).bfloat16(), d(bproj), cos, sin, B, T)
    a1_u = torch.nn.functional.layer_norm(d(x), (C,), d(ln_g), d(ln_b), 1e-5).bfloat16()
    qkv_u = ops.gemm_qkv_rope(a1_u, d(Wqkv).bfloat16(), d(bqkv), T, cos, sin, 16)
    y_u, lse_u = ops.attn_fwd(qkv_u, B, T, 16, 16)
    check("attn_block vs 3 launches: a1", a1.float(), a1_u.float(), 8e-3)
    check("attn_block vs 3 launches: qkv", qkv.float(), qkv_u.float(), 1.2e-2)
    check("attn_block vs 3 launches: y", y.float(), y_u.float(), 2e-2)
    check("attn_block vs 3 launches: lse", lse, lse_u, 5e-3)
    xmid_u = d(x) + y_u.float() @ d(Wproj).bfloat16().float().t() + d(bproj)
    check("attn_block vs 3 launches: xmid", xmid, xmid_u, 6e-3)


_STEP_SCRIPT = """
import sys, torch
sys.path.insert(0, {root!r})
from coati_amd.engine import Engine, ModelConfig
from coati_amd.synthetic import make_batch
kw = dict(n_layer_e3gnn=1, n_layer_xformer=3, n_hidden_xformer=256, n_hidden_e3nn=64, n_embd_common=256, n_head=16, n_seq=100, n_tok=600)
eng = Engine(ModelConfig(**kw), "cuda:0")
g = torch.Generator().manual_seed(11)
with torch.no_grad():
    for name, (off, shape) in eng.layout.items():
        v = eng.view(name)
        if len(shape) == 2:
            v.copy_((torch.randn(shape, generator=g) * (0.05 if "tok_emb" not in name else 1.0)).to("cuda:0"))
        elif (".ln_" in name and name.endswith("weight")) or name.endswith("clip.0.weight"):
            v.copy_((1.0 + 0.1 * torch.randn(shape, generator=g)).to("cuda:0"))
        else:
            v.copy_((0.02 * torch.randn(shape, generator=g)).to("cuda:0"))
eng.refresh_shadows()
b, up = make_batch(300, 82, 6, 600, seed=3, n_special=12, min_len=12, with_rows={packed})
db = {{k: (v.to("cuda:0") if k != "rows" else v) for k, v in b.items()}}
h_e, h_s, bad = eng.train_step(db, up.to("cuda:0"), lr=1e-3, optimizer=False)
torch.cuda.synchronize()
torch.save({{"losses": eng.losses(), "h_s": h_s.cpu(), "grads": {{k: v.cpu().clone() for k, v in eng.named_views("grads").items()}}}}, {out!r})
"""


@pytest.mark.parametrize("packed", [True, False])
def test_engine_step_with_the_fused_attention_half(tmp_path, packed):
    """COATI_ATTN_BLOCK=1 routes the attention half of every block through attn_block.hip; the same training step (3 layers, d = 256,
    300 molecules) in two fresh processes: embeddings, losses and every gradient agree to bf16 rounding"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, val in (("fused", "1"), ("three", "0")):
        out = str(tmp_path / f"{tag}.pt")
        e = dict(os.environ, COATI_ATTN_BLOCK=val)
        r = subprocess.run([sys.executable, "-c", _STEP_SCRIPT.format(root=root, out=out, packed=packed)], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = torch.load(out)
    a, b = res["fused"], res["three"]
    tag = "packed" if packed else "padded"
    check(f"fused attention half in the step [{tag}]: h_smiles", a["h_s"], b["h_s"], 1e-2)
    for k in ("ar_loss", "clip_loss"):
        assert abs(a["losses"][k] - b["losses"][k]) <= 2e-3 * abs(b["losses"][k]), (k, a["losses"], b["losses"])
    worst = sorted(((float((a["grads"][k] - b["grads"][k]).abs().max()) / max(float(b["grads"][k].abs().max()), 1e-30), k)
                    for k in b["grads"] if float(b["grads"][k].abs().max()) > 0), reverse=True)
    log(f"fused attention half vs three launches in the step [{tag}]: losses {a['losses']['ar_loss']:.6f}/{b['losses']['ar_loss']:.6f}, worst gradient deviations {worst[:3]}")
    assert worst[0][0] <= 3e-2, worst[:5]
