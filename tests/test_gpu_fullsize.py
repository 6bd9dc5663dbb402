"""BASELINE.json full-size configuration (grande_closed, B=1024, T=80, A=16, V=10322) through size-independent
properties -- the oracle cannot run this size in seconds:
  * batch-row permutation equivariance of forward_dist and invariance of both losses;
  * the analytic gradient of the whole step against a central finite difference of the engine's own loss along the
    gradient direction (checks the complete backward at full size: <g, d> = dL/d eps);
  * a small move against the gradient lowers the loss; the optimiser's clip-norm equals the gradient buffer's norm."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.gpu_util import log  # noqa: E402

DEV = "cuda:0"
GRANDE = dict(n_layer_e3gnn=5, n_layer_xformer=16, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256,
              n_head=16, n_seq=250, n_tok=10322)


@pytest.fixture(scope="module", params=["padded", "packed"])
def big(request):
    """both row layouts: "packed" (batch["rows"]: the transformer passes on the rows' real prefixes) is what bench.py times"""
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    eng = Engine(ModelConfig(**GRANDE), DEV)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for name, (off, shape) in eng.layout.items():
            v = eng.view(name)
            if len(shape) == 2:
                v.copy_((torch.randn(shape, generator=g) * (0.03 if "tok_emb" not in name else 1.0)).to(DEV))
            elif (".ln_" in name and name.endswith("weight")) or name.endswith("clip.0.weight"):
                v.fill_(1.0)
            else:
                v.copy_((0.01 * torch.randn(shape, generator=g)).to(DEV))
    eng.refresh_shadows()
    batch, up = make_batch(1024, 80, 16, GRANDE["n_tok"], seed=77, with_rows=request.param == "packed")
    log(f"fullsize fixture: {request.param} layout, rows {batch['rows'].tolist() if 'rows' in batch else [1024 * 78, 1024 * 80]}")
    return eng, {k: (v.to(DEV) if k != "rows" else v) for k, v in batch.items()}, up.to(DEV)


def total_loss(eng, batch, up):
    h_e, h_s, bad = eng.forward(batch["raw_tokens"], batch["tokens"], batch["atoms"], batch["coords"], up,
                                y_next=batch["y_next"], train=False, rows=batch.get("rows"))
    eng.infonce(h_s, h_e, h_s, h_e, bad, row0=0, gscale=1.0)
    return eng.losses(), h_e.clone(), h_s.clone()


def test_batch_permutation(big):
    eng, batch, up = big
    L0, he0, hs0 = total_loss(eng, batch, up)
    perm = torch.randperm(1024, generator=torch.Generator().manual_seed(1)).to(DEV)
    pb = {k: (v[perm].contiguous() if k != "rows" else v) for k, v in batch.items()}   # (the packed-row COUNTS are sums over the batch)
    L1, he1, hs1 = total_loss(eng, pb, up[perm].contiguous())
    assert torch.equal(he1, he0[perm]) and torch.equal(hs1, hs0[perm])          # per-molecule maths is row-independent
    assert abs(L1["ar_loss"] - L0["ar_loss"]) < 2e-5 * abs(L0["ar_loss"])        # sums re-associate (fp32 atomics)
    assert abs(L1["clip_loss"] - L0["clip_loss"]) < 2e-5 * abs(L0["clip_loss"])
    log(f"fullsize permutation: ar {L0['ar_loss']:.6f}/{L1['ar_loss']:.6f} clip {L0['clip_loss']:.6f}/{L1['clip_loss']:.6f}")


def test_gradient_matches_finite_difference_and_step_descends(big):
    eng, batch, up = big
    p0 = eng.params.clone()
    eng.train_step(batch, up, lr=0.0, optimizer=False)
    L = eng.losses()
    g = eng.grads.clone()
    assert torch.isfinite(g).all()
    gn = float(g.double().norm())
    assert gn > 0 and math.isfinite(gn)
    d = g / gn
    # five-point central difference: the h^2 curvature term cancels, so the step can be four times the two-point one (h moves the loss by
    # ~ 1 each way, 2 h by ~ 2) and the bf16 noise of the forward (~ 1.5e-2 on the loss) weighs a quarter as much
    h = 1.0 / gn
    f = {}
    for k in (+1, -1, +2, -2):
        eng.params.copy_(p0 + k * h * d)
        eng.refresh_shadows()
        f[k] = total_loss(eng, batch, up)[0]["loss"]
    fd2 = (f[1] - f[-1]) / (2 * h)
    fd = (8 * (f[1] - f[-1]) - (f[2] - f[-2])) / (12 * h)
    log(f"fullsize directional derivative: analytic {gn:.4f}  finite-difference {fd:.4f} (five-point; two-point at the same h: {fd2:.4f})  loss {L['loss']:.4f}")
    assert abs(fd - gn) < 0.03 * gn, (fd, fd2, gn)
    vals = [f[1], f[-1]]
    assert vals[1] < L["loss"] < vals[0]       # a small move against / along the gradient lowers / raises the loss
    # the optimiser kernel's pre-clip norm equals the norm of the gradient buffer (lr = 0: parameters unchanged)
    eng.params.copy_(p0)
    eng.refresh_shadows()
    eng.adam_m.zero_(); eng.adam_v.zero_(); eng.step_count = 0
    eng.train_step(batch, up, lr=0.0)
    Ls = eng.losses()
    log(f"fullsize step: loss {Ls['loss']:.4f}, grad norm {Ls['grad_norm']:.3f}")
    assert abs(Ls["grad_norm"] - gn) < 1e-3 * gn
    assert torch.equal(eng.params, p0)
    eng.adam_m.zero_(); eng.adam_v.zero_(); eng.step_count = 0


FAMILIES = [
    ("tok_emb", lambda n: "tok_emb" in n),
    ("ln_1", lambda n: ".ln_1." in n),
    ("c_attn", lambda n: ".attn.c_attn." in n),
    ("c_proj", lambda n: ".attn.c_proj." in n),
    ("ln_2", lambda n: ".ln_2." in n),
    ("mlp fc", lambda n: ".mlpf.0." in n),
    ("mlp proj", lambda n: ".mlpf.2." in n),
    ("ln_f", lambda n: ".ln_f." in n),
    ("lm_head", lambda n: "lm_head" in n),
    ("gnn edge_mlp", lambda n: ".edge_mlp." in n),
    ("gnn node_mlp", lambda n: ".node_mlp." in n),
    ("gnn embedding + node_dec", lambda n: "point_encoder.embedding" in n or "point_encoder.node_dec" in n),
    ("point_to_clip", lambda n: n.startswith("point_to_clip")),
    ("smiles_to_clip", lambda n: n.startswith("smiles_to_clip")),
    ("special-token map", lambda n: n.startswith("point_clip_to_special_tokens")),
]


def test_gradient_of_every_parameter_family_matches_finite_differences(big):
    """The finite difference along g itself cannot see a gradient block that is never written (zeros contribute nothing to <g, g>).
    Here the direction is a random +-1 pattern RESTRICTED to one parameter family (independent of g), scaled to 2 % of the family's rms:
    <g, d> must reproduce the central difference of the engine's own loss; a block of zeros, a lost factor or a wrong sign in one
    kernel family fails its row.  Measured: the two columns agree to ~ 2e-3 of a loss unit; a family whose whole directional change
    stays below 0.015 (lm_head and the special-token map: a random sign pattern over millions of weights has almost no component along
    their gradient) is held to that absolute agreement and to a NON-ZERO gradient instead."""
    eng, batch, up = big
    p0 = eng.params.clone()
    eng.train_step(batch, up, lr=0.0, optimizer=False)
    g = eng.grads.clone()
    gen = torch.Generator().manual_seed(123)
    rows, judged = [], 0
    for fam, pred in FAMILIES:
        d = torch.zeros_like(p0)
        for name, (off, shape) in eng.layout.items():
            if pred(name) and "coord_mlp" not in name:
                n = int(torch.tensor(shape).prod())
                w = p0[off:off + n]
                rms = float(w.double().pow(2).mean().sqrt())
                sgn = (torch.randint(0, 2, (n,), generator=gen).float() * 2 - 1).to(DEV)
                d[off:off + n] = sgn * (0.02 * rms if rms > 0 else 2e-4)
        assert float(d.abs().max()) > 0, fam
        ana = float((g.double() * d.double()).sum())
        vals = []
        for sgn in (+1.0, -1.0):
            eng.params.copy_(p0 + sgn * d)
            eng.refresh_shadows()
            vals.append(total_loss(eng, batch, up)[0]["loss"])
        fd = (vals[0] - vals[1]) / 2.0
        rows.append((fam, ana, fd))
        gmax = max(float(g[off:off + int(torch.tensor(shape).prod())].abs().max()) for name, (off, shape) in eng.layout.items() if pred(name) and "coord_mlp" not in name)
        assert gmax > 0.0, f"the gradient of family {fam!r} is all zeros"
        if abs(fd) >= 0.015:
            judged += 1
            assert abs(ana - fd) <= 0.10 * abs(fd) + 0.004, (fam, ana, fd)
        else:
            assert abs(ana - fd) <= 0.004, (fam, ana, fd)       # small either way: the two must still agree in absolute terms
    eng.params.copy_(p0)
    eng.refresh_shadows()
    log("fullsize per-family directional derivatives (analytic <g, d> | central difference): " + "; ".join(f"{f}: {a:+.3f} | {b:+.3f}" for f, a, b in rows))
    assert judged >= 9, rows


def test_ragged_full_width_step_vs_oracle():
    """Full-width model (d=256, 16 heads, V not a multiple of anything) at sizes where NOTHING is a multiple of a tile:
    897 molecules x 83 tokens (74 451 rows: the row-block GEMMs run with a ragged last workgroup), 13-atom clouds (generic
    GNN kernels), vocabulary 997.  One layer each so the oracle's CPU pass stays short."""
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=1, n_layer_xformer=1, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256, n_head=16,
              n_seq=250, n_tok=997)
    ocfg = O.OracleConfig(**kw)
    P = O.init_params(ocfg, seed=3)
    eng = Engine(ModelConfig(**kw), DEV)
    eng.load_state_dict(P)
    batch, up = make_batch(897, 83, 13, 997, seed=897, n_special=12, p_bad=0.02, min_len=10)
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    L = eng.losses()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.sim_bf16():
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    assert abs(L["ar_loss"] - float(ar.detach())) < 2e-3 * abs(float(ar.detach()))
    assert abs(L["clip_loss"] - float(cl.detach())) < 2e-3 * abs(float(cl.detach()))
    g = eng.named_views("grads")
    worst = []
    for k in eng.layout:
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
        sc = float(ref.abs().max())
        if sc > 0:
            worst.append((float((g[k].cpu() - ref).abs().max()) / sc, k))
    worst.sort(reverse=True)
    log(f"ragged full-width step: worst gradient deviations {worst[:3]}")
    assert worst[0][0] < 1.7e-2, worst[:5]     # measured 8.3e-3


COATI2 = dict(n_layer_e3gnn=5, n_layer_xformer=12, n_hidden_xformer=512, n_hidden_e3nn=512, n_embd_common=512,
              n_head=16, n_seq=250, n_tok=4266)


@pytest.mark.parametrize("fp8", [False, True])
def test_coati2_full_shape_properties(fp8):
    """BASELINE.json configs[4] at its FULL shape -- d = 512, 12 layers, 16 heads of 32, V = 4266, batch 2048, packed rows, bf16 and
    MXFP8 operands -- through the size-independent properties (no reference code exists for this configuration: parity unpinned, the
    properties hold the engine to itself): batch-row permutation (bitwise on the embeddings; the packed row COUNTS are sums over the
    batch), the analytic gradient against a five-point central difference of the engine's own loss along the gradient, and a
    non-zero gradient in every parameter family."""
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    eng = Engine(ModelConfig(fp8=fp8, **COATI2), DEV)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, (off, shape) in eng.layout.items():
            v = eng.view(name)
            if len(shape) == 2:
                v.copy_((torch.randn(shape, generator=g) * (0.02 if "tok_emb" not in name else 1.0)).to(DEV))
            elif (".ln_" in name and name.endswith("weight")) or name.endswith("clip.0.weight"):
                v.fill_(1.0)
            else:
                v.copy_((0.01 * torch.randn(shape, generator=g)).to(DEV))
    eng.refresh_shadows()
    batch, up = make_batch(2048, 80, 16, COATI2["n_tok"], seed=78, n_special=330, with_rows=True)
    batch = {k: (v.to(DEV) if k != "rows" else v) for k, v in batch.items()}
    up = up.to(DEV)
    tag = "coati2 full shape " + ("fp8" if fp8 else "bf16")
    # permutation
    L0, he0, hs0 = total_loss(eng, batch, up)
    perm = torch.randperm(2048, generator=torch.Generator().manual_seed(2)).to(DEV)
    pb = {k: (v[perm].contiguous() if k != "rows" else v) for k, v in batch.items()}
    L1, he1, hs1 = total_loss(eng, pb, up[perm].contiguous())
    assert torch.equal(he1, he0[perm]) and torch.equal(hs1, hs0[perm])
    assert abs(L1["ar_loss"] - L0["ar_loss"]) < 2e-5 * abs(L0["ar_loss"]) and abs(L1["clip_loss"] - L0["clip_loss"]) < 2e-5 * abs(L0["clip_loss"])
    # gradient vs finite difference along g
    p0 = eng.params.clone()
    eng.train_step(batch, up, lr=0.0, optimizer=False)
    L = eng.losses()
    gr = eng.grads.clone()
    assert torch.isfinite(gr).all()
    gn = float(gr.double().norm())
    for fam, pred in FAMILIES:
        tot = sum(float(gr[off:off + int(torch.tensor(shape).prod())].abs().max()) for name, (off, shape) in eng.layout.items() if pred(name) and "coord_mlp" not in name)
        assert tot > 0.0, f"{tag}: the gradient of family {fam!r} is all zeros"
    d = gr / gn
    h = 1.0 / gn
    f = {}
    for k in (+1, -1, +2, -2):
        eng.params.copy_(p0 + k * h * d)
        eng.refresh_shadows()
        f[k] = total_loss(eng, batch, up)[0]["loss"]
    fd = (8 * (f[1] - f[-1]) - (f[2] - f[-2])) / (12 * h)
    log(f"{tag}: loss {L['loss']:.4f}, permutation ok, directional derivative analytic {gn:.4f} finite-difference {fd:.4f}")
    assert abs(fd - gn) < (0.10 if fp8 else 0.04) * gn, (fd, gn)       # fp8: weight gradients stay bf16, the forward is quantised: more noise on the loss
    assert f[-1] < L["loss"] < f[1]
