"""End-to-end parity of the HIP engine against the oracle on the reference-generated golden model:
forward_dist outputs, both losses, every parameter gradient, clip-norm and AdamW, and a 3-step loss curve."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.gpu_util import check, log  # noqa: E402

DEV = "cuda:0"
# Stated tolerances = at most 2x the worst error measured on the MI355X (round 2, gpurun_out/test_report.txt).
# fp tolerance of the bf16-operand HIP path against the fp32 reference path (relative to tensor scale):
TOL_FWD = 6.5e-3       # measured 3.03e-3 (logits), 2.65e-3 (h_e3gnn), 1.39e-3 (h_smiles)
TOL_GRAD = 3.8e-2      # measured 1.88e-2 (worst parameter gradient of the golden step)
TOL_LOSS = 1e-3        # measured 4.8e-4 (clip), 1.5e-4 (ar)
# against the oracle with bf16 storage simulated at the same points
TOL_FWD_SIM = 6.5e-3   # measured 3.16e-3
TOL_GRAD_SIM = 4e-2    # measured 2.67e-2 (golden), 2.55e-2 (tall), 1.66e-2 (medium)
TOL_LOSS_SIM = 1.2e-3  # measured 5.4e-4 (tall clip); the edge-shape cases keep 5e-3 (2.8e-3 measured on a 1-valid-row batch)
TOL_GRADNORM = 1.3e-2
TOL_CURVE = 2.5e-3     # 40-step curve: measured 2.2e-3 (loss), 1.2e-3 (clip)
# Gradient NORM of the 40-step toy curve: this model's gradient norm (5 .. 45, lr 5e-4 on 64-wide layers) is sensitive to operand
# rounding -- the ORACLE with bf16 storage simulated (CPU, shares no kernel with the engine) deviates from the reference's
# curve by up to 2.3e-2 (step 28; 1.8e-2 at step 32, 2.0e-2 at step 12; median 3.6e-3), and the fp32 oracle started from weights
# perturbed by 1e-3 by up to 8.6e-3 (tools/curve_bf16_sim.py, profiles/r03_curve_bf16_sim.txt).  The engine measured 1.75e-2 at
# step 32 (round 2): inside that envelope.  Bound = 2x the simulation's worst step, plus a median bound that a systematic
# gradient error would break.
TOL_CURVE_GN_MAX = 4.6e-2
TOL_CURVE_GN_MEDIAN = 1.0e-2


@pytest.fixture(scope="module")
def setup(golden_dir):
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    z = np.load(os.path.join(golden_dir, "small_model.npz"))
    P = {k: torch.from_numpy(z[k]) for k in z.files}
    v = np.load(os.path.join(golden_dir, "small_vectors.npz"))
    vec = {k: torch.from_numpy(v[k]) for k in v.files}
    ocfg = O.OracleConfig(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64,
                          n_head=4, n_seq=24, n_tok=48)
    cfg = ModelConfig(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64,
                      n_head=4, n_seq=24, n_tok=48)
    eng = Engine(cfg, DEV)
    eng.load_state_dict(P)
    batch = {k: vec["b_" + k].to(DEV) for k in ("raw_tokens", "tokens", "atoms", "coords", "y_next")}
    return O, eng, P, vec, ocfg, batch


def test_layout_matches_reference_state_dict(setup):
    O, eng, P, vec, ocfg, batch = setup
    assert set(eng.layout) == set(P)
    for k, (off, shape) in eng.layout.items():
        assert tuple(P[k].shape) == tuple(shape), k


@pytest.mark.parametrize("tag,val", [("p0", True), ("p1", False)])
def test_forward_dist_vs_golden(setup, tag, val):
    O, eng, P, vec, ocfg, batch = setup
    B = batch["atoms"].shape[0]
    up = torch.full((B,), val, device=DEV)
    he, hs, bad = eng.forward(batch["raw_tokens"], batch["tokens"], batch["atoms"], batch["coords"], up, y_next=batch["y_next"])
    logits = eng.logits()
    check(f"golden {tag} h_e3gnn", he.cpu(), vec[f"fd_{tag}_h_e3gnn"], TOL_FWD)
    check(f"golden {tag} h_smiles", hs.cpu(), vec[f"fd_{tag}_h_smiles"], TOL_FWD)
    check(f"golden {tag} logits", logits.cpu(), vec[f"fd_{tag}_logits"], TOL_FWD)
    assert torch.equal(bad.cpu().bool(), vec[f"fd_{tag}_bad"])
    with O.sim_bf16():
        he_o, hs_o, lg_o, _ = O.forward_dist(P, ocfg, vec["b_raw_tokens"], vec["b_tokens"], vec["b_atoms"], vec["b_coords"], up.cpu())
    check(f"sim {tag} h_e3gnn", he.cpu(), he_o, TOL_FWD_SIM)
    check(f"sim {tag} h_smiles", hs.cpu(), hs_o, TOL_FWD_SIM)
    check(f"sim {tag} logits", logits.cpu(), lg_o, TOL_FWD_SIM)


def test_mixed_injection_vs_golden(setup):
    O, eng, P, vec, ocfg, batch = setup
    up = vec["fd_mix_use_point"].to(DEV)
    eng.forward(batch["raw_tokens"], batch["tokens"], batch["atoms"], batch["coords"], up, y_next=batch["y_next"])
    check("golden mixed logits", eng.logits().cpu(), vec["fd_mix_logits"], TOL_FWD)


def test_step_losses_and_grads(setup, golden_dir):
    O, eng, P, vec, ocfg, batch = setup
    B = batch["atoms"].shape[0]
    up = torch.ones(B, dtype=torch.bool, device=DEV)
    eng.step_count = 0
    eng.load_state_dict(P)
    eng.train_step(batch, up, lr=5e-4, optimizer=False)
    L = eng.losses()
    log(f"losses {L}")
    check("step ar loss", torch.tensor([L["ar_loss"]]), vec["step_ar"].reshape(1), TOL_LOSS)
    check("step clip loss", torch.tensor([L["clip_loss"]]), vec["step_clip"].reshape(1), TOL_LOSS)
    check("step loss", torch.tensor([L["loss"]]), vec["step_loss"].reshape(1), TOL_LOSS)
    G = np.load(os.path.join(golden_dir, "small_step_grads.npz"))
    # oracle with bf16 storage simulation for the tight comparison
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.sim_bf16():
        loss, *_ = O.step_loss(Pg, ocfg, {k: vec["b_" + k] for k in ("raw_tokens", "tokens", "atoms", "coords", "y_next")}, up.cpu())
    loss.backward()
    grads = eng.named_views("grads")
    worst = []
    for k in sorted(eng.layout):
        g = grads[k].cpu()
        ref = torch.from_numpy(G["grad." + k])
        sim = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(ref)
        scale = max(float(ref.abs().max()), 1e-30)
        e_ref = float((g - ref).abs().max()) / scale if float(ref.abs().max()) > 0 else float(g.abs().max())
        e_sim = float((g - sim).abs().max()) / scale if float(ref.abs().max()) > 0 else float(g.abs().max())
        log(f"grad {k:60s} vs reference {e_ref:.3e}  vs sim-oracle {e_sim:.3e}  scale {scale:.3e}")
        worst.append((e_ref, e_sim, k))
    bad = [(a, b, k) for a, b, k in worst if a > TOL_GRAD or b > TOL_GRAD_SIM]
    assert not bad, f"gradient mismatch: {sorted(bad, reverse=True)[:8]}"
    assert all(float(grads[k].abs().max()) == 0.0 for k in grads if "coord_mlp" in k)
    # clip-norm of this step's gradients (coati_grad_sqnorm alone: optimizer=False above did not run it)
    from coati_amd import _lib
    from coati_amd.ops import ptr, stream
    part = torch.zeros(1024, device=DEV); out = torch.zeros(2, device=DEV)
    _lib.call("coati_grad_sqnorm", ptr(eng.grads), eng.n_params, ptr(part), 1024, ptr(out[0:1]), 10.0, ptr(out[1:2]), stream())
    check("grad norm of the step vs clip_grad_norm_", out[0:1].cpu(), vec["step_gradnorm"].reshape(1).float(), TOL_GRADNORM)


def test_three_step_loss_curve(setup, golden_dir):
    O, eng, P, vec, ocfg, batch = setup
    B = batch["atoms"].shape[0]
    up = torch.ones(B, dtype=torch.bool, device=DEV)
    eng.load_state_dict(P)
    eng.step_count = 0
    eng.adam_m.zero_(); eng.adam_v.zero_()
    losses, norms = [], []
    for _ in range(3):
        eng.train_step(batch, up, lr=5e-4)
        L = eng.losses()
        losses.append(L["loss"]); norms.append(L["grad_norm"])
    log(f"loss curve {losses} ref {vec['step_losses'].tolist()} gradnorm {norms} ref0 {float(vec['step_gradnorm'])}")
    check("loss curve", torch.tensor(losses), vec["step_losses"].float(), 3.5e-3)
    check("grad norm step0", torch.tensor([norms[0]]), vec["step_gradnorm"].reshape(1), TOL_GRADNORM)
    A3 = np.load(os.path.join(golden_dir, "small_model_after3.npz"))
    sd = eng.state_dict()
    # 3 AdamW steps at lr 5e-4 move each weight by <= ~1.5e-3.  Adam's first steps are sign-like (m/sqrt(v) ~ +-1), so an
    # element whose gradient is near zero can flip direction under bf16 noise; compare the DISPLACEMENT direction per tensor.
    for k in sorted(eng.layout):
        if "coord_mlp" in k:
            # no gradient ever reaches coord_mlp (the reference discards its output): torch's AdamW skips it -- no decay
            assert torch.equal(sd[k].cpu(), P[k]) and torch.equal(torch.from_numpy(A3[k]), P[k]), k
            continue
        d_hip = (sd[k].cpu() - P[k]).flatten().double()
        d_ref = (torch.from_numpy(A3[k]) - P[k]).flatten().double()
        cos = float((d_hip @ d_ref) / (d_hip.norm() * d_ref.norm() + 1e-30))
        log(f"adamw displacement {k:55s} cosine {cos:.4f}")
        assert cos > 0.9, (k, cos)


def test_forty_step_loss_curve_vs_reference(golden_dir):
    """north_star "loss-curve equivalent to reference": 40 optimiser steps (clip-norm 10, AdamW lr 5e-4, wd 0.1) cycling
    over eight different batches from the golden weights, against the curve the reference itself produced
    (tests/golden/loss_curve.npz): total loss, AR and InfoNCE parts and the gradient norm, step by step."""
    from coati_amd.engine import Engine, ModelConfig
    z = np.load(os.path.join(golden_dir, "small_model.npz"))
    c = np.load(os.path.join(golden_dir, "loss_curve.npz"))
    eng = Engine(ModelConfig(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64,
                             n_head=4, n_seq=24, n_tok=48), DEV)
    eng.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})
    batches = [{k: torch.from_numpy(c[f"b{i}_{k}"]).to(DEV) for k in ("raw_tokens", "tokens", "atoms", "coords", "y_next")} for i in range(8)]
    n = len(c["loss"])
    rec = dict(loss=[], ar=[], clip=[], gradnorm=[])
    for step in range(n):
        b = batches[step % 8]
        up = torch.ones(b["atoms"].shape[0], dtype=torch.bool, device=DEV)
        eng.train_step(b, up, lr=5e-4, weight_decay=0.1, max_norm=10.0)
        L = eng.losses()
        rec["loss"].append(L["loss"]); rec["ar"].append(L["ar_loss"]); rec["clip"].append(L["clip_loss"]); rec["gradnorm"].append(L["grad_norm"])
    dev_ = {k: (np.abs(np.array(v) - c[k]) / np.maximum(np.abs(c[k]), 1e-6)) for k, v in rec.items()}
    log("40-step curve: reference loss " + " ".join(f"{x:.3f}" for x in c["loss"][::4]))
    log("40-step curve: hip       loss " + " ".join(f"{x:.3f}" for x in rec["loss"][::4]))
    log("40-step curve: max relative deviation " + ", ".join(f"{k} {v.max():.3e} (step {int(v.argmax())})" for k, v in dev_.items()))
    assert c["loss"][-8:].mean() < 0.85 * c["loss"][:8].mean()              # the curve really descends
    assert dev_["loss"].max() <= TOL_CURVE and dev_["ar"].max() <= TOL_CURVE
    assert np.abs(np.array(rec["clip"]) - c["clip"]).max() <= TOL_CURVE * max(1.0, float(np.abs(c["clip"]).max()))
    log(f"40-step curve: grad-norm deviation median {np.median(dev_['gradnorm']):.3e}")
    assert dev_["gradnorm"].max() <= TOL_CURVE_GN_MAX and np.median(dev_["gradnorm"]) <= TOL_CURVE_GN_MEDIAN


def test_mid_curve_step_from_the_references_weights(golden_dir):
    """A mid-curve check that is NOT an envelope (tests/golden/loss_curve_mid.npz): the engine loads the weights the REFERENCE
    held at the start of step 21 of its 40-step curve and must reproduce that one step -- losses, clip-norm and every parameter
    gradient at the single-step tolerances -- wherever its own trajectory had drifted to by then."""
    from coati_amd.engine import Engine, ModelConfig
    m = np.load(os.path.join(golden_dir, "loss_curve_mid.npz"))
    c = np.load(os.path.join(golden_dir, "loss_curve.npz"))
    step = int(m["step"])
    eng = Engine(ModelConfig(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64,
                             n_head=4, n_seq=24, n_tok=48), DEV)
    eng.load_state_dict({k[2:]: torch.from_numpy(m[k]) for k in m.files if k.startswith("w.")})
    b = {k: torch.from_numpy(c[f"b{step % 8}_{k}"]).to(DEV) for k in ("raw_tokens", "tokens", "atoms", "coords", "y_next")}
    eng.train_step(b, torch.ones(b["atoms"].shape[0], dtype=torch.bool, device=DEV), lr=0.0, weight_decay=0.0, max_norm=10.0)
    L = eng.losses()
    check("mid-curve loss", torch.tensor([L["loss"]]), torch.from_numpy(m["loss"]).reshape(1), TOL_LOSS)
    check("mid-curve ar", torch.tensor([L["ar_loss"]]), torch.from_numpy(m["ar"]).reshape(1), TOL_LOSS)
    check("mid-curve clip", torch.tensor([L["clip_loss"]]), torch.from_numpy(m["clip"]).reshape(1), TOL_LOSS)
    check("mid-curve clip-norm", torch.tensor([L["grad_norm"]]), torch.from_numpy(m["gradnorm"]).reshape(1).float(), TOL_GRADNORM)
    grads = eng.named_views("grads")
    worst = []
    for k in sorted(eng.layout):
        ref = torch.from_numpy(m["g." + k])
        scale = float(ref.abs().max())
        g = grads[k].cpu()
        worst.append((float((g - ref).abs().max()) / scale if scale > 0 else float(g.abs().max()), k))
    worst.sort(reverse=True)
    log(f"mid-curve step (reference weights of step {step}): worst gradient deviations {worst[:3]}")
    assert worst[0][0] <= TOL_GRAD, worst[:6]


def test_medium_random_model_grads():
    """A wider config (C=H=128, 3+2 layers, V=300, T up to 40, A=12) with random weights, checked against the oracle
    in bf16-simulation mode: exercises partial tiles, several heads and the non-square GNN shapes."""
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=2, n_layer_xformer=3, n_hidden_xformer=128, n_hidden_e3nn=128, n_embd_common=128, n_head=8,
              n_seq=64, n_tok=300)
    ocfg = O.OracleConfig(**kw)
    P = O.init_params(ocfg, seed=3)
    eng = Engine(ModelConfig(**kw), DEV)
    eng.load_state_dict(P)
    batch, up = make_batch(24, 40, 12, 300, seed=5, n_special=12, p_bad=0.1, min_len=6)
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    L = eng.losses()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.sim_bf16():
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    log(f"medium losses hip {L} oracle ar {float(ar):.6f} clip {float(cl):.6f}")
    check("medium ar", torch.tensor([L["ar_loss"]]), ar.detach().reshape(1), TOL_LOSS_SIM)
    check("medium clip", torch.tensor([L["clip_loss"]]), cl.detach().reshape(1), TOL_LOSS_SIM)
    grads = eng.named_views("grads")
    bad = []
    for k in sorted(eng.layout):
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
        scale = max(float(ref.abs().max()), 1e-30)
        e = float((grads[k].cpu() - ref).abs().max()) / scale if float(ref.abs().max()) > 0 else float(grads[k].abs().max())
        log(f"medium grad {k:60s} relerr {e:.3e} scale {scale:.3e}")
        if e > TOL_GRAD_SIM:
            bad.append((e, k))
    assert not bad, sorted(bad, reverse=True)[:8]


@pytest.mark.parametrize("H", [128, 256], ids=["h128_three_launches", "h256_fused_edge_launch"])
def test_more_than_64_atoms_per_molecule_vs_oracle(H):
    """(H = 256 takes the fused forward edge launch, gemm_rb16.hip gnn_edge_fwd_fused_kernel: receivers with up to 71 edges = five 16-edge
    slabs summed in the wave's LDS strip.)  Point clouds wider than one wavefront: the reference pads every batch to its largest molecule (batch_pipe.py:18) and
    hydrogenates by default, so a drug-like molecule can exceed 64 atoms.  72-slot clouds packed inside the 5 A cutoff give
    receivers with up to 71 edges: the compacted-edge kernels walk such a segment in two 64-edge chunks."""
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=2, n_layer_xformer=1, n_hidden_xformer=64, n_hidden_e3nn=H, n_embd_common=64, n_head=4, n_seq=32, n_tok=64)
    ocfg = O.OracleConfig(**kw)
    P = O.init_params(ocfg, seed=21)
    eng = Engine(ModelConfig(**kw), DEV)
    eng.load_state_dict(P)
    batch, up = make_batch(10, 20, 72, 64, seed=23, n_special=12, p_bad=0.0, min_len=4)
    batch["coords"] = batch["coords"] * 0.8                      # std 1.2 A: nearly every pair inside the cutoff
    n_at = (batch["atoms"] > 0).sum(1)
    assert int(n_at.max()) > 66
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    L = eng.losses()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.sim_bf16():
        loss, ar, cl, (he, hs, _) = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    check("A=72 clip", torch.tensor([L["clip_loss"]]), cl.detach().reshape(1), 5e-3)
    check("A=72 ar", torch.tensor([L["ar_loss"]]), ar.detach().reshape(1), 5e-3)
    he_hip, _ = eng.encode(atoms=db["atoms"], coords=db["coords"])[::-1]
    check("A=72 h_e3gnn", he_hip.cpu(), he.detach(), TOL_FWD_SIM)
    grads = eng.named_views("grads")
    worst = []
    for k in sorted(eng.layout):
        if not k.startswith("point_encoder") or "coord_mlp" in k:
            continue
        ref = Pg[k].grad
        worst.append((float((grads[k].cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30), k))
    worst.sort(reverse=True)
    log(f"A=72 point-encoder gradients: worst {worst[:3]}")
    assert worst[0][0] <= TOL_GRAD_SIM, worst[:5]


def test_tall_closed_shape_vs_oracle():
    """BASELINE.json configs[0] shape (d=256, nh=16 -> head size 16, batch 64, seq_len 128, 16-atom clouds) at reduced
    depth (2 + 2 layers) and vocabulary so the oracle finishes in seconds: four 32-token attention blocks, the
    row-block K=256 GEMMs at a ragged M (not a multiple of 320 rows), lm_head tiles with a partial last tile."""
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256, n_head=16,
              n_seq=250, n_tok=1003)
    ocfg = O.OracleConfig(**kw)
    P = O.init_params(ocfg, seed=11)
    eng = Engine(ModelConfig(**kw), DEV)
    eng.load_state_dict(P)
    batch, up = make_batch(64, 128, 16, 1003, seed=7, n_special=12, p_bad=0.05, min_len=20)
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    L = eng.losses()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.sim_bf16():
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    log(f"tall losses hip {L} oracle ar {float(ar):.6f} clip {float(cl):.6f}")
    check("tall ar", torch.tensor([L["ar_loss"]]), ar.detach().reshape(1), TOL_LOSS_SIM)
    check("tall clip", torch.tensor([L["clip_loss"]]), cl.detach().reshape(1), TOL_LOSS_SIM)
    grads = eng.named_views("grads")
    bad = []
    for k in sorted(eng.layout):
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
        scale = max(float(ref.abs().max()), 1e-30)
        e = float((grads[k].cpu() - ref).abs().max()) / scale if float(ref.abs().max()) > 0 else float(grads[k].abs().max())
        log(f"tall grad {k:60s} relerr {e:.3e} scale {scale:.3e}")
        if e > TOL_GRAD_SIM:
            bad.append((e, k))
    assert not bad, sorted(bad, reverse=True)[:8]


@pytest.mark.parametrize("flags", [{}, {"residual": True}, {"torch_emb": True, "old_architecture": True}], ids=["grande_flags", "residual", "torch_emb+old_architecture"])
def test_wide_batch_kernels_vs_oracle(flags):
    """57 720 token rows (481 x 120) at d = 256, one layer each: the size gates of the big-batch kernels are all open here
    -- row-block GEMMs with LayerNorm fused into the operand load, the N = 256 ring GEMM, the LDS-DMA weight gradient, the
    fused attention backward at four 32-row blocks -- and the oracle still finishes in seconds.  The constructor flags that
    change the point encoder / the heads run here too: 7 696 atoms take the grouped point-encoder weight gradient
    (node_mlp.0.weight with rows 2H + 28 apart under residual = True)."""
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=1, n_layer_xformer=1, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256, n_head=16,
              n_seq=250, n_tok=600, **flags)
    ocfg = O.OracleConfig(**kw)
    P = O.init_params(ocfg, seed=13)
    eng = Engine(ModelConfig(**kw), DEV)
    eng.load_state_dict(P)
    batch, up = make_batch(481, 120, 16, 600, seed=9, n_special=12, p_bad=0.02, min_len=100)   # 57 720 rows: ragged last 32-row slab
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    L = eng.losses()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.sim_bf16():
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    log(f"wide losses hip {L} oracle ar {float(ar):.6f} clip {float(cl):.6f}")
    check("wide ar", torch.tensor([L["ar_loss"]]), ar.detach().reshape(1), TOL_LOSS_SIM)
    # (old_architecture: the heads end in a LayerNorm, clip_loss's raw dot products reach ~ 256 and the bf16 rounding of the embeddings
    #  shows on the loss: 1.04e-3 measured, see tests/test_gpu_flags.py)
    check("wide clip", torch.tensor([L["clip_loss"]]), cl.detach().reshape(1), 2.5e-3 if flags.get("old_architecture") else TOL_LOSS_SIM)
    grads = eng.named_views("grads")
    bad = []
    for k in sorted(eng.layout):
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
        scale = max(float(ref.abs().max()), 1e-30)
        e = float((grads[k].cpu() - ref).abs().max()) / scale if float(ref.abs().max()) > 0 else float(grads[k].abs().max())
        log(f"wide grad {k:60s} relerr {e:.3e} scale {scale:.3e}")
        if e > TOL_GRAD_SIM:
            bad.append((e, k))
    assert not bad, sorted(bad, reverse=True)[:8]


def test_row_split_products_vs_oracle():
    """100 200 token rows (835 x 120, padded layout) at d = 256, one layer: above 65 536 rows every row-wise product of the passes runs as
    TWO launches of the 16-row-slab / one-round ring kernels on equal row ranges (engine.cpp gemm_rows: LayerNorm fused into the operand
    load, RoPE from the sequence boundary, NewGELU' codes, lm_head partials / dlogits, the fused LayerNorm backward with its partial rows
    of both launches) -- against the oracle, losses and every gradient."""
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=1, n_layer_xformer=1, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256, n_head=16, n_seq=250, n_tok=600)
    ocfg = O.OracleConfig(**kw)
    P = O.init_params(ocfg, seed=14)
    eng = Engine(ModelConfig(**kw), DEV)
    eng.load_state_dict(P)
    batch, up = make_batch(835, 120, 16, 600, seed=10, n_special=12, p_bad=0.02, min_len=100)
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    L = eng.losses()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with O.sim_bf16():
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    log(f"row-split losses hip {L} oracle ar {float(ar):.6f} clip {float(cl):.6f}")
    check("row-split ar", torch.tensor([L["ar_loss"]]), ar.detach().reshape(1), TOL_LOSS_SIM)
    check("row-split clip", torch.tensor([L["clip_loss"]]), cl.detach().reshape(1), TOL_LOSS_SIM)
    grads = eng.named_views("grads")
    bad = []
    for k in sorted(eng.layout):
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
        scale = max(float(ref.abs().max()), 1e-30)
        e = float((grads[k].cpu() - ref).abs().max()) / scale if float(ref.abs().max()) > 0 else float(grads[k].abs().max())
        log(f"row-split grad {k:60s} relerr {e:.3e} scale {scale:.3e}")
        if e > TOL_GRAD_SIM:
            bad.append((e, k))
    assert not bad, sorted(bad, reverse=True)[:8]


def test_staged_backward_equals_whole_backward():
    """The multi-GPU step runs the backward in three stages (lm_head + decoder + heads | encoder (+ point encoder on the
    side stream) | remaining point-encoder work) so gradient buckets can be all-reduced as they complete: the staged
    gradients must equal the single-call backward (fp32 atomics: 1e-5 of tensor scale)."""
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=2, n_layer_xformer=3, n_hidden_xformer=128, n_hidden_e3nn=128, n_embd_common=128, n_head=8,
              n_seq=64, n_tok=300)
    eng = Engine(ModelConfig(**kw), DEV)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for name, (off, shape) in eng.layout.items():
            eng.view(name).copy_((torch.randn(shape, generator=g) * 0.05).to(DEV))
    eng.refresh_shadows()
    batch, up = make_batch(24, 40, 12, 300, seed=5, n_special=12, p_bad=0.1, min_len=6)
    db = {k: v.to(DEV) for k, v in batch.items()}
    up = up.to(DEV)
    eng.grads.zero_()
    eng.train_step(db, up, lr=1e-3, optimizer=False)
    whole = eng.grads.clone()
    eng.grads.zero_()
    h_e, h_s, bad = eng.forward(db["raw_tokens"], db["tokens"], db["atoms"], db["coords"], up, y_next=db["y_next"], train=True)
    dS, dC = eng.infonce(h_s, h_e, h_s, h_e, bad, row0=0, gscale=0.5 * eng.token_entropy_unit())
    for order in ((1, 2, 3), (1, 4, 5, 3)):   # 4, 5 = the encoder stage in two halves (what the data-parallel step runs)
        eng.grads.zero_()
        for stage in order:
            eng.backward(dS if stage == 1 else None, dC if stage == 1 else None, stage)
        torch.cuda.synchronize()
        for name, (off, shape) in eng.layout.items():
            n = int(np.prod(shape))
            a, b = whole[off:off + n], eng.grads[off:off + n]
            scale = max(float(a.abs().max()), 1e-20)
            assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-12, (order, name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("layout", ["padded", "packed"])
def test_coati2_width_d512_step_vs_oracle(layout):
    """BASELINE.json configs[4] at its real width: d = 512, 16 heads of size 32, vocabulary 4266 (coati2_12_12), E(3)-GNN at
    h = 512 -- the only pinned maths of that config is the transformer block at this width (reference
    coati/models/simple_coati2/transformer_only.py:43, same block as basic_transformer.py:157-174; the 3D encoder and the
    training step have no reference code: parity unpinned).  One layer each and 128 molecules x 80 tokens (10 240 rows) so
    that the oracle finishes in seconds while the K = 512 / N = 1536 / 2048 products, the grouped 256-wide weight gradient at
    K = 512 and 2048, the head-size-32 attention and the lm_head at V = 4266 all run at their real widths."""
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=1, n_layer_xformer=1, n_hidden_xformer=512, n_hidden_e3nn=512, n_embd_common=512, n_head=16,
              n_seq=250, n_tok=4266)
    ocfg = O.OracleConfig(**kw)
    P = O.init_params(ocfg, seed=512)
    eng = Engine(ModelConfig(**kw), DEV)
    eng.load_state_dict(P)
    batch, up = make_batch(128, 80, 16, 4266, seed=51, n_special=330, p_bad=0.03, min_len=16, with_rows=(layout == "packed"))
    db = {k: (v if k == "rows" else v.to(DEV)) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    assert eng._packed == (layout == "packed")
    L = eng.losses()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with O.sim_bf16():
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    check(f"d512 [{layout}] ar", torch.tensor([L["ar_loss"]]), ar.detach().reshape(1), TOL_LOSS_SIM)
    check(f"d512 [{layout}] clip", torch.tensor([L["clip_loss"]]), cl.detach().reshape(1), TOL_LOSS_SIM)
    grads = eng.named_views("grads")
    worst = []
    for k in sorted(eng.layout):
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
        if float(ref.abs().max()) > 0:
            worst.append((float((grads[k].cpu() - ref).abs().max()) / float(ref.abs().max()), k))
    worst.sort(reverse=True)
    log(f"d512 [{layout}] worst gradient deviations {worst[:4]}")
    assert worst[0][0] <= TOL_GRAD_SIM, worst[:6]


def test_head_size_32_model_grads_and_decode():
    """The COATI2-size transformer shape (n_embd / n_head = 32): a full training step against
    the oracle (pinned at head size 32 by tests/test_oracle_golden.py::test_block_head_size_32), and the KV-cached decode
    path against the engine's own full-sequence logits."""
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=1, n_layer_xformer=2, n_hidden_xformer=128, n_hidden_e3nn=128, n_embd_common=128, n_head=4,
              n_seq=64, n_tok=300)
    ocfg = O.OracleConfig(**kw)
    P = O.init_params(ocfg, seed=4)
    eng = Engine(ModelConfig(**kw), DEV)
    eng.load_state_dict(P)
    batch, up = make_batch(20, 40, 10, 300, seed=6, n_special=12, p_bad=0.1, min_len=6)
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    L = eng.losses()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.sim_bf16():
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    check("hs32 ar", torch.tensor([L["ar_loss"]]), ar.detach().reshape(1), TOL_LOSS_SIM)
    check("hs32 clip", torch.tensor([L["clip_loss"]]), cl.detach().reshape(1), TOL_LOSS_SIM)
    grads = eng.named_views("grads")
    bad = []
    for k in sorted(eng.layout):
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
        scale = max(float(ref.abs().max()), 1e-30)
        e = float((grads[k].cpu() - ref).abs().max()) / scale if float(ref.abs().max()) > 0 else float(grads[k].abs().max())
        log(f"hs32 grad {k:60s} relerr {e:.3e} scale {scale:.3e}")
        if e > TOL_GRAD_SIM:
            bad.append((e, k))
    assert not bad, sorted(bad, reverse=True)[:8]
    # decode: feed the decoder tokens position by position, compare with the full-sequence logits of the same engine
    eng.forward(db["raw_tokens"], db["tokens"], db["atoms"], db["coords"], up.to(DEV), y_next=None, train=False)
    full = eng.logits().clone()                      # [B, T2, V]
    B, T2 = db["tokens"].shape
    inj = torch.zeros(B, 128, device=DEV)            # rows with an [UNK] slot were injected with cliptok in the full pass
    has_unk = (db["tokens"] == 7).any(1)
    rows = (~has_unk).nonzero().flatten()            # compare rows without injection (their embeddings are table rows)
    assert rows.numel() > 0
    eng.decode_begin(B, T2)
    worst = 0.0
    for t in range(T2):
        lg = eng.decode_step(db["tokens"][:, t].contiguous(), inj)
        ref = full[rows, t]
        worst = max(worst, float((lg[rows] - ref).abs().max()) / max(float(ref.abs().max()), 1e-6))
    log(f"hs32 decode vs full forward: worst relative error {worst:.3e}")
    assert worst < 5e-3        # measured 2.3e-3


def test_two_rank_emulation_equals_global_batch():
    """SURVEY 8(e) equivalence: W = 2 ranks with per-rank batch B must reproduce a W = 1 run with batch 2B (same rows in
    rank-major order) for the InfoNCE value and the parameter gradients.  Only one GPU is available to the tests, so the
    two ranks are two engines in one process and the three collectives are done by hand exactly as
    coati_amd.distributed does them (all-gather = cat, reduce-scatter = sum of the rank's row block, all-reduce(AVG) =
    mean).  The AR targets are masked out (the distributed AR loss is a mean of per-rank means by design)."""
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    kw = dict(n_layer_e3gnn=1, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64, n_head=4,
              n_seq=40, n_tok=200)
    engs = [Engine(ModelConfig(**kw), DEV) for _ in range(3)]      # rank 0, rank 1, global
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for name, (off, shape) in engs[0].layout.items():
            v = (torch.randn(shape, generator=g) * 0.08).to(DEV)
            for e in engs:
                e.view(name).copy_(v)
    for e in engs:
        e.refresh_shadows()
        e.grads.zero_()
    B, W = 12, 2
    parts = []
    for r in range(W):
        b, up = make_batch(B, 24, 8, 200, seed=20 + r, n_special=12, p_bad=0.1, min_len=5)
        b["y_next"] = torch.full_like(b["y_next"], -1)
        parts.append(({k: v.to(DEV) for k, v in b.items()}, up.to(DEV)))
    assert parts[0][0]["tokens"].shape == parts[1][0]["tokens"].shape and parts[0][0]["raw_tokens"].shape == parts[1][0]["raw_tokens"].shape
    teu = engs[0].token_entropy_unit()
    # ---- two ranks
    outs = []
    for r in range(W):
        b, up = parts[r]
        outs.append(engs[r].forward(b["raw_tokens"], b["tokens"], b["atoms"], b["coords"], up, y_next=b["y_next"], train=True))
    s_all = torch.cat([o[1] for o in outs]); c_all = torch.cat([o[0] for o in outs]); bad_all = torch.cat([o[2] for o in outs])
    part_grads = []
    for r in range(W):
        part_grads.append(engs[r].infonce(outs[r][1], outs[r][0], s_all, c_all, bad_all, row0=r * B, gscale=0.5 * teu * W))
    for r in range(W):
        dS = sum(pg[0][r * B:(r + 1) * B] for pg in part_grads)
        dC = sum(pg[1][r * B:(r + 1) * B] for pg in part_grads)
        engs[r].backward(dS.contiguous(), dC.contiguous(), 0)
    avg = 0.5 * (engs[0].grads + engs[1].grads)
    clip_ranks = [e.losses()["clip_loss"] for e in engs[:2]]
    # ---- one rank, batch 2B
    bg = {k: torch.cat([parts[0][0][k], parts[1][0][k]]) for k in parts[0][0]}
    upg = torch.cat([parts[0][1], parts[1][1]])
    engs[2].train_step(bg, upg, lr=1e-3, optimizer=False)
    Lg = engs[2].losses()
    # every rank holds its local rows' share of the global InfoNCE sums: rank sums add up to the global value
    sc = sum(float(e.scal[2] + e.scal[3]) for e in engs[:2]) * 0.5 / float(engs[2].scal[4])
    log(f"two-rank emulation: clip global {Lg['clip_loss']:.6f} from rank sums {sc:.6f}; per-rank views {clip_ranks}")
    assert abs(sc - Lg["clip_loss"]) < 2e-4 * max(1.0, abs(Lg["clip_loss"]))
    worst = 0.0
    bad, devs = [], []
    for name, (off, shape) in engs[2].layout.items():
        n = int(np.prod(shape))
        a, b_ = engs[2].grads[off:off + n], avg[off:off + n]
        scale = float(a.abs().max())
        if scale == 0.0:
            assert float(b_.abs().max()) == 0.0, name
            continue
        dev_ = float((a - b_).abs().max()) / scale
        worst = max(worst, dev_)
        devs.append(dev_)
        # Rows are processed identically in both runs; what differs is the fp32 summation order of the InfoNCE sums
        # (1e-7), which can flip single bf16 roundings of activation gradients.  A flipped element changes a bias
        # gradient (a heavily cancelling column sum: |sum| ~ 0.1 |terms|) by a few percent of its small scale.
        if dev_ > 5e-2:
            bad.append((dev_, name, scale))
    devs.sort()
    log(f"two-rank emulation: parameter-gradient deviation median {devs[len(devs) // 2]:.3e}, worst {worst:.3e} of tensor scale")
    assert not bad, sorted(bad, reverse=True)[:6]
    assert devs[len(devs) // 2] < 1e-3


@pytest.mark.parametrize("case", ["single_molecule_min_width", "full_n_seq", "no_ar_targets_all_bad_but_one", "one_atom"])
def test_edge_shapes_vs_oracle(case):
    """Edge shapes of the step against the oracle (bf16-simulation mode): a single molecule at the smallest width the
    tensorisation can produce, the maximum sequence length (n_seq), a batch whose rows are all failures but one (no AR
    targets survive in the failed rows; InfoNCE labels -1), and single-atom point clouds."""
    from oracle import coati_oracle as O
    from coati_amd.engine import Engine, ModelConfig
    from coati_amd.synthetic import make_batch
    n_seq = 250 if case == "full_n_seq" else 40
    kw = dict(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64, n_head=4,
              n_seq=n_seq, n_tok=120)
    ocfg = O.OracleConfig(**kw)
    P = O.init_params(ocfg, seed=21)
    eng = Engine(ModelConfig(**kw), DEV)
    eng.load_state_dict(P)
    if case == "single_molecule_min_width":
        batch, up = make_batch(1, 6, 3, 120, seed=1, n_special=12, p_bad=0.0, min_len=1)
    elif case == "full_n_seq":
        # (seed 2 gives an almost saturated InfoNCE: gradients 30x smaller, and the bf16 simulation itself then sits 19 %
        # from fp32 -- a conditioning property of that batch, not of the kernels; tools/dbg_edge.py)
        batch, up = make_batch(3, 250, 7, 120, seed=3, n_special=12, p_bad=0.0, min_len=200)
    elif case == "no_ar_targets_all_bad_but_one":
        batch, up = make_batch(6, 20, 5, 120, seed=3, n_special=12, p_bad=1.0, min_len=4)   # row 0 is always kept valid
    else:
        batch, up = make_batch(5, 18, 1, 120, seed=4, n_special=12, p_bad=0.0, min_len=4)
    db = {k: v.to(DEV) for k, v in batch.items()}
    eng.train_step(db, up.to(DEV), lr=1e-3, optimizer=False)
    L = eng.losses()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with O.sim_bf16():
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
    loss.backward()
    log(f"edge[{case}] hip {L} oracle ar {float(ar.detach()):.6f} clip {float(cl.detach()):.6f}")
    check(f"edge {case} ar", torch.tensor([L["ar_loss"]]), ar.detach().reshape(1), 5e-3)
    check(f"edge {case} clip", torch.tensor([L["clip_loss"]]), cl.detach().reshape(1), 5e-3)
    grads = eng.named_views("grads")
    bad = []
    for k in sorted(eng.layout):
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
        scale = max(float(ref.abs().max()), 1e-30)
        e = float((grads[k].cpu() - ref).abs().max()) / scale if float(ref.abs().max()) > 0 else float(grads[k].abs().max())
        if e > TOL_GRAD_SIM:
            bad.append((e, k))
    assert not bad, sorted(bad, reverse=True)[:8]
