"""Parity of the HIP engine at the HEADLINE architecture -- grande_closed: d = 256, 16 transformer layers, 16 heads, E(3)-GNN
256 x 5, V = 10 322 (examples/training/train_grande.py:21-35) -- against vectors the REFERENCE produced at that architecture
(tests/golden/grande_golden.npz, written by tests/golden/gen_golden_grande.py from the imported reference): forward_dist
outputs, the training step's losses, every parameter gradient, clip-norm, the first AdamW update and a 20-step loss curve;
and against the fp32 oracle / the oracle with bf16 storage simulated on a second, larger batch.

Tolerances: at most 2x the deviation measured on the MI355X (gpurun_out/test_report.txt, round 3), relative to tensor scale."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.gpu_util import check, log  # noqa: E402
from tests import grande_util as GU  # noqa: E402

DEV = "cuda:0"
TOL_FWD = 1.1e-2         # measured 5.3e-3 (logits), 2.8e-3 (h_smiles), 2.3e-3 (h_e3gnn), 2.7e-5 (log-sum-exp): bf16 operands through 16 layers
TOL_LOSS = 1e-3          # measured 5.1e-4 (clip), 1.4e-5 (ar)
TOL_GRAD = 3e-2          # full parameter gradients, relative to each tensor's scale: measured 1.42e-2 (h.0.ln_1.weight), 1.15e-2 (gcl_0.edge_mlp.0)
TOL_GRADNORM_EACH = 6e-3  # every parameter's gradient norm, relative to that norm: measured 2.9e-3
TOL_GRADNORM = 3e-3      # clip_grad_norm_ value: measured 1.5e-3
TOL_CURVE = 8e-3         # 20-step curve: measured 3.4e-3 (loss), 1.7e-3 (ar), 4.0e-3 (clip)
# Gradient NORM along the curve: the reference's norm collapses from 1144 (step 0) to 3..10 (steps 9+) -- the late gradients are
# small residuals of cancelling terms and their norm is ill-conditioned: the fp32 ORACLE started from weights perturbed by 1e-3
# deviates from the reference's curve by up to 2.3e-1 (step 9; median 1.6e-2 over 14 steps), the oracle with bf16 storage
# simulated by up to 1.5e-1 (tools/curve_bf16_sim_grande.py, profiles/r03_curve_bf16_sim_grande.txt), while both keep the LOSS
# within 4e-3.  The engine measured 2.7e-1 at step 11, median 4.6e-2.  So: the first five steps (norm > 90, well conditioned) are
# held tightly, the rest to 2x that envelope plus a median bound.
TOL_CURVE_GN_EARLY = 1.4e-2   # steps 0..4: measured 6.5e-3
# Round 6: the bound of the later steps is RELATIVE to what rounding alone does to this trajectory -- tests/golden/grande_curve_envelope.json
# (tools/curve_bf16_sim_grande.py --envelope: per step, the largest deviation of the bf16-storage-simulating oracle and of three fp32
# oracles started from weights perturbed by 1e-3): step i is held to ENVELOPE_FACTOR x envelope[i] (never tighter than the early-step
# bound), the median over the steps to the envelope's own median.  The mid-curve step itself is pinned WITHOUT trajectory noise by
# test_mid_curve_step_from_replayed_weights below.
ENVELOPE_FACTOR = 3.0
TOL_MID_GRADNORM_MEDIAN = 2.5e-3   # per-parameter gradient norms at step 10 from the replayed weights: measured 1.04e-3 (the bf16-storage-simulating oracle: 3.6e-3)
TOL_MID_GRADNORM_MAX = 8e-3        # measured 3.3e-3 (bf16-sim oracle: 5.3e-3)


@pytest.fixture(scope="module")
def gr(golden_dir):
    from coati_amd.engine import Engine, ModelConfig
    g, ocfg, P, names, batches, masks = GU.load(golden_dir)
    eng = Engine(ModelConfig(**GU.GRANDE), DEV)
    eng.load_state_dict(P)
    db = [{k: v.to(DEV) for k, v in b.items()} for b in batches]
    return g, ocfg, P, names, batches, masks, eng, db


def _layout(db, batches, layout):
    """padded: the [B, T] matrices as the reference computes them; packed: + the host-side row counts, so that the engine
    skips the positions behind each row's last token"""
    if layout == "padded":
        return db
    from coati_amd.synthetic import packed_rows
    return [dict(d, rows=torch.tensor(packed_rows(b["raw_tokens"], b["tokens"], b["y_next"]))) for d, b in zip(db, batches)]


def test_forward_dist_grande_vs_reference(gr):
    g, ocfg, P, names, batches, masks, eng, db = gr
    b = db[0]
    up = masks[int(g["n_steps"])].to(DEV)
    he, hs, bad = eng.forward(b["raw_tokens"], b["tokens"], b["atoms"], b["coords"], up, y_next=b["y_next"], train=False)
    lg = eng.logits().cpu()
    check("grande h_e3gnn", he.cpu(), torch.from_numpy(g["fd_h_e3gnn"]), TOL_FWD)
    check("grande h_smiles", hs.cpu(), torch.from_numpy(g["fd_h_smiles"]), TOL_FWD)
    assert torch.equal(bad.cpu().bool(), torch.from_numpy(g["fd_bad"]))
    check("grande logits rows", lg[[int(i) for i in g["fd_rows"]]], torch.from_numpy(g["fd_logits_rows"]), TOL_FWD)
    check("grande log-sum-exp", torch.logsumexp(lg, -1), torch.from_numpy(g["fd_lse"]), TOL_FWD)
    tgt = torch.gather(lg, 2, batches[0]["y_next"].clamp(min=0).unsqueeze(-1)).squeeze(-1)
    # logit scale (not the target logits' own range) is the reference for the gathered values
    sc = float(np.abs(g["fd_logits_rows"]).max())
    e = float((tgt - torch.from_numpy(g["fd_logit_at_target"])).abs().max()) / sc
    log(f"grande logit at target: max deviation {e:.3e} of the logit scale")
    assert e <= TOL_FWD
    agree = float((lg.argmax(-1) == torch.from_numpy(g["fd_argmax"])).float().mean())
    log(f"grande arg-max agreement with the reference: {agree:.4f}")
    assert agree > 0.97


@pytest.mark.parametrize("layout", ["padded", "packed"])
def test_step_grads_adamw_grande_vs_reference(gr, layout):
    g, ocfg, P, names, batches, masks, eng, db = gr
    db = _layout(db, batches, layout)
    eng.load_state_dict(P)
    eng.adam_m.zero_(); eng.adam_v.zero_(); eng.step_count = 0
    eng.train_step(db[0], masks[0].to(DEV), lr=5e-4, weight_decay=0.1, max_norm=10.0, optimizer=False)
    assert eng._packed == (layout == "packed")
    L = eng.losses()
    log(f"grande step [{layout}]: hip {L}  reference ar {float(g['step_ar']):.6f} clip {float(g['step_clip']):.6f} loss {float(g['step_loss']):.6f}")
    check("grande ar", torch.tensor([L["ar_loss"]]), torch.from_numpy(g["step_ar"]).reshape(1), TOL_LOSS)
    check("grande clip", torch.tensor([L["clip_loss"]]), torch.from_numpy(g["step_clip"]).reshape(1), TOL_LOSS)
    check("grande loss", torch.tensor([L["loss"]]), torch.from_numpy(g["step_loss"]).reshape(1), TOL_LOSS)
    grads = {k: v.cpu() for k, v in eng.named_views("grads").items()}
    gn = np.array([float(grads[n].double().norm()) for n in names])
    gp = np.array([float((grads[n].double().flatten() * GU.projection(n, grads[n].numel()).double()).sum()) for n in names])
    big = g["grad_norms"] > 1e-3 * g["grad_norms"].max()
    rel = np.abs(gn - g["grad_norms"]) / np.maximum(g["grad_norms"], 1e-30)
    worst = sorted(((rel[i], names[i]) for i in range(len(names)) if big[i]), reverse=True)[:4]
    log(f"grande per-parameter gradient norms: worst relative deviations {worst}")
    assert worst[0][0] <= TOL_GRADNORM_EACH, worst
    # a +-1 projection of a gradient has the scale of its norm
    prel = np.abs(gp - g["grad_projs"]) / np.maximum(g["grad_norms"], 1e-3 * g["grad_norms"].max())
    log(f"grande gradient projections: worst deviation {prel.max():.3e} of the parameter's gradient norm ({names[int(prel.argmax())]})")
    assert prel.max() <= 4.6e-2     # measured 2.3e-2
    assert all(float(grads[n].abs().max()) == 0.0 for n in names if "coord_mlp" in n)
    rows = torch.from_numpy(g["row_subset"])
    for k in sorted(g):
        if k.startswith("grad."):
            check("grande " + k, grads[k[5:]], torch.from_numpy(g[k]), TOL_GRAD)
        elif k.startswith("gradrows."):
            check("grande " + k, grads[k[9:]][rows], torch.from_numpy(g[k]), TOL_GRAD)
    # clip-norm + AdamW on these gradients
    eng.optimizer_step(5e-4, weight_decay=0.1, max_norm=10.0)
    L = eng.losses()
    check("grande clip_grad_norm", torch.tensor([L["grad_norm"]]), torch.from_numpy(g["step_gradnorm"]).reshape(1), TOL_GRADNORM)
    sd = eng.state_dict()
    dn = np.array([float((sd[n].cpu() - P[n]).double().norm()) for n in names])
    reln = np.abs(dn - g["delta_norms"]) / np.maximum(g["delta_norms"], 1e-3 * g["delta_norms"].max())
    log(f"grande AdamW update norms: worst relative deviation {reln.max():.3e} ({names[int(reln.argmax())]})")
    assert reln.max() <= 5e-3     # measured 2.5e-3
    # Adam's first step is sign-like (lr * g / (|g| + eps)): the direction of each tensor's update against the reference's
    for k in sorted(g):
        if k.startswith("after1."):
            n = k[7:]
            d_ref = torch.from_numpy(g[k]).double() - P[n].flatten()[::7].double()
            d_hip = (sd[n].cpu() - P[n]).flatten()[::7].double()
            cos = float((d_hip @ d_ref) / (d_hip.norm() * d_ref.norm() + 1e-30))
            log(f"grande adamw displacement {n:55s} cosine {cos:.4f}")
            assert cos > 0.9, (n, cos)     # measured >= 0.9528 (h.0.attn.c_attn.bias), 0.994+ for every matrix


def _mid_curve_weights(g, P, names, eng, layout):
    """Mid-curve pin on the WEIGHTS (not a loss envelope): every 97th element of every tensor the reference holds at the start of
    step 10.  Ten AdamW steps move an element by <= 10 lr = 5e-3 whatever its gradient's size, so the displacement from the initial
    weights is compared as a vector per tensor -- direction (cosine) and length; a backward that had gone wrong after step 0
    (which test_step_grads_adamw pins in full) turns the displacement of the tensors it touches."""
    views = eng.named_views("params")
    cos_all, len_all, worst = [], [], []
    for n_ in names:
        ref = torch.from_numpy(g["mid.w." + n_]).double()
        w0 = P[n_].flatten()[::97].double()
        w = views[n_].detach().cpu().flatten()[::97].double()
        dr, de = ref - w0, w - w0
        if float(dr.norm()) == 0.0:
            assert float(de.norm()) == 0.0, n_          # coord_mlp: never moves
            continue
        if dr.numel() < 64:
            continue                                     # (too few samples of a small tensor for a direction)
        c = float((dr * de).sum() / (dr.norm() * de.norm() + 1e-300))
        cos_all.append(c); len_all.append(float(de.norm() / dr.norm())); worst.append((c, n_))
    worst.sort()
    log(f"grande mid-curve weights [{layout}]: displacement after 10 steps vs reference: cosine median {np.median(cos_all):.4f} min {worst[0][0]:.4f} "
        f"({worst[0][1]}), length ratio {min(len_all):.3f} .. {max(len_all):.3f}")
    assert np.median(cos_all) >= 0.99 and worst[0][0] >= 0.95, worst[:4]     # measured 0.9995 / 0.9983 (tok_emb)
    assert 0.9 <= min(len_all) and max(len_all) <= 1.1, (min(len_all), max(len_all))


def _mid_curve_grad_norms(g, names, eng, layout):
    """every parameter's gradient norm at step 10 (pre-clip) against the reference's: median over the parameters -- a systematic
    backward error shows in all of them, trajectory noise does not"""
    grads = eng.named_views("grads")
    ref = g["mid_grad_norms"]
    dev = []
    for i, n_ in enumerate(names):
        if ref[i] > 0:
            dev.append(abs(float(grads[n_].double().norm()) - ref[i]) / ref[i])
    log(f"grande mid-curve gradient norms [{layout}] (step 10, {len(dev)} parameters, along the engine's OWN trajectory): median deviation {np.median(dev):.3e}, 90th percentile {np.percentile(dev, 90):.3e}")
    # along the engine's own ten-step trajectory this number is trajectory noise (rounds 5-6 measured 2.0e-2 .. 1.5e-1 for arithmetic-
    # equivalent kernels; the total gradient norm's envelope around this step is 0.25): logged, and bounded only against a
    # backward that has gone wrong altogether.  The tight statement is test_mid_curve_step_from_replayed_weights.
    assert np.median(dev) <= 0.5, np.median(dev)


@pytest.mark.parametrize("layout", ["padded", "packed"])
def test_twenty_step_loss_curve_grande_vs_reference(gr, layout):
    """north_star "loss-curve equivalent to reference" AT THE HEADLINE ARCHITECTURE: 20 optimiser steps (clip-norm 10, AdamW
    lr 5e-4, wd 0.1, betas (0.9, 0.99): train_grande.py's settings) cycling over 4 batches, mixed point / SMILES injection,
    against the curve the reference produced on its fp32 CPU path."""
    g, ocfg, P, names, batches, masks, eng, db = gr
    db = _layout(db, batches, layout)
    eng.load_state_dict(P)
    eng.adam_m.zero_(); eng.adam_v.zero_(); eng.step_count = 0
    n = int(g["n_steps"])
    rec = dict(loss=[], ar=[], clip=[], gradnorm=[])
    mid = int(g["mid_step"])
    for step in range(n):
        if step == mid:
            _mid_curve_weights(g, P, names, eng, layout)
        eng.train_step(db[step % 4], masks[step].to(DEV), lr=5e-4, weight_decay=0.1, max_norm=10.0)
        L = eng.losses()
        if step == mid:
            _mid_curve_grad_norms(g, names, eng, layout)
        rec["loss"].append(L["loss"]); rec["ar"].append(L["ar_loss"]); rec["clip"].append(L["clip_loss"]); rec["gradnorm"].append(L["grad_norm"])
    dev_ = {k: np.abs(np.array(v) - g["curve_" + k]) / np.maximum(np.abs(g["curve_" + k]), 1e-6) for k, v in rec.items()}
    log("grande 20-step curve: reference ar " + " ".join(f"{x:.3f}" for x in g["curve_ar"]))
    log("grande 20-step curve: hip       ar " + " ".join(f"{x:.3f}" for x in rec["ar"]))
    log(f"grande 20-step curve [{layout}]: max relative deviation " + ", ".join(f"{k} {v.max():.3e} (step {int(v.argmax())}, median {np.median(v):.2e})" for k, v in dev_.items()))
    assert g["curve_ar"][-4:].mean() < 0.9 * g["curve_ar"][:4].mean()      # the curve really descends
    for k in ("loss", "ar", "clip"):
        assert dev_[k].max() <= TOL_CURVE, (k, dev_[k].max())
    log("grande 20-step curve: gradnorm deviation per step " + " ".join(f"{x:.1e}" for x in dev_["gradnorm"]))
    assert dev_["gradnorm"][:5].max() <= TOL_CURVE_GN_EARLY, dev_["gradnorm"][:5]
    import json, os
    env = np.array(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grande_curve_envelope.json")))["gradnorm_dev_max_per_step"])[:n]
    # (an excursion's phase along the trajectory is as unpredictable as its size -- the four envelope runs peak at steps 9, 11-13, 15 --
    #  so step i is held to the envelope's largest value within two steps of it)
    env = np.array([env[max(0, i - 2):i + 3].max() for i in range(len(env))])
    bound = np.maximum(ENVELOPE_FACTOR * env, TOL_CURVE_GN_EARLY)
    bound[:5] = TOL_CURVE_GN_EARLY
    log("grande 20-step curve: bound per step (3 x rounding envelope)  " + " ".join(f"{x:.1e}" for x in bound))
    assert (dev_["gradnorm"] <= bound).all(), (dev_["gradnorm"] / bound)
    assert np.median(dev_["gradnorm"]) <= np.median(env), (np.median(dev_["gradnorm"]), np.median(env))


@pytest.mark.parametrize("layout", ["padded", "packed"])
def test_mid_curve_step_from_replayed_weights(gr, layout):
    """The mid-curve pin without trajectory noise.  The ORACLE (fp32, host) replays the reference's first ten optimiser steps -- it
    reproduces the reference's curve to 8e-5 (tools/curve_bf16_sim_grande.py), and the weights it arrives at are checked here against
    the strided sample of the reference's step-10 weights the fixture holds -- then the ENGINE evaluates ONE step from those weights
    and every parameter's gradient norm is compared with the reference's at that step (`mid_grad_norms`).  A backward that is wrong
    shows in all of them; ten steps of rounding history do not enter."""
    from oracle import coati_oracle as O
    g, ocfg, P, names, batches, masks, eng, db = gr
    mid = int(g["mid_step"])
    cache = getattr(test_mid_curve_step_from_replayed_weights, "_w", None)
    if cache is None:
        torch.set_num_threads(min(32, torch.get_num_threads()))
        W = {k: v.clone() for k, v in P.items()}
        M = {k: torch.zeros_like(v) for k, v in W.items()}
        V = {k: torch.zeros_like(v) for k, v in W.items()}
        for step in range(mid):
            Pg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
            loss, *_ = O.step_loss(Pg, ocfg, batches[step % 4], masks[step])
            loss.backward()
            grads = {k: (Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])) for k in Pg}
            norm, coef = O.clip_grad_norm(grads, 10.0)
            assert abs(float(norm) - float(g["curve_gradnorm"][step])) <= 2e-3 * float(g["curve_gradnorm"][step]), (step, float(norm))
            for k in W:
                if "coord_mlp" in k:
                    continue
                W[k], M[k], V[k] = O.adamw_update(W[k], grads[k] * coef, M[k], V[k], step=step + 1, lr=5e-4)
        worst = max(float((W[n_].flatten()[::97] - torch.from_numpy(g["mid.w." + n_])).abs().max()) for n_ in names)
        log(f"grande replayed weights at step {mid} vs the reference's strided sample: max |difference| {worst:.2e}")
        assert worst <= 4e-5, worst      # measured 1.1e-5 (ten AdamW steps of fp32 summation-order differences between hosts)
        test_mid_curve_step_from_replayed_weights._w = cache = W
    db = _layout(db, batches, layout)
    eng.load_state_dict(cache)
    eng.train_step(db[mid % 4], masks[mid].to(DEV), lr=0.0, optimizer=False)
    L = eng.losses()
    grads = eng.named_views("grads")
    ref = g["mid_grad_norms"]
    dev = np.array([abs(float(grads[n_].double().norm()) - ref[i]) / ref[i] for i, n_ in enumerate(names) if ref[i] > 0])
    log(f"grande step {mid} from the replayed weights [{layout}]: loss {L['loss']:.5f} (reference {float(g['curve_loss'][mid]):.5f}); per-parameter gradient norms: "
        f"median deviation {np.median(dev):.3e}, 90th percentile {np.percentile(dev, 90):.3e}, max {dev.max():.3e}")
    check(f"grande step {mid} loss from replayed weights [{layout}]", torch.tensor([L["loss"]]), torch.tensor([float(g["curve_loss"][mid])]), TOL_LOSS)
    assert np.median(dev) <= TOL_MID_GRADNORM_MEDIAN and dev.max() <= TOL_MID_GRADNORM_MAX, (np.median(dev), dev.max())
    eng.load_state_dict(P)


def test_grande_step_vs_oracle_fp32_and_bf16_sim(gr):
    """The same architecture on a second batch (32 molecules, 80-token rows, ragged atom counts, bad rows) against the
    oracle run here on the host: fp32 (the reference's arithmetic) and with bf16 storage simulated where the engine rounds."""
    from oracle import coati_oracle as O
    from coati_amd.synthetic import make_batch
    g, ocfg, P, names, batches, masks, eng, db = gr
    eng.load_state_dict(P)
    batch, up = make_batch(32, 80, 16, GU.GRANDE["n_tok"], seed=77, n_special=1596, p_bad=0.06, min_len=16)
    eng.train_step({k: v.to(DEV) for k, v in batch.items()}, up.to(DEV), lr=0.0, optimizer=False)
    L = eng.losses()
    grads = {k: v.cpu() for k, v in eng.named_views("grads").items()}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    for tag, sim, tl, tg in (("fp32", False, TOL_LOSS, TOL_GRAD), ("bf16-sim", True, TOL_LOSS, TOL_GRAD)):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        if sim:
            with O.sim_bf16():
                loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
        else:
            loss, ar, cl, _ = O.step_loss(Pg, ocfg, batch, up)
        loss.backward()
        check(f"grande-32 ar vs {tag} oracle", torch.tensor([L["ar_loss"]]), ar.detach().reshape(1), tl)
        check(f"grande-32 clip vs {tag} oracle", torch.tensor([L["clip_loss"]]), cl.detach().reshape(1), tl)
        worst = []
        for k in names:
            ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(P[k])
            sc = float(ref.abs().max())
            if sc > 0:
                worst.append((float((grads[k] - ref).abs().max()) / sc, k))
        worst.sort(reverse=True)
        log(f"grande-32 vs {tag} oracle: worst gradient deviations {worst[:4]}")
        assert worst[0][0] <= tg, worst[:5]
