"""The oracle at the HEADLINE architecture (d = 256, 16 layers, 16 heads, GNN 256 x 5, V = 10 322) against vectors the
reference produced (tests/golden/grande_golden.npz, gen_golden_grande.py): forward_dist, the training step's losses,
every parameter gradient (norm + fixed projection; full tensors for 26 representative parameters), clip-norm, the first
AdamW update, and the first steps of the 20-step curve.  CPU only; fp32 re-association across 16 layers: 2e-4 of scale."""
import numpy as np
import pytest
import torch

from oracle import coati_oracle as O
from tests import grande_util as GU

TOL = 2e-4


def close(a, b, tol=TOL, name=""):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{name}: max err {err:.3e} (scale {scale:.3e})"


@pytest.fixture(scope="module")
def gr(golden_dir):
    torch.set_num_threads(min(16, torch.get_num_threads()))
    return GU.load(golden_dir)


def test_forward_dist_grande(gr):
    g, ocfg, P, names, batches, masks = gr
    b = batches[0]
    with torch.no_grad():
        he, hs, lg, bad = O.forward_dist(P, ocfg, b["raw_tokens"], b["tokens"], b["atoms"], b["coords"], masks[int(g["n_steps"])])
    close(he, g["fd_h_e3gnn"], name="h_e3gnn")
    close(hs, g["fd_h_smiles"], name="h_smiles")
    assert torch.equal(bad, torch.from_numpy(g["fd_bad"]))
    close(torch.logsumexp(lg, -1), g["fd_lse"], name="lse")
    close(lg[[int(i) for i in g["fd_rows"]]], g["fd_logits_rows"], name="logits rows")
    tgt = torch.gather(lg, 2, b["y_next"].clamp(min=0).unsqueeze(-1)).squeeze(-1)
    close(tgt, g["fd_logit_at_target"], name="logit at target")
    assert (lg.argmax(-1) == torch.from_numpy(g["fd_argmax"])).float().mean() > 0.995


def test_step_grads_adamw_and_curve_grande(gr):
    g, ocfg, P, names, batches, masks = gr
    P = {k: v.clone() for k, v in P.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    rows = torch.from_numpy(g["row_subset"])
    n_check = 3                                    # steps of the curve the oracle replays here (CPU time)
    for step in range(n_check):
        b = batches[step % 4]
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        loss, ar, cl, _ = O.step_loss(Pg, ocfg, b, masks[step])
        loss.backward()
        grads = {k: (Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])) for k in Pg}
        norm, coef = O.clip_grad_norm(grads, 10.0)
        close(loss.detach(), g["curve_loss"][step], name=f"loss[{step}]")
        close(ar.detach(), g["curve_ar"][step], name=f"ar[{step}]")
        close(cl.detach(), g["curve_clip"][step], name=f"clip[{step}]")
        close(norm, g["curve_gradnorm"][step], tol=5e-4, name=f"gradnorm[{step}]")
        if step == 0:
            close(loss.detach(), g["step_loss"], name="loss")
            gn = np.array([float(grads[n].double().norm()) for n in names])
            gp = np.array([float((grads[n].double().flatten() * GU.projection(n, grads[n].numel()).double()).sum()) for n in names])
            sc = g["grad_norms"].max()
            assert np.abs(gn - g["grad_norms"]).max() <= 5e-4 * sc, "per-parameter gradient norms"
            # a +-1 projection of n elements has scale ~ norm; compare against each parameter's own norm
            assert (np.abs(gp - g["grad_projs"]) <= 2e-3 * np.maximum(g["grad_norms"], 1e-3 * sc)).all(), "gradient projections"
            for k in g:
                if k.startswith("grad."):
                    close(grads[k[5:]], g[k], tol=5e-4, name=k)
                elif k.startswith("gradrows."):
                    close(grads[k[9:]][rows], g[k], tol=5e-4, name=k)
        P0 = P if step == 0 else None
        newP = {}
        for k in P:
            newP[k], M[k], V[k] = O.adamw_update(P[k], grads[k] * coef, M[k], V[k], step=step + 1, lr=5e-4)
        if "coord_mlp" in "".join(P):            # coord_mlp never receives a gradient: torch skips it (no weight decay either)
            for k in P:
                if "coord_mlp" in k:
                    newP[k] = P[k]
        if step == 0:
            dn = np.array([float((newP[n] - P0[n]).double().norm()) for n in names])
            assert np.abs(dn - g["delta_norms"]).max() <= 2e-3 * g["delta_norms"].max(), "AdamW update norms"
            # step 1 of Adam moves every weight by lr * g / (|g| + eps'): compare the UPDATE in units of lr; elements whose
            # gradient is ~ eps (1e-8) flip with fp32 noise, so a 1e-3 fraction may be off by up to 2 lr
            for k in g:
                if k.startswith("after1."):
                    n = k[7:]
                    d_ref = torch.from_numpy(g[k]).double() - P0[n].flatten()[::7].double()
                    d_me = (newP[n] - P0[n]).flatten()[::7].double()
                    e = (d_me - d_ref).abs() / 5e-4
                    assert float((e > 2e-2).double().mean()) <= 1e-3 and float(e.max()) <= 2.05, (k, float(e.max()), float((e > 2e-2).double().mean()))
        P = newP
