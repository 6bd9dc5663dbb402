"""Constructor flags of e3gnn_smiles_clip_e2e outside the grande setting (clip_e2e.py:405-437, 454-463; the reference's own
do_args() defaults are norm_clips=False, token_mlp=False: train_coati.py:520-523): the oracle against vectors the reference
produced (tests/golden/gen_golden_flags.py -> flags_golden.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import coati_oracle as O

CASES = {
    "doargs": dict(norm_clips=False, token_mlp=False, use_point_encoder=True),
    "nopoint": dict(norm_clips=False, token_mlp=False, use_point_encoder=False),
    "mixed": dict(norm_clips=True, token_mlp=False, use_point_encoder=True),
    "mlp_nopoint": dict(norm_clips=True, token_mlp=True, use_point_encoder=False),
    "nobias": dict(norm_clips=True, token_mlp=True, use_point_encoder=True, biases=False),
    "normembed": dict(norm_clips=True, token_mlp=True, use_point_encoder=True, norm_embed=True),
    "torchemb": dict(norm_clips=True, token_mlp=True, use_point_encoder=True, torch_emb=True),
    "oldarch": dict(norm_clips=True, token_mlp=True, use_point_encoder=True, old_architecture=True),
    "residual": dict(norm_clips=True, token_mlp=True, use_point_encoder=True, residual=True),
}
SMALL = dict(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64, n_head=4, n_seq=24, n_tok=48)


def close(a, b, tol, name):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{name}: max err {err:.3e} (scale {scale:.3e})"


def load_case(golden_dir, case):
    z = np.load(os.path.join(golden_dir, "flags_golden.npz"))
    pre = case + "."
    d = {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    batch = {k: torch.from_numpy(z["b_" + k]) for k in ("raw_tokens", "tokens", "atoms", "coords", "y_next")}
    return d, batch, torch.from_numpy(z["use_point"])


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_flags_vs_reference(golden_dir, case):
    d, batch, up = load_case(golden_dir, case)
    cfg = O.OracleConfig(**SMALL, **CASES[case])
    P = {k[2:]: v for k, v in d.items() if k.startswith("w.")}
    # the state_dict contract: the same names and shapes as the reference model built with these flags
    shapes = O.param_shapes(cfg)
    assert set(shapes) == set(P)
    assert all(tuple(P[k].shape) == tuple(shapes[k]) for k in P)
    he, hs, lg, bad = O.forward_dist(P, cfg, batch["raw_tokens"], batch["tokens"], batch["atoms"], batch["coords"], up)
    close(he, d["h_e3gnn"], 2e-5, "h_e3gnn")
    close(hs, d["h_smiles"], 2e-5, "h_smiles")
    close(lg, d["logits"], 2e-5, "logits")
    assert torch.equal(bad, d["bad"])
    if not CASES[case]["use_point_encoder"]:
        assert float(he.abs().max()) == 0.0
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss, ar, cl, _ = O.step_loss(Pg, cfg, batch, up)
    close(ar, d["ar"], 2e-5, "ar")
    close(cl, d["clip"], 2e-5, "clip")
    close(loss, d["loss"], 2e-5, "loss")
    loss.backward()
    grads, trainable = {}, {}
    for k in Pg:
        if "nograd." + k in d:
            # p.grad is None in the reference: torch's clip_grad_norm_ / AdamW skip the parameter (no weight decay either)
            assert Pg[k].grad is None or float(Pg[k].grad.abs().max()) == 0.0, k
            continue
        g = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])
        close(g, d["grad." + k], 1e-4, "grad " + k)
        trainable[k] = g
    norm, coef = O.clip_grad_norm(trainable, 10.0)
    close(norm, d["gradnorm"], 1e-4, "gradnorm")
    for k in P:
        if k in trainable:
            p1, _, _ = O.adamw_update(P[k], trainable[k] * coef, torch.zeros_like(P[k]), torch.zeros_like(P[k]), step=1, lr=5e-4)
        else:
            p1 = P[k]
        # (the first AdamW step moves every element by ~ lr = 5e-4 times g / (|g| + eps): an element whose gradient is ~ 1e-8 turns a 1e-9
        #  difference of the two fp32 gradient sums into a few percent of lr -- 3.0e-5 measured on one element of the normembed case)
        close(p1.reshape(-1)[::13], d["after1." + k], 4e-5, "adamw " + k)
