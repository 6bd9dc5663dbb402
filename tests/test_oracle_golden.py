"""The oracle (oracle/coati_oracle.py) against golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  CPU only.  fp32 tolerance: 2e-5 relative-to-scale unless stated."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import coati_oracle as O

TOL = 2e-5


def close(a, b, tol=TOL, name=""):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{name}: max err {err:.3e} (scale {scale:.3e})"


@pytest.fixture(scope="module")
def vec(golden_dir):
    z = np.load(os.path.join(golden_dir, "small_vectors.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="module")
def small(golden_dir):
    z = np.load(os.path.join(golden_dir, "small_model.npz"))
    P = {k: torch.from_numpy(z[k]) for k in z.files}
    cfg = O.OracleConfig(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64,
                         n_embd_common=64, n_head=4, n_seq=24, n_tok=48)
    return P, cfg


def batch_of(vec):
    return {k: vec["b_" + k] for k in ("raw_tokens", "tokens", "atoms", "coords", "y_next")}


def test_constants_and_lut(golden_dir):
    c = json.load(open(os.path.join(golden_dir, "constants.json")))
    for name in ("mar", "may_closedparen"):
        t = c["tokenizers"][name]
        cfg = O.OracleConfig()
        assert (t["pad"], t["stop"], t["smiles"], t["suffix"], t["middle"], t["unk"], t["clip"]) == (
            cfg.pad_token, cfg.stop_token, cfg.smiles_token, cfg.suffix_token, cfg.middle_token,
            cfg.unk_token, cfg.clip_token)
    assert c["tokenizers"]["may_closedparen"]["n_token"] == 10322
    assert c["tokenizers"]["mar"]["n_token"] == 13603
    for z, (x, y) in enumerate(c["xy_lut"]):
        assert O.xy_position(z) == (x, y), z
    assert O.onehot_indices(0) == (27, 17)
    assert O.onehot_indices(2) == (18, 19)
    with pytest.raises(IndexError):
        O.onehot_indices(92)


def test_param_shapes_match_state_dict(small):
    P, cfg = small
    shapes = O.param_shapes(cfg)
    assert set(shapes) == set(P)
    for k, s in shapes.items():
        assert tuple(P[k].shape) == s, k
    # architecture pin: grande 'closed' = 20.36M parameters
    g = O.param_shapes(O.OracleConfig())
    n_blocks = sum(int(np.prod(s)) for k, s in g.items() if ".transformer." in k)
    n_gnn = sum(int(np.prod(s)) for k, s in g.items() if k.startswith("point_encoder."))
    n_x = sum(int(np.prod(s)) for k, s in g.items() if k.startswith("xformer."))
    assert round(n_blocks / 1e6, 2) == 12.64
    assert round(n_gnn / 1e6, 2) == 2.44
    assert round(n_x / 1e6, 2) == 17.92
    assert round((n_gnn + n_x) / 1e6, 2) == 20.36


def test_rotary_and_gelu(vec):
    cos, sin = O.rope_tables(24, 16)
    close(cos, vec["rot_cos"], name="cos")
    close(sin, vec["rot_sin"], name="sin")
    qr, kr = O.rotary_embed(vec["rot_q"], vec["rot_k"], cos, sin)
    close(qr, vec["rot_qr"], name="qr")
    close(kr, vec["rot_kr"], name="kr")
    close(O.new_gelu(vec["gelu_x"]), vec["gelu_y"], name="gelu")


def test_neighbor_list_and_cutoff(vec):
    nm = (vec["nl_atoms"] > 0).float()
    Is, Js, Ks, Ds = O.neighbor_list(vec["nl_coords"], nm)
    assert torch.equal(Is, vec["nl_Is"]) and torch.equal(Js, vec["nl_Js"]) and torch.equal(Ks, vec["nl_Ks"])
    close(Ds, vec["nl_Ds"], tol=1e-5, name="Ds")
    close(O.cubic_cutoff(vec["cut_d"]), vec["cut_f"], name="cutoff")


def test_gnn(vec, small):
    P, cfg = small
    atoms, coords = vec["b_atoms"], vec["b_coords"]
    nodes = O.atom_onehot(atoms)
    h0 = O.instance_norm(nodes @ P["point_encoder.embedding.weight"].t() + P["point_encoder.embedding.bias"])
    close(h0, vec["gnn_h0"], name="h0")
    emask, d = O.neighbor_mask(coords, (atoms > 0).float())
    h1 = O.gcl_layer(h0, emask, d, P, "point_encoder.gcl_0.")
    close(h1, vec["gnn_h1"], name="h1")
    close(O.point_encoder(atoms, coords, P, cfg), vec["gnn_out"], name="gnn_out")


def test_block_and_encode(vec, small):
    P, cfg = small
    cos, sin = O.rope_tables(cfg.n_seq, 16)
    pre = "xformer.transformer.h.0."
    x = vec["blk_x"]
    a1 = torch.nn.functional.layer_norm(x, (64,), P[pre + "ln_1.weight"], P[pre + "ln_1.bias"], 1e-5)
    close(O.attention(a1, P, pre + "attn.", 4, cos, sin), vec["blk_attn"], name="attn")
    close(O.block(x, P, pre, 4, cos, sin), vec["blk_y"], name="block")
    xe = O.xformer(vec["b_raw_tokens"], P, cfg)
    close(xe, vec["enc_x"], name="enc_x")
    close(O.stop_token_embs(xe, vec["b_raw_tokens"], 1), vec["enc_stop"], name="stop")
    bad = vec["b_raw_tokens"].clone()
    bad[0][bad[0] == 1] = 0
    with pytest.raises(RuntimeError):
        O.stop_token_embs(xe, bad, 1)


@pytest.mark.parametrize("tag,val", [("p0", True), ("p1", False)])
def test_forward_dist(vec, small, tag, val):
    P, cfg = small
    b = batch_of(vec)
    use_point = torch.full((b["atoms"].shape[0],), val)
    he, hs, lg, bad = O.forward_dist(P, cfg, b["raw_tokens"], b["tokens"], b["atoms"], b["coords"], use_point)
    close(he, vec[f"fd_{tag}_h_e3gnn"], name="h_e3gnn")
    close(hs, vec[f"fd_{tag}_h_smiles"], name="h_smiles")
    close(lg, vec[f"fd_{tag}_logits"], tol=5e-5, name="logits")
    assert torch.equal(bad, vec[f"fd_{tag}_bad"])


def test_forward_dist_mixed(vec, small):
    P, cfg = small
    b = batch_of(vec)
    _, _, lg, _ = O.forward_dist(P, cfg, b["raw_tokens"], b["tokens"], b["atoms"], b["coords"], vec["fd_mix_use_point"])
    close(lg, vec["fd_mix_logits"], tol=5e-5, name="logits_mixed")


def test_clip_loss(vec):
    a = vec["cl_a"].clone().requires_grad_(True)
    b = vec["cl_b"].clone().requires_grad_(True)
    close(O.clip_loss(a, b, torch.zeros(6, dtype=torch.bool)), vec["cl_l0"], name="l0")
    l1 = O.clip_loss(a, b, vec["cl_bad"])
    close(l1, vec["cl_l1"], name="l1")
    ga, gb = torch.autograd.grad(l1.sum(), (a, b))
    close(ga, vec["cl_ga"], name="ga")
    close(gb, vec["cl_gb"], name="gb")


def test_y_next(vec):
    assert torch.equal(O.y_next_from_tokens(vec["b_tokens"], O.OracleConfig()), vec["b_y_next"])


def test_xform_tail(golden_dir):
    z = np.load(os.path.join(golden_dir, "xform_tail.npz"))
    tok = torch.from_numpy(z["tokens"])
    assert torch.equal(O.y_next_from_tokens(tok, O.OracleConfig()), torch.from_numpy(z["y_next"]))
    # failing (oversized) row -> all-PAD tokens row and [STOP],PAD.. raw row (clip_e2e.py:255-270)
    raw = torch.from_numpy(z["raw_tokens"])
    assert int(tok[3].sum()) == 0 and int(raw[3, 0]) == 1 and int(raw[3, 1:].sum()) == 0


def test_full_step_grads_and_adamw(vec, small, golden_dir):
    P, cfg = small
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    b = batch_of(vec)
    use_point = torch.ones(b["atoms"].shape[0], dtype=torch.bool)
    loss, ar, cl, _ = O.step_loss(P, cfg, b, use_point)
    close(ar, vec["step_ar"], name="ar")
    close(cl, vec["step_clip"], name="clip")
    close(loss, vec["step_loss"], name="loss")
    assert abs(O.token_entropy_unit(48) - float(vec["teu"])) < 1e-12
    loss.backward()
    G = np.load(os.path.join(golden_dir, "small_step_grads.npz"))
    grads = {}
    for k in P:
        g = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
        grads[k] = g
        close(g, torch.from_numpy(G["grad." + k]), tol=1e-4, name="grad " + k)
    # coord_mlp is computed-and-discarded in the reference: zero gradient
    assert all(float(grads[k].abs().max()) == 0.0 for k in grads if "coord_mlp" in k)
    norm, coef = O.clip_grad_norm(grads, 10.0)
    close(norm, vec["step_gradnorm"], tol=1e-4, name="gradnorm")
    A1 = np.load(os.path.join(golden_dir, "small_model_after1.npz"))
    for k in P:
        p1, _, _ = O.adamw_update(P[k].detach(), grads[k] * coef, torch.zeros_like(grads[k]),
                                  torch.zeros_like(grads[k]), step=1, lr=5e-4)
        close(p1, torch.from_numpy(A1[k]), tol=2e-5, name="adamw " + k)


def test_three_steps_loss_curve(vec, small, golden_dir):
    """loss-curve equivalence on identical batches: 3 optimiser steps."""
    P, cfg = small
    P = {k: v.clone() for k, v in P.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    b = batch_of(vec)
    use_point = torch.ones(b["atoms"].shape[0], dtype=torch.bool)
    losses = []
    for step in range(1, 4):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        loss, *_ = O.step_loss(Pg, cfg, b, use_point)
        loss.backward()
        grads = {k: (Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])) for k in Pg}
        _, coef = O.clip_grad_norm(grads, 10.0)
        for k in P:
            P[k], M[k], V[k] = O.adamw_update(P[k], grads[k] * coef, M[k], V[k], step=step, lr=5e-4)
        losses.append(float(loss))
    close(torch.tensor(losses), vec["step_losses"], tol=2e-5, name="losses")
    A3 = np.load(os.path.join(golden_dir, "small_model_after3.npz"))
    for k in P:
        close(P[k], torch.from_numpy(A3[k]), tol=1e-4, name="after3 " + k)


def test_mid_curve_step_from_the_references_weights(small, golden_dir):
    """tests/golden/loss_curve_mid.npz: the reference's weights at the start of step 21 of its 40-step curve and the gradients
    it computes from them (not an envelope along a trajectory: one step from pinned weights)."""
    _, cfg = small
    m = np.load(os.path.join(golden_dir, "loss_curve_mid.npz"))
    c = np.load(os.path.join(golden_dir, "loss_curve.npz"))
    step = int(m["step"])
    P = {k[2:]: torch.from_numpy(m[k]).clone().requires_grad_(True) for k in m.files if k.startswith("w.")}
    b = {k: torch.from_numpy(c[f"b{step % 8}_{k}"]) for k in ("raw_tokens", "tokens", "atoms", "coords", "y_next")}
    loss, ar, cl, _ = O.step_loss(P, cfg, b, torch.ones(b["atoms"].shape[0], dtype=torch.bool))
    close(loss, m["loss"], name="mid loss")
    assert abs(float(m["loss"]) - float(c["loss"][step])) < 1e-6 * abs(float(m["loss"]))   # it IS the curve's step
    loss.backward()
    grads = {k: (P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])) for k in P}
    for k in P:
        close(grads[k], torch.from_numpy(m["g." + k]), tol=1e-4, name="mid grad " + k)
    norm, _ = O.clip_grad_norm(grads, 10.0)
    close(norm, m["gradnorm"], tol=1e-4, name="mid gradnorm")


def test_sim_bf16_is_close_to_fp32(vec, small):
    """The bf16-storage simulation stays within the tolerance the GPU tests use vs fp32."""
    P, cfg = small
    b = batch_of(vec)
    use_point = torch.ones(b["atoms"].shape[0], dtype=torch.bool)
    l32, *_ = O.step_loss(P, cfg, b, use_point)
    with O.sim_bf16():
        l16, *_ = O.step_loss(P, cfg, b, use_point)
    assert abs(float(l32) - float(l16)) < 3e-2 * abs(float(l32))


def test_block_head_size_32(vec):
    """The COATI2-size transformer shape (n_embd / n_head = 32): the oracle's RotaryBlock restatement against the
    reference's RotaryBlock (forward, attention output, rope tables, input gradient)."""
    P = {k[len("hs32_"):].replace("__", "."): v for k, v in vec.items() if k.startswith("hs32_") and "__" in k}
    P = {"b." + k: v for k, v in P.items()}
    cos, sin = O.rope_tables(24, 32)
    close(cos, vec["hs32_cos"][:24], name="cos32")
    close(sin, vec["hs32_sin"][:24], name="sin32")
    x = vec["hs32_x"].clone().requires_grad_(True)
    a1 = torch.nn.functional.layer_norm(x, (128,), P["b.ln_1.weight"], P["b.ln_1.bias"], 1e-5)
    close(O.attention(a1, P, "b.attn.", 4, cos, sin), vec["hs32_attn"], name="attn32")
    y = O.block(x, P, "b.", 4, cos, sin)
    close(y, vec["hs32_y"], name="block32")
    (y * vec["hs32_gy"]).sum().backward()
    close(x.grad, vec["hs32_dx"], name="dx32")
