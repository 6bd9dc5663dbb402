"""Build-container check (needs /root/reference): a checkpoint document written by THIS package -- serialize_model +
optimizer_state_dict over the flat AdamW buffers -- is read by the reference's own unpickler, model class, load_state_dict
and torch.optim.AdamW.load_state_dict, and the reference continues training from it (its next loss equals the golden
loss of step 2).  Called by tests/test_checkpoint_xform_cpu.py in a subprocess; prints REFERENCE READ OK."""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np
import torch

GOLD = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLD))
sys.path.insert(0, ROOT)


def main():
    # this package first (its `coati` alias is not used below), then the reference (which takes over the name `coati`)
    from tests.test_checkpoint_xform_cpu import FlatHost, SMALL, _load_doc
    from coati_amd.training.train_coati import serialize_model, optimizer_state_dict, load_optimizer_state
    from coati_amd.models.encoding.clip_e2e import reference_parameter_order
    doc_ref = _load_doc()
    eng = FlatHost(SMALL)
    load_optimizer_state(eng, doc_ref["optimizer"], list(doc_ref["model"].keys()))
    z = np.load(os.path.join(GOLD, "small_model_after1.npz"))
    sd = {k: torch.from_numpy(z[k]) for k in reference_parameter_order(list(z.files))}
    for k in [m for m in sys.modules if m == "coati" or m.startswith("coati.")]:
        del sys.modules[k]
    sys.meta_path[:] = [f for f in sys.meta_path if type(f).__name__ != "_AliasFinder"]
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(GOLD, "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(gg)          # stubs rdkit / boto3 and imports the reference
        blob = serialize_model({"tokenizer_vocab": "mar"}, {"dataset_type": "x"}, sd, dict(gg.SMALL),
                               optimizer_state_dict(eng, lr=5e-4, weight_decay=0.1), n_toks_processed=7, n_grads_processed=3,
                               offline_loss={})
        from coati.models.io.coati import CPU_Unpickler as RefUnpickler
        import coati
        assert coati.__file__.startswith("/root/reference"), coati.__file__
        doc = RefUnpickler(io.BytesIO(blob), encoding="UTF-8").load()
        model = gg.ref_clip.e3gnn_smiles_clip_e2e(**doc["model_kwargs"])
    res = model.load_state_dict(doc["model"], strict=False)
    assert not res.unexpected_keys and all(k.endswith(".attn.bias") for k in res.missing_keys), res
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.1, betas=(0.9, 0.99), eps=1e-8)
    opt.load_state_dict(doc["optimizer"])
    p0 = next(iter(model.parameters()))
    assert torch.equal(opt.state[p0]["exp_avg"], doc_ref["optimizer"]["state"][0]["exp_avg"])
    v = np.load(os.path.join(GOLD, "small_vectors.npz"))
    b = {k: torch.from_numpy(v["b_" + k]) for k in ("raw_tokens", "tokens", "atoms", "coords", "y_next")}
    losses = []
    for step in range(2):       # the reference's steps 2 and 3, resumed from our document
        opt.zero_grad()
        he, hs, lg, bad = model.forward_dist(b["raw_tokens"], b["tokens"], b["atoms"], b["coords"], gg.Tok(48, 24), p_clip_emb_smi=0.0)
        ar = torch.nn.functional.cross_entropy(lg.view(-1, lg.size(-1)), b["y_next"].view(-1), ignore_index=-1)
        loss = ar + gg.ref_clip.clip_loss()(hs, he, bad).mean() * float(v["teu"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        losses.append(float(loss))
    ref = [float(x) for x in v["step_losses"][1:3]]
    assert all(abs(a - r) <= 1e-5 * abs(r) for a, r in zip(losses, ref)), (losses, ref)
    A3 = np.load(os.path.join(GOLD, "small_model_after3.npz"))
    worst = max(float((p.detach() - torch.from_numpy(A3[n])).abs().max()) for n, p in model.named_parameters())
    assert worst <= 1e-6, worst
    print("REFERENCE READ OK: losses", losses, "weights after step 3 within", worst)


if __name__ == "__main__":
    main()
