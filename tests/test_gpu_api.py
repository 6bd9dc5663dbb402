"""The reference-shaped Python API on the GPU: model mirror (constructor kwargs, state_dict keys, forward_dist,
encode_*), the trainer entry point with its checkpoint document, and the loader."""
import os
import pickle
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.gpu_util import check, log  # noqa: E402

SMALL = dict(n_layer_e3gnn=2, n_layer_xformer=2, n_hidden_xformer=64, n_hidden_e3nn=64, n_embd_common=64, n_head=4,
             n_seq=24, n_tok=48, biases=True, torch_emb=False, residual=False, norm_clips=True, norm_embed=False,
             token_mlp=True)


class Tok:
    pad_token, stop_token, smiles_token, suffix_token, middle_token, unk_token, clip_token = 0, 1, 2, 5, 6, 7, 8
    vocab = {"[UNK]": 7, "[STOP]": 1, "[PAD]": 0}
    n_token, n_seq = 48, 24
    keys = list(range(48))


def test_model_mirror_against_golden(golden_dir):
    from coati.models.encoding.clip_e2e import e3gnn_smiles_clip_e2e
    z = np.load(os.path.join(golden_dir, "small_model.npz"))
    v = np.load(os.path.join(golden_dir, "small_vectors.npz"))
    model = e3gnn_smiles_clip_e2e(**SMALL, device=torch.device("cuda:0"))
    sd_keys = set(model.state_dict().keys())
    assert set(z.files) <= sd_keys
    assert {k for k in sd_keys - set(z.files)} == {f"xformer.transformer.h.{l}.attn.bias" for l in range(2)}
    missing = model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=False)
    assert not missing.unexpected_keys
    assert model.embed_dim == 64 and hasattr(model, "point_encoder") and hasattr(model, "smiles_to_clip")
    b = {k: torch.from_numpy(v["b_" + k]) for k in ("raw_tokens", "tokens", "atoms", "coords")}
    he, hs, logits, bad = model.forward_dist(b["raw_tokens"], b["tokens"], b["atoms"], b["coords"], Tok(), p_clip_emb_smi=0.0)
    check("api h_e3gnn", he.cpu(), torch.from_numpy(v["fd_p0_h_e3gnn"]), 6.5e-3)
    check("api logits", logits.cpu(), torch.from_numpy(v["fd_p0_logits"]), 6.5e-3)
    assert bad.dtype == torch.bool and torch.equal(bad.cpu(), torch.from_numpy(v["fd_p0_bad"]))
    check("api encode_tokens", model.encode_tokens(b["raw_tokens"], Tok()).cpu(), torch.from_numpy(v["fd_p0_h_smiles"]), 6.5e-3)
    check("api encode_points", model.encode_points(b["atoms"], b["coords"]).cpu(), torch.from_numpy(v["fd_p0_h_e3gnn"]), 6.5e-3)
    a, c, badr = torch.from_numpy(v["cl_a"]), torch.from_numpy(v["cl_b"]), torch.from_numpy(v["cl_bad"])
    check("api clip_loss", model.clip_loss(a.cuda(), c.cuda(), badr.cuda()).cpu(), torch.from_numpy(v["cl_l1"]), 1e-5)
    raw_bad = b["raw_tokens"].clone()
    raw_bad[0][raw_bad[0] == 1] = 0
    with pytest.raises(RuntimeError, match="stop tokens"):
        model.forward_dist(raw_bad, b["tokens"], b["atoms"], b["coords"], Tok())


def test_trainer_and_checkpoint_roundtrip(tmp_path):
    from coati.training.train_coati import train_autoencoder, do_args
    from coati.models.io.coati import load_e3gnn_smiles_clip_e2e
    from coati.data.dataset import COATI_dataset
    from coati_amd.data.dataset import SyntheticTokenizer
    args = do_args([])
    args.nodes, args.nr, args.gpus, args.world_size = 1, 0, 1, 1
    args.n_layer_e3gnn, args.n_hidden_e3nn, args.n_hidden_xformer, args.n_embd_common = 2, 64, 64, 64
    args.n_layer_xformer, args.n_head, args.max_n_seq, args.n_seq = 2, 4, 40, 24
    args.norm_clips, args.token_mlp = True, True
    args.batch_size, args.n_epochs, args.lr, args.test_interval = 16, 2, 5e-4, 1
    args.log_batch_loss, args.log_interval = 1, 1
    args.output_dir, args.model_dir, args.data_dir = str(tmp_path / "logs"), str(tmp_path / "ckpt"), str(tmp_path)
    args.run_name = "t"
    tk = SyntheticTokenizer(n_seq=24, n_token=200, n_special=12)
    ds = COATI_dataset(cache_dir=str(tmp_path), tokenizer=tk, n_batches=6, n_atoms=8)
    model = train_autoencoder(0, args, dataset=ds, tokenizer=tk)
    files = os.listdir(args.model_dir)
    assert len(files) == 1
    doc = pickle.load(open(os.path.join(args.model_dir, files[0]), "rb"))
    assert set(doc) >= {"train_args", "dataset_summary", "model", "optimizer", "model_kwargs", "n_toks_processed", "n_grads_processed"}
    recs = open(os.path.join(args.output_dir, "t", "log.json")).read().strip().split("\n")
    losses = [eval(r.rstrip(","), {"null": None})["value"] for r in recs if "train_batch_loss" in r]
    log(f"trainer losses {losses}")
    assert len(losses) == 12 and all(np.isfinite(losses)) and losses[-1] < losses[0]
    m2, _ = load_e3gnn_smiles_clip_e2e(os.path.join(args.model_dir, files[0]), device="cuda:0")
    for k, t in model.state_dict().items():
        assert torch.equal(t.cpu(), m2.state_dict()[k].cpu()), k


DEV = "cuda:0"


def test_device_batch_tail_matches_reference_golden(golden_dir):
    """truncate + y_next on the device (coati_batch_ncols / coati_batch_tail) against the reference's own
    clip_ar_xform output (tests/golden/xform_tail.npz) and against the host mirror on a synthetic batch."""
    import numpy as np
    from coati_amd.data.batch_pipe import device_tail
    from coati_amd.models.encoding.clip_e2e import tensorize_batch
    g = np.load(os.path.join(golden_dir, "xform_tail.npz"))

    class Tk:  # ids of the 'mar' vocabulary used by the golden generator (tests/golden/constants.json)
        pad_token, clip_token, unk_token, suffix_token, middle_token = 0, 8, 7, 5, 6

    n_seq = 40
    tok = torch.zeros(g["tokens"].shape[0], n_seq, dtype=torch.long)
    raw = torch.zeros_like(tok)
    tok[:, : g["tokens"].shape[1]] = torch.from_numpy(g["tokens"])
    raw[:, : g["raw_tokens"].shape[1]] = torch.from_numpy(g["raw_tokens"])
    t, r, y = device_tail(tok.to(DEV), raw.to(DEV), Tk)
    assert torch.equal(t.cpu(), torch.from_numpy(g["tokens"]))
    assert torch.equal(r.cpu(), torch.from_numpy(g["raw_tokens"]))
    assert torch.equal(y.cpu(), torch.from_numpy(g["y_next"]))
    # larger seeded batch: device tail == host mirror (bit-exact integer work)
    gen = torch.Generator().manual_seed(3)
    B, S = 257, 250
    lens = torch.randint(3, 90, (B,), generator=gen)
    tok = torch.zeros(B, S, dtype=torch.long)
    raw = torch.zeros(B, S, dtype=torch.long)
    for b in range(B):
        tok[b, : lens[b]] = torch.randint(0, 300, (int(lens[b]),), generator=gen)
        raw[b, : max(int(lens[b]) - 2, 1)] = torch.randint(1, 300, (max(int(lens[b]) - 2, 1),), generator=gen)
    tok[:, 0] = 8
    host = tensorize_batch({"tokens": tok.clone(), "raw_tokens": raw.clone(), "atoms": torch.zeros(B, 4, dtype=torch.long),
                            "coords": torch.zeros(B, 4, 3)}, Tk, device="cpu")
    t, r, y = device_tail(tok.to(DEV), raw.to(DEV), Tk)
    assert torch.equal(t.cpu(), host["tokens"]) and torch.equal(r.cpu(), host["raw_tokens"]) and torch.equal(y.cpu(), host["y_next"])


def test_reference_written_checkpoint_loads_and_resumes(golden_dir):
    """n1 against a document the REFERENCE wrote (ref_checkpoint_after1.pkl: its serialize_model + torch AdamW state,
    right after its first optimizer step): (i) the loader returns (model, TrieTokenizer) with exactly those weights and
    the model reproduces the reference's next loss; (ii) --resume_optimizer semantics: the per-parameter AdamW state
    mapped into the flat m / v lets the next two steps follow the reference's steps 2 and 3."""
    import json
    from coati.models.io.coati import load_e3gnn_smiles_clip_e2e, CPU_Unpickler
    from coati.models.encoding.tokenizers.trie_tokenizer import TrieTokenizer
    from coati.training.train_coati import load_optimizer_state
    path = os.path.join(golden_dir, "ref_checkpoint_after1.pkl")
    with open(os.path.join(golden_dir, "clip_ar_xform.json")) as f:
        g = json.load(f)
    model, tok = load_e3gnn_smiles_clip_e2e(path, device="cuda:0", freeze=False,
                                            vocab={"special_tokens": g["special"], "smiles_tokens": g["smiles_tokens"]})
    assert isinstance(tok, TrieTokenizer) and tok.n_seq == 24 and tok.stop_token == 1 and tok.vocab["[UNK]"] == 7
    A1 = np.load(os.path.join(golden_dir, "small_model_after1.npz"))
    sd = model.state_dict()
    for k in A1.files:
        assert torch.equal(sd[k].cpu(), torch.from_numpy(A1[k])), k
    v = np.load(os.path.join(golden_dir, "small_vectors.npz"))
    batch = {k: torch.from_numpy(v["b_" + k]).to(DEV) for k in ("raw_tokens", "tokens", "atoms", "coords", "y_next")}
    up = torch.ones(batch["atoms"].shape[0], dtype=torch.bool, device=DEV)
    eng = model.engine
    with open(path, "rb") as f:
        doc = CPU_Unpickler(f, encoding="UTF-8").load()
    load_optimizer_state(eng, doc["optimizer"], list(doc["model"].keys()))
    assert eng.step_count == 1
    losses = []
    for _ in range(2):
        eng.train_step(batch, up, lr=5e-4, weight_decay=0.1, max_norm=10.0)
        losses.append(eng.losses()["loss"])
    log(f"resumed from the reference's checkpoint: losses {losses} reference {v['step_losses'][1:3].tolist()}")
    check("resumed loss curve (reference steps 2, 3)", torch.tensor(losses), torch.from_numpy(v["step_losses"][1:3]).float(), 8e-4)
    A3 = np.load(os.path.join(golden_dir, "small_model_after3.npz"))
    sd = model.state_dict()
    for k in A1.files:
        a1, a3 = torch.from_numpy(A1[k]), torch.from_numpy(A3[k])
        if "coord_mlp" in k:
            assert torch.equal(sd[k].cpu(), a3) and torch.equal(a1, a3), k     # never touched, on either side
            continue
        d_hip, d_ref = (sd[k].cpu() - a1).flatten().double(), (a3 - a1).flatten().double()
        cos = float((d_hip @ d_ref) / (d_hip.norm() * d_ref.norm() + 1e-30))
        assert cos > 0.9, (k, cos)
