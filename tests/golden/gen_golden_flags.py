"""
Golden vectors for the constructor flags of e3gnn_smiles_clip_e2e the grande configuration does not exercise
(clip_e2e.py:405-437, 454-463), produced by IMPORTING THE REFERENCE in the build container (stubs of gen_golden.py):

  case "doargs" : norm_clips=False, token_mlp=False, use_point_encoder=True   -- the reference's own do_args() defaults
                  (train_coati.py:520-523): plain Linear heads, nn.Identity special-token map
  case "nopoint": norm_clips=False, token_mlp=False, use_point_encoder=False  -- encode_points returns zeros; the point
                  encoder and point_to_clip never receive a gradient
  case "mixed"  : norm_clips=True,  token_mlp=False, use_point_encoder=True
  case "mlp_nopoint": norm_clips=True, token_mlp=True, use_point_encoder=False
  case "nobias" : the grande flags with biases=False -- the blocks' four Linear layers without bias (basic_transformer.py:113-115, 166-168)
  case "normembed": the grande flags with norm_embed=True -- LayerNorm behind the token embedding (basic_transformer.py:72-76), the
                  injection overwrites its output; + the registered-but-unused xformer.norm_embed module (smiles_xformer.py:81-84)
  case "torchemb": the grande flags with torch_emb=True -- node features from nn.Embedding(84, H), embedding = Identity
                  (e3gnn_clip.py:49-56, 74-77, 113-115)
  case "oldarch": the grande flags with old_architecture=True -- point_to_clip / smiles_to_clip = Linear -> LayerNorm (clip_e2e.py:409-417)
  case "residual": the grande flags with residual=True -- the one-hot node features as a third input of every node MLP
                  (e3gnn_clip.py:97-100, e_gcl_sparse.py:141, 282-290)

Per case (small model of gen_golden.py: d = 64, 2 + 2 layers, V = 48; batch of 5 rows incl. a bad row): the weights, forward_dist
with a mixed injection mask (h_e3gnn, h_smiles, logits, bad_rows), the training step (train_coati.py:216-277: ar / clip / total
loss, every parameter gradient -- None gradients recorded as "nograd." names --, clip_grad_norm_ value) and the weights after
one AdamW step (lr 5e-4, wd 0.1, betas (0.9, 0.99); every 13th element of each tensor).

    python tests/golden/gen_golden_flags.py            # (re)write tests/golden/flags_golden.npz
    python tests/golden/gen_golden_flags.py --verify   # regenerate into a scratch directory and compare contents
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("GOLDEN_OUT", HERE)
sys.path.insert(0, HERE)

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


def main():
    import gen_golden as G   # inserts the stubs, imports the reference
    ref_clip = G.ref_clip
    out = {}
    raw, tok, atoms, coords = G.synth_batch(5, 18, 6, 48, seed=31)
    y = G.y_next(tok)
    out.update(b_raw_tokens=raw, b_tokens=tok, b_atoms=atoms, b_coords=coords, b_y_next=y)
    mixed = torch.tensor([0.9, 0.1, 0.7, 0.2, 0.6])
    out["use_point"] = mixed > 0.5
    tokz = G.Tok(48, 24)
    teu = float(np.log(float(48)) / np.log(2.0))
    for ci, (case, flags) in enumerate(CASES.items()):
        torch.manual_seed(100 + ci)
        kw = dict(G.SMALL)
        kw.update(flags)
        model = ref_clip.e3gnn_smiles_clip_e2e(**kw)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() == 1:
                    p.add_(0.05 * torch.randn_like(p))
        sd = {k: v.clone() for k, v in model.state_dict().items() if not k.endswith(".attn.bias")}
        out.update({f"{case}.w.{k}": v for k, v in sd.items()})
        real_rand = torch.rand
        torch.rand = lambda *a, **k: mixed.clone()
        try:
            he, hs, lg, bad = model.forward_dist(raw, tok, atoms, coords, tokz, p_clip_emb_smi=0.5)
            out.update({f"{case}.h_e3gnn": he, f"{case}.h_smiles": hs, f"{case}.logits": lg, f"{case}.bad": bad})
            opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.1, betas=(0.9, 0.99), eps=1e-8)
            opt.zero_grad()
            he, hs, lg, bad = model.forward_dist(raw, tok, atoms, coords, tokz, p_clip_emb_smi=0.5)
        finally:
            torch.rand = real_rand
        ar = torch.nn.functional.cross_entropy(lg.view(-1, lg.size(-1)), y.view(-1), ignore_index=-1)
        c = model.clip_loss(hs, he, bad).mean()
        loss = ar + c * teu
        loss.backward()
        for n, p in model.named_parameters():
            if p.grad is None:
                out[f"{case}.nograd.{n}"] = np.zeros(1, dtype=np.float32)
            else:
                out[f"{case}.grad.{n}"] = p.grad.clone()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        out.update({f"{case}.ar": ar, f"{case}.clip": c, f"{case}.loss": loss, f"{case}.gradnorm": gn})
        opt.step()
        # (every 13th element of each tensor: the update is element-wise)
        out.update({f"{case}.after1.{k}": v.reshape(-1)[::13].clone() for k, v in model.state_dict().items() if not k.endswith(".attn.bias")})
    np.savez_compressed(os.path.join(OUT, "flags_golden.npz"), **G.npify(out))


def verify():
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, GOLDEN_OUT=tmp), check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        x, y = np.load(os.path.join(tmp, "flags_golden.npz")), np.load(os.path.join(HERE, "flags_golden.npz"))
        ok = x.files == y.files and all(np.array_equal(x[k], y[k]) and x[k].dtype == y[k].dtype for k in x.files)
        print(("same     " if ok else "DIFFERENT") + " flags_golden.npz")
        return ok


if __name__ == "__main__":
    if "--verify" in sys.argv:
        sys.exit(0 if verify() else 1)
    main()
