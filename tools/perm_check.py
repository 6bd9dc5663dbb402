"""Batch-permutation check of the forward at full size (which output differs, and by how much)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coati_amd.engine import Engine, ModelConfig
from coati_amd.synthetic import make_batch
DEV = "cuda:0"
GRANDE = dict(n_layer_e3gnn=5, n_layer_xformer=16, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256,
              n_head=16, n_seq=250, n_tok=10322)
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
batch, up = make_batch(1024, 80, 16, GRANDE["n_tok"], seed=77)
batch = {k: v.to(DEV) for k, v in batch.items()}; up = up.to(DEV)
def fwd(b, u):
    h_e, h_s, bad = eng.forward(b["raw_tokens"], b["tokens"], b["atoms"], b["coords"], u, y_next=b["y_next"], train=False)
    return h_e.clone(), h_s.clone()
he0, hs0 = fwd(batch, up)
he0b, hs0b = fwd(batch, up)
print("same batch twice: he equal", torch.equal(he0, he0b), "hs equal", torch.equal(hs0, hs0b), float((hs0 - hs0b).abs().max()))
perm = torch.randperm(1024, generator=torch.Generator().manual_seed(1)).to(DEV)
pb = {k: v[perm].contiguous() for k, v in batch.items()}
he1, hs1 = fwd(pb, up[perm].contiguous())
print("permuted: he equal", torch.equal(he1, he0[perm]), "hs equal", torch.equal(hs1, hs0[perm]),
      "max |dhs|", float((hs1 - hs0[perm]).abs().max()), "rows differing", int(((hs1 - hs0[perm]).abs().amax(1) > 0).sum()))
