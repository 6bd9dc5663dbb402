"""Shared by the grande-architecture parity tests (tests/test_oracle_grande.py, tests/test_gpu_grande.py): loads
tests/golden/grande_golden.npz (written by tests/golden/gen_golden_grande.py from the imported reference) and regenerates
the 20.4 M weights it was produced with."""
import os

import numpy as np
import torch

GRANDE = dict(n_layer_e3gnn=5, n_layer_xformer=16, n_hidden_xformer=256, n_hidden_e3nn=256, n_embd_common=256, n_head=16,
              n_seq=250, n_tok=10322)     # examples/training/train_grande.py:21-35 with the may_closedparen vocabulary size


def projection(name, numel):
    """the fixed +-1 vector of tests/golden/gen_golden_grande.py: projection"""
    s = 0
    for ch in name:
        s = (s * 131 + ord(ch)) % 2147483647
    return (torch.randint(0, 2, (numel,), generator=torch.Generator().manual_seed(s)) * 2 - 1).float()


def load(golden_dir):
    from oracle import coati_oracle as O
    z = np.load(os.path.join(golden_dir, "grande_golden.npz"))
    g = {k: z[k] for k in z.files}
    ocfg = O.OracleConfig(**GRANDE)
    P = O.init_params(ocfg, seed=int(g["seed"]))
    names = [str(n) for n in g["names"]]
    # the generator of the weights must be the one the fixture was made with: per-parameter checksums
    ws = np.array([float(P[n].double().sum()) for n in names])
    wa = np.array([float(P[n].double().abs().sum()) for n in names])
    assert np.allclose(ws, g["wsum"], rtol=0, atol=1e-6 * np.abs(g["wabs"]).max()) and np.allclose(wa, g["wabs"], rtol=1e-9), \
        "init_params(seed) no longer reproduces the weights grande_golden.npz was generated with"
    batches = [{k: torch.from_numpy(g[f"b{i}_{k}"]) for k in ("raw_tokens", "tokens", "y_next", "atoms", "coords")} for i in range(4)]
    masks = torch.from_numpy(g["rand"]) > 0.5        # use_point = rand > p_clip_emb_smi (clip_e2e.py:802-808)
    return g, ocfg, P, names, batches, masks
