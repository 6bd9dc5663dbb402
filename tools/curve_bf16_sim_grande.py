"""The grande 20-step curve (tests/golden/grande_golden.npz) replayed by the ORACLE on the CPU: fp32, with bf16 storage
simulated where the engine rounds, and fp32 from weights perturbed by 1e-3 -- how sensitive is each step's gradient norm to
operand rounding, independently of any kernel?  (same purpose as tools/curve_bf16_sim.py for the toy curve)
    python tools/curve_bf16_sim_grande.py [n_steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import coati_oracle as O  # noqa: E402
from tests import grande_util as GU  # noqa: E402

torch.set_num_threads(min(32, os.cpu_count() or 1))
g, ocfg, P0, names, batches, masks = GU.load(os.path.join(ROOT, "tests", "golden"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 14


def run(sim, perturb=0.0, seed=0):
    P = {k: v.clone() for k, v in P0.items()}
    if perturb:
        gen = torch.Generator().manual_seed(seed)
        P = {k: v * (1 + perturb * torch.randn(v.shape, generator=gen)) for k, v in P.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    rec = dict(loss=[], gradnorm=[])
    for step in range(N):
        b = batches[step % 4]
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        if sim:
            with O.sim_bf16():
                loss, *_ = O.step_loss(Pg, ocfg, b, masks[step])
        else:
            loss, *_ = O.step_loss(Pg, ocfg, b, masks[step])
        loss.backward()
        grads = {k: (Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])) for k in Pg}
        norm, coef = O.clip_grad_norm(grads, 10.0)
        for k in P:
            if "coord_mlp" in k:
                continue
            P[k], M[k], V[k] = O.adamw_update(P[k], grads[k] * coef, M[k], V[k], step=step + 1, lr=5e-4)
        rec["loss"].append(float(loss.detach())); rec["gradnorm"].append(float(norm))
    return {k: np.abs(np.array(v) - g["curve_" + k][:N]) / np.abs(g["curve_" + k][:N]) for k, v in rec.items()}


print("reference gradnorm per step:", " ".join("%.3f" % x for x in g["curve_gradnorm"][:N]))
for tag, kw in (("fp32 oracle", dict(sim=False)), ("bf16-sim oracle", dict(sim=True)), ("fp32, weights perturbed 1e-3", dict(sim=False, perturb=1e-3))):
    d = run(**kw)
    print("%-30s loss dev max %.2e   gradnorm dev max %.2e (step %d) median %.2e" % (tag, d["loss"].max(), d["gradnorm"].max(), d["gradnorm"].argmax(), np.median(d["gradnorm"])))
    print("   gradnorm dev per step:", " ".join("%.1e" % x for x in d["gradnorm"]), flush=True)
