"""The grande 20-step curve (tests/golden/grande_golden.npz) replayed by the ORACLE on the CPU: fp32, with bf16 storage
simulated where the engine rounds, and fp32 from weights perturbed by 1e-3 -- how sensitive is each step's gradient norm to
operand rounding, independently of any kernel?  (same purpose as tools/curve_bf16_sim.py for the toy curve)
    python tools/curve_bf16_sim_grande.py [n_steps] [--envelope]
--envelope (round 6): also writes tests/golden/grande_curve_envelope.json -- per step, the largest gradient-norm deviation from the
reference's curve over the bf16-storage-simulating oracle and three fp32 oracles started from weights perturbed by 1e-3 (seeds 0-2),
and the same for the median over the parameters of the per-parameter gradient-norm deviation at the mid-curve step.  The GPU test
(tests/test_gpu_grande.py) bounds the engine by 2 x that envelope instead of by a constant: the envelope is what rounding alone does
to this trajectory, independently of any kernel."""
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
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(_pos[0]) if _pos else 14
ENVELOPE = "--envelope" in sys.argv
MID = int(g["mid_step"])


def run(sim, perturb=0.0, seed=0):
    P = {k: v.clone() for k, v in P0.items()}
    if perturb:
        gen = torch.Generator().manual_seed(seed)
        P = {k: v * (1 + perturb * torch.randn(v.shape, generator=gen)) for k, v in P.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    rec = dict(loss=[], gradnorm=[])
    mid_dev = None
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
        if step == MID:
            ref = g["mid_grad_norms"]
            mid_dev = float(np.median([abs(float(grads[n_].double().norm()) - ref[i]) / ref[i] for i, n_ in enumerate(names) if ref[i] > 0]))
        for k in P:
            if "coord_mlp" in k:
                continue
            P[k], M[k], V[k] = O.adamw_update(P[k], grads[k] * coef, M[k], V[k], step=step + 1, lr=5e-4)
        rec["loss"].append(float(loss.detach())); rec["gradnorm"].append(float(norm))
    out = {k: np.abs(np.array(v) - g["curve_" + k][:N]) / np.abs(g["curve_" + k][:N]) for k, v in rec.items()}
    out["mid_median"] = mid_dev
    return out


print("reference gradnorm per step:", " ".join("%.3f" % x for x in g["curve_gradnorm"][:N]))
runs = [("fp32 oracle", dict(sim=False)), ("bf16-sim oracle", dict(sim=True)), ("fp32, weights perturbed 1e-3", dict(sim=False, perturb=1e-3))]
if ENVELOPE:
    runs += [("fp32, weights perturbed 1e-3, seed 1", dict(sim=False, perturb=1e-3, seed=1)), ("fp32, weights perturbed 1e-3, seed 2", dict(sim=False, perturb=1e-3, seed=2))]
env = []
for tag, kw in runs:
    d = run(**kw)
    if tag != "fp32 oracle":
        env.append(d)
    if d["mid_median"] is not None:
        print("   per-parameter gradient norms at step %d: median deviation %.3e" % (MID, d["mid_median"]))
    print("%-30s loss dev max %.2e   gradnorm dev max %.2e (step %d) median %.2e" % (tag, d["loss"].max(), d["gradnorm"].max(), d["gradnorm"].argmax(), np.median(d["gradnorm"])))
    print("   gradnorm dev per step:", " ".join("%.1e" % x for x in d["gradnorm"]), flush=True)

if ENVELOPE:
    import json
    out = {"note": "written by tools/curve_bf16_sim_grande.py --envelope: what operand rounding / a 1e-3 weight perturbation alone do to the reference's "
                   "20-step grande curve (oracle runs on the CPU; no kernel involved)",
           "runs": [t for t, _ in runs[1:]], "n_steps": N, "mid_step": MID,
           "gradnorm_dev_max_per_step": [float(max(d["gradnorm"][i] for d in env)) for i in range(N)],
           "loss_dev_max_per_step": [float(max(d["loss"][i] for d in env)) for i in range(N)],
           "mid_median_max": float(max(d["mid_median"] for d in env if d["mid_median"] is not None))}
    with open(os.path.join(ROOT, "tests", "golden", "grande_curve_envelope.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tests/golden/grande_curve_envelope.json")
