"""BASELINE.md section 3 protocol for the reported CPU baseline, run once per round on the GPU box's host cores:
the oracle (fp32 torch restatement of the reference step, validated against the reference's golden vectors) on the
config-1 shape (grande architecture, B = 64, T = 128, A = 16, V = 10 322), >= 3 warm-up steps, median of >= 10 timed
steps, with torch.set_num_threads(N) for N = all cores and N = 8.  Prints one JSON object (profiles/rNN_cpu_baseline_full.json).

    python tools/cpu_baseline_full.py [--steps 10] [--warmup 3] [--batch 64] [--seq 128]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import GRANDE  # noqa: E402
from coati_amd.synthetic import make_batch  # noqa: E402
from oracle import coati_oracle as O  # noqa: E402


def run(threads, batch, up, steps, warmup):
    torch.set_num_threads(threads)
    cfg = O.OracleConfig(**GRANDE)
    P = O.init_params(cfg, seed=0)
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    times = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        Pg = {k: v.detach().requires_grad_(True) for k, v in P.items()}
        loss, *_ = O.step_loss(Pg, cfg, batch, up)
        loss.backward()
        grads = {k: (Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])) for k in Pg}
        _, coef = O.clip_grad_norm(grads, 10.0)
        for k in P:
            P[k], M[k], V[k] = O.adamw_update(P[k], grads[k] * coef, M[k], V[k], step=i + 1, lr=5e-4)
        dt = time.perf_counter() - t0
        if i >= warmup:
            times.append(dt)
    med = statistics.median(times)
    return {"threads": threads, "median_s_per_step": round(med, 3), "molecules_per_s": round(batch["tokens"].shape[0] / med, 3),
            "steps": steps, "warmup": warmup, "min_s": round(min(times), 3), "max_s": round(max(times), 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--atoms", type=int, default=16)
    a = ap.parse_args()
    batch, up = make_batch(a.batch, a.seq, a.atoms, GRANDE["n_tok"], seed=1234)
    ncores = os.cpu_count()
    out = {"what": "oracle (fp32 CPU port of the reference step: fwd + InfoNCE + AR CE + bwd + clip-norm + AdamW), BASELINE.md section 3 protocol",
           "shape": {"batch": a.batch, "seq_len": a.seq, "atoms": a.atoms, "n_tok": GRANDE["n_tok"], "arch": "grande d=256 L=16 + E3GNN 256x5"},
           "host_logical_cpus": ncores, "runs": []}
    for n in (torch.get_num_threads(), 8):
        out["runs"].append(run(n, batch, up, a.steps, a.warmup))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
