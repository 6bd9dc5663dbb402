import os

import torch

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "test_report.txt")


def log(msg):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(msg + "\n")


def relerr(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    scale = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / scale


def check(name, a, b, tol):
    assert a.shape == b.shape, (name, a.shape, b.shape)
    bad = (~torch.isfinite(a.detach().float().cpu())).sum().item()
    e = relerr(a, b)
    log(f"{name:60s} relerr {e:.3e}  tol {tol:.1e}  nonfinite {bad}  {'OK' if (e <= tol and bad == 0) else 'FAIL'}")
    assert bad == 0, f"{name}: {bad} non-finite values"
    assert e <= tol, f"{name}: rel err {e:.3e} > {tol:.1e}"


def rbf(x):
    return x.bfloat16().float()
