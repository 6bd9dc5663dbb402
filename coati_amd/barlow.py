"""Barlow-Twins contrastive head (BASELINE.json configs[3] `barlow_closed`) on the HIP kernels.

The reference repository holds no Barlow code (only checkpoint names), so this follows the Barlow-Twins formulation;
parity is UNPINNED and the only check is the oracle's own restatement (oracle.barlow_loss).  Exchange pattern at
world size > 1: all-reduce of the per-dimension batch statistics (2 x 2E floats, twice: mean, then
centred second moment), of the E x E
cross-correlation matrix, and of the 2 x 2E backward statistics -- no embedding all-gather is needed.  FOUR dependent collectives
per step ([column sums | valid-row count] in one buffer, the centred second moments, C, the backward statistics); the second one stays
because the variance is taken of rows centred on the GLOBAL mean (a one-pass E[z^2] - E[z]^2 lost 1e-4 to cancellation and made the
result depend on the number of ranks).  All scratch lives in one buffer per (B, E) on the caller's engine-side cache."""
import torch

from . import _lib
from .ops import ptr, stream, sgemm


_SCRATCH = {}


def _default_all_reduce(t):
    from .distributed import all_reduce_sum
    return all_reduce_sum(t)


def barlow_head(h_s: torch.Tensor, h_e: torch.Tensor, bad: torch.Tensor, lam: float = 5e-3, gscale: float = 1.0,
                distributed: bool = False, all_reduce=None):
    """Returns (loss[1] on device, dL/dh_s, dL/dh_e); gradients are multiplied by gscale.
    all_reduce: in-place sum-over-ranks callable (default: coati_amd.distributed.all_reduce_sum)."""
    _all_reduce = all_reduce or _default_all_reduce
    B, E = h_s.shape
    dev = h_s.device
    f32 = dict(device=dev, dtype=torch.float32)
    bad = bad.to(torch.uint8).contiguous()
    # one scratch buffer per (B, E, device), reused every step (was ~ 15 allocations per step): small statistics first, then the four
    # [B, E] row buffers.  The outputs (loss, dS, dC) are fresh tensors: the caller keeps them across the backward.
    import threading
    key = (B, E, str(dev), threading.get_ident())       # (per thread: the two-rank emulation of the tests runs both ranks in one process)
    scr = _SCRATCH.get(key)
    if scr is None:
        scr = _SCRATCH[key] = torch.empty(4 * E + 2 + 4 * E + 2 * E + 4 * E + 4 * B * E, **f32)
        if len(_SCRATCH) > 8:
            _SCRATCH.pop(next(iter(_SCRATCH)))
    o = 0

    def take(n, shape):
        nonlocal o
        t = scr[o:o + n].view(shape)
        o += n
        return t
    sc = take(4 * E + 2, (4 * E + 2,))                   # [column sums of h_s, h_e (2 x 2E) | count, 1 / count]: ONE collective
    stats, cnt = sc[:4 * E].view(2, 2 * E), sc[4 * E:]
    _lib.call("coati_count_valid", ptr(bad), B, ptr(cnt[0:1]), ptr(cnt[1:2]), stream())
    # two-pass batch statistics: global mean first, then the variance of the CENTRED rows (E[z^2] - E[z]^2 loses the
    # variance to cancellation when |mean| >> sigma; the result then depended on how many ranks the sums were split over)
    _lib.call("coati_colsum2", ptr(h_s), None, ptr(bad), ptr(stats[0]), B, E, stream())
    _lib.call("coati_colsum2", ptr(h_e), None, ptr(bad), ptr(stats[1]), B, E, stream())
    if distributed:
        _all_reduce(sc[:4 * E + 1])
    stats2, rs, m = take(4 * E, (2, 2 * E)), take(2 * E, (2, E)), take(4 * E, (2, 2 * E))
    cs, ce, zs, ze = (take(B * E, (B, E)) for _ in range(4))
    _lib.call("coati_center_rows", ptr(h_s), ptr(bad), ptr(stats[0]), ptr(cnt), ptr(cs), B, E, stream())
    _lib.call("coati_center_rows", ptr(h_e), ptr(bad), ptr(stats[1]), ptr(cnt), ptr(ce), B, E, stream())
    _lib.call("coati_colsum2", ptr(cs), None, ptr(bad), ptr(stats2[0]), B, E, stream())
    _lib.call("coati_colsum2", ptr(ce), None, ptr(bad), ptr(stats2[1]), B, E, stream())
    if distributed:
        _all_reduce(stats2)
    _lib.call("coati_standardize", ptr(cs), ptr(bad), ptr(stats2[0]), ptr(cnt), ptr(zs), ptr(rs[0]), B, E, stream())
    _lib.call("coati_standardize", ptr(ce), ptr(bad), ptr(stats2[1]), ptr(cnt), ptr(ze), ptr(rs[1]), B, E, stream())
    C = sgemm(zs, ze, trans_a=True)                      # [E,E] raw cross-correlation of the local rows
    if distributed:
        _all_reduce(C)
    loss = torch.zeros(1, **f32)
    _lib.call("coati_barlow_dc", ptr(C), ptr(cnt), float(lam), ptr(loss), E, stream())   # C -> G = dL/dC / n
    dzs = sgemm(ze, C, trans_b=True)                     # dL/dzs~ = Ze~ G^T
    dze = sgemm(zs, C)                                   # dL/dze~ = Zs~ G
    _lib.call("coati_colsum2", ptr(dzs), ptr(zs), ptr(bad), ptr(m[0]), B, E, stream())
    _lib.call("coati_colsum2", ptr(dze), ptr(ze), ptr(bad), ptr(m[1]), B, E, stream())
    if distributed:
        _all_reduce(m)
    dS, dC = torch.empty(B, E, **f32), torch.empty(B, E, **f32)
    _lib.call("coati_standardize_bwd", ptr(dzs), ptr(zs), ptr(bad), ptr(rs[0]), ptr(m[0]), ptr(cnt), float(gscale), ptr(dS), B, E, stream())
    _lib.call("coati_standardize_bwd", ptr(dze), ptr(ze), ptr(bad), ptr(rs[1]), ptr(m[1]), ptr(cnt), float(gscale), ptr(dC), B, E, stream())
    return loss, dS, dC
