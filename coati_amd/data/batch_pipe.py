"""Batch assembly in front of the training step.

Host side -- `stack_batch` / `get_mod_from_str` mirror coati/data/batch_pipe.py:9-76 (same names, arguments and output
dict): ragged per-molecule `atoms [n_i]` / `coords [n_i, 3]` arrays are zero-padded to the longest molecule of the batch.
Device side -- `device_tail` is the tensorisation tail of clip_ar_xform (clip_e2e.py:288-330) as two HIP kernels
(`coati_batch_ncols`, `coati_batch_tail`): count the occupied token columns, then truncate + build `y_next` with the five
masked special ids, all on the GPU (one 8-byte device->host read per batch for the two column counts)."""
import hashlib
from typing import Any, Dict, List

import numpy as np


def stack_batch(rows: List[Dict[str, Any]], return_coords=True, return_grads=False, return_dipole=False):
    """rows: list of dicts, optionally with 'atoms' [n] and 'coords' [n,3] (or flat [3n]); every other key is
    collected into an object array.  Returns {'atoms': [B, Amax] float64, 'coords': [B, Amax, 3] float64, ...}
    exactly like the reference (zeros for molecules without atoms)."""
    batch: Dict[str, Any] = {}
    if return_coords:
        n = len(rows)
        amax = max([int(r["atoms"].shape[0]) if "atoms" in r else 0 for r in rows]) if n else 0
        atoms = np.zeros((n, amax))
        coords = np.zeros((n, amax, 3))
        grads = np.zeros((n, amax, 3)) if return_grads else None
        dipoles = np.zeros((n, 3)) if return_dipole else None
        for i, r in enumerate(rows):
            if "atoms" not in r:
                continue
            a = np.asarray(r["atoms"])
            c = np.asarray(r["coords"])
            if return_grads and "gradients" in r:
                g = np.asarray(r["gradients"])
                grads[i, : g.shape[0], :] = g
            if return_dipole and "dipole" in r:
                dipoles[i, :] = np.asarray(r["dipole"])
            atoms[i, : a.shape[0]] = a
            if c.ndim == 1:                       # flat [3n] coordinates (reference: the except branch, :47-52)
                c = c.reshape((-1, 3), order="C")
            coords[i, : c.shape[0], :] = c
        batch["atoms"] = atoms
        batch["coords"] = coords
        if return_grads:
            batch["gradients"] = grads
        if return_grads and return_dipole:
            batch["dipoles"] = dipoles
    keys: List[str] = []
    for r in rows:
        for k in r:
            if k not in keys:
                keys.append(k)
    for k in keys:
        if k in batch:
            continue
        col = np.empty(len(rows), dtype=object)
        for i, r in enumerate(rows):
            col[i] = r.get(k, np.nan)
        batch[k] = col
    return batch


def get_mod_from_str(x: str, divisor: int = 100_000) -> int:
    """md5-based shard id of a SMILES string (batch_pipe.py:75-76; rank r keeps rows with id % world == r)."""
    return int.from_bytes(hashlib.md5(x.encode("utf-8")).digest(), "little") % divisor


def shard_rows(rows: List[Dict[str, Any]], rank: int, world: int, key: str = "smiles"):
    """The reference's per-rank row filter (batch_pipe.py:114-123)."""
    return [r for r in rows if get_mod_from_str(r[key], world) == rank]


def device_tail(tokens, raw_tokens, tokenizer):
    """tokens / raw_tokens: [B, n_seq] int64 CUDA tensors, zero-padded.  Returns (tokens[:, :n1], raw_tokens[:, :n2],
    y_next[:, :n1]) as fresh contiguous tensors, computed by the HIP kernels (no CPU fallback)."""
    import torch
    from .. import _lib
    from ..ops import ptr, stream
    if not (tokens.is_cuda and raw_tokens.is_cuda):
        raise RuntimeError("device_tail needs CUDA tensors (the HIP path has no CPU fallback)")
    B, S = tokens.shape
    tokens = tokens.contiguous()
    raw_tokens = raw_tokens.contiguous()
    ncols = torch.zeros(2, device=tokens.device, dtype=torch.int32)
    _lib.call("coati_batch_ncols", ptr(tokens), B, S, ptr(ncols), stream())
    _lib.call("coati_batch_ncols", ptr(raw_tokens), B, raw_tokens.shape[1], ptr(ncols[1:]), stream())
    n1, n2 = (int(v) for v in ncols.tolist())
    masked = torch.tensor([tokenizer.clip_token, tokenizer.pad_token, tokenizer.unk_token, tokenizer.suffix_token,
                           tokenizer.middle_token], device=tokens.device, dtype=torch.int64)
    tok_c = torch.empty(B, n1, device=tokens.device, dtype=torch.int64)
    y_c = torch.empty(B, n1, device=tokens.device, dtype=torch.int64)
    raw_c = torch.empty(B, n2, device=tokens.device, dtype=torch.int64)
    if n1 > 0:
        _lib.call("coati_batch_tail", ptr(tokens), B, S, n1, ptr(tok_c), ptr(y_c), ptr(masked), 5, stream())
    if n2 > 0:
        _lib.call("coati_batch_tail", ptr(raw_tokens), B, raw_tokens.shape[1], n2, ptr(raw_c), None, ptr(masked), 5, stream())
    return tok_c, raw_c, y_c
