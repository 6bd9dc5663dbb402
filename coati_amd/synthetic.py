"""Synthetic SMILES-token + 3D-coordinate batches in the format clip_ar_xform emits (clip_e2e.py:288-330), as
SURVEY.md section 8(d) specifies them.  Used by bench.py, the smoke test and the parity tests (there is no network
for the 340 GB dataset and rdkit is absent, so tokens are drawn directly)."""
import torch

PAD, STOP, SMILES, SUFFIX, MIDDLE, UNK, CLIP = 0, 1, 2, 5, 6, 7, 8
ELEMENTS = (1, 6, 7, 8, 9, 16, 17)


def y_next_from_tokens(tokens):
    """tail of clip_ar_xform, clip_e2e.py:317-329"""
    y = torch.zeros_like(tokens)
    y[:, : tokens.shape[1] - 1] = tokens[:, 1:]
    for t in (CLIP, PAD, UNK, SUFFIX, MIDDLE):
        y[y == t] = -1
    return y


def packed_rows(raw_tokens, tokens, y_next=None):
    """Host-side row counts of the packed layout (include/coati_hip.h, coati_engine_forward rows1 / rows2): per row
    1 + the last position that holds a non-[PAD] token (or, for `tokens`, a target that is not -1), summed over the batch.
    The batch assembler calls this while the tokens are still on the host; on device tensors it costs one synchronisation."""
    l1, l2 = packed_lengths(raw_tokens, tokens, y_next)
    return int(l1.sum().item()), int(l2.sum().item())


def packed_lengths(raw_tokens, tokens, y_next=None):
    """Per-row lengths of the packed layout: [B] int32 for each of the two passes (their sums are packed_rows)."""
    def length(tok, y):
        live = tok != PAD
        if y is not None:
            live = live | (y >= 0)
        T = tok.shape[1]
        return (live.to(torch.int32) * torch.arange(1, T + 1, device=tok.device, dtype=torch.int32)).amax(dim=1)
    return length(raw_tokens, None), length(tokens, y_next)


def attention_score_efficiency(lengths, block=16):
    """useful / computed score elements of causal attention on `block`-row granularity (csrc/attention16.hip: block = 16; the 32-row
    kernels of rounds 1-5: block = 32): a row of T positions needs T (T + 1) / 2 scores, the kernel evaluates nb (nb + 1) / 2 whole
    block x block tiles with nb = ceil(T / block).  lengths: per-row lengths of a pass (packed_lengths)."""
    t = torch.as_tensor(lengths).to(torch.int64)
    t = t[t > 0]
    nb = (t + block - 1) // block
    return float((t * (t + 1) // 2).sum()) / float((nb * (nb + 1) // 2).sum() * block * block)


def make_batch(B, T, A, V, seed=1234, n_special=1596, p_clip=0.9, p_bad=0.01, min_len=16, device="cpu", with_rows=False):
    """with_rows: add batch["rows"] = CPU int64 [rows1, rows2], the packed-row counts (packed_rows above) -- Engine.train_step
    then runs the transformer passes on the rows' real prefixes only."""
    g = torch.Generator().manual_seed(seed)
    n_special = min(n_special, V - 2)
    Lmax = T - 4                      # [CLIP][UNK][SMILES] body [STOP]
    lo = min(min_len, Lmax)
    lens = torch.randint(lo, Lmax + 1, (B,), generator=g)
    lens[0] = Lmax                    # keeps the truncated width at T
    body = torch.randint(n_special, V, (B, Lmax), generator=g)
    clip_row = torch.rand(B, generator=g) < p_clip
    clip_row[0] = True
    bad = torch.rand(B, generator=g) < p_bad
    bad[0] = False
    raw = torch.zeros(B, T, dtype=torch.long)
    tok = torch.zeros(B, T, dtype=torch.long)
    ar = torch.arange(Lmax).unsqueeze(0)
    bm = ar < lens.unsqueeze(1)
    raw[:, 0] = SMILES
    raw[:, 1:1 + Lmax] = torch.where(bm, body, torch.zeros_like(body))
    raw[torch.arange(B), 1 + lens] = STOP
    t3 = torch.zeros(B, T, dtype=torch.long)
    t3[:, 0], t3[:, 1], t3[:, 2] = CLIP, UNK, SMILES
    t3[:, 3:3 + Lmax] = torch.where(bm, body, torch.zeros_like(body))
    t3[torch.arange(B), 3 + lens] = STOP
    t1 = torch.zeros(B, T, dtype=torch.long)
    t1[:, :T - 2] = raw[:, :T - 2]
    tok = torch.where(clip_row.unsqueeze(1), t3, t1)
    tok[bad] = 0
    raw[bad] = 0
    raw[bad, 0] = STOP
    raw = raw[:, : T - 2].contiguous()  # clip_ar_xform truncates every stack to its longest row
    el = torch.tensor(ELEMENTS)
    n_atoms = torch.randint(min(8, A), A + 1, (B,), generator=g)
    atoms = el[torch.randint(0, len(el), (B, A), generator=g)]
    atoms = torch.where(torch.arange(A).unsqueeze(0) < n_atoms.unsqueeze(1), atoms, torch.zeros_like(atoms))
    coords = (torch.randn(B, A, 3, generator=g) * 1.5).float()
    batch = dict(raw_tokens=raw, tokens=tok, y_next=y_next_from_tokens(tok), atoms=atoms, coords=coords)
    use_point = torch.rand(B, generator=g) > 0.5
    rows = packed_rows(raw, tok, batch["y_next"])        # on the host, before the upload (what a data loader does)
    batch = {k: v.to(device) for k, v in batch.items()}
    if with_rows:
        batch["rows"] = torch.tensor(rows, dtype=torch.int64)     # stays on the host
    return batch, use_point.to(device)
