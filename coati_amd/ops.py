"""Tensor-level wrappers of the fine-grained C-ABI operators.  Every function enqueues HIP kernels on the current
torch stream and returns torch tensors; nothing here computes on the CPU or falls back to aten."""
import ctypes
import math

import torch

from . import _lib

EPI_BF16, EPI_F32, EPI_RES_F32, EPI_GELU, EPI_DGELU, EPI_SILU, EPI_DSILU, EPI_ACC_F32 = range(8)
EPI_GELU_GRAD, EPI_MUL_AUX = 12, 13
BF16 = torch.bfloat16


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("coati_amd ops need device tensors: the HIP path has no CPU fallback")


def gemm_nt(A, W, bias=None, epi=EPI_BF16, aux_in=None, out=None, n_store=0):
    """C = epilogue(A @ W^T + bias).  A [M,K] bf16|f32, W [N,K] bf16.  Returns C or (C, pre) for GELU/SILU, (C, NewGELU'(pre)) for GELU_GRAD."""
    _need_cuda(A, W)
    M, K = A.shape
    N = W.shape[0]
    a_f32 = 1 if A.dtype == torch.float32 else 0
    out_f32 = epi in (EPI_F32, EPI_RES_F32, EPI_ACC_F32)
    ncol = max(N, n_store)
    if out is None:
        out = torch.empty(M, ncol, device=A.device, dtype=torch.float32 if out_f32 else BF16)
    aux_out = None
    ld_aux = 0
    if epi in (EPI_GELU, EPI_SILU, EPI_GELU_GRAD):
        # EPI_GELU_GRAD saves NewGELU' as 8-bit fixed point (dequantise with dq8 below); EPI_MUL_AUX reads that format
        aux_out = torch.empty(M, N, device=A.device, dtype=torch.uint8 if epi == EPI_GELU_GRAD else BF16)
        ld_aux = aux_out.stride(0)
    if aux_in is not None:
        ld_aux = aux_in.stride(0)
    _lib.call("coati_gemm_nt", ptr(A), a_f32, A.stride(0), ptr(W), W.stride(0), M, N, K, ptr(out), out.stride(0),
              n_store, ptr(bias), ptr(aux_in), ptr(aux_out), ld_aux, epi, stream())
    return (out, aux_out) if aux_out is not None else out


def gemm_lnbwd(dY, WT, x, mean, rstd, gamma, dres, want16=True, chain_W=None):
    """dy = dY @ WT^T is the gradient w.r.t. the output of LayerNorm(x; gamma): returns (dx f32 = dres + LN-backward(dy), dx16 bf16 | None,
    dgamma, dbeta) -- the ring GEMM with the LayerNorm backward in its write-out (coati_gemm_lnbwd)."""
    import ctypes
    _need_cuda(dY, WT, x)
    M, K = dY.shape
    dx = torch.empty(M, 256, device=dY.device, dtype=torch.float32)
    dx16 = torch.empty(M, 256, device=dY.device, dtype=BF16) if want16 else None
    partial = torch.zeros(256, 512, device=dY.device, dtype=torch.float32)
    n = ctypes.c_int32(0)
    chain_C = torch.empty(M, 256, device=dY.device, dtype=BF16) if chain_W is not None else None
    _lib.call("coati_gemm_lnbwd", ptr(dY), dY.stride(0), ptr(WT), WT.stride(0), M, K, ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres),
              ptr(dx), ptr(dx16), ptr(partial), ctypes.byref(n), ptr(chain_W), ptr(chain_C), stream())
    s = partial[: n.value].sum(0)
    if chain_W is not None:        # + chain_C = dx16 @ chain_W^T, computed in the same launch
        return dx, dx16, s[:256], s[256:], chain_C
    return dx, dx16, s[:256], s[256:]


def mlp_fwd(x, gamma, beta, W1, b1, W2, b2):
    """out = x + c_proj(NewGELU(c_fc(ln_2(x)))) in one launch (coati_mlp_fwd): returns (out f32, a2 bf16, mean, rstd, g bf16, codes u8)."""
    _need_cuda(x, W1, W2)
    M = x.shape[0]
    dev = x.device
    a2 = torch.empty(M, 256, device=dev, dtype=BF16)
    mean = torch.empty(M, device=dev, dtype=torch.float32)
    rstd = torch.empty(M, device=dev, dtype=torch.float32)
    g = torch.empty(M, 1024, device=dev, dtype=BF16)
    codes = torch.empty(M, 1024, device=dev, dtype=torch.uint8)
    out = torch.empty(M, 256, device=dev, dtype=torch.float32)
    _lib.call("coati_mlp_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(a2), ptr(mean), ptr(rstd), ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(g), ptr(codes),
              ptr(out), M, stream())
    return out, a2, mean, rstd, g, codes


def quant_mx8(x):
    """rows of bf16 / f32 x [M, K] -> (q [M, K] uint8 holding OCP e4m3, scales [M, K / 32] uint8 holding E8M0): MXFP8 blocks of 32 along k"""
    _need_cuda(x)
    M, K = x.shape
    q = torch.empty(M, K, device=x.device, dtype=torch.uint8)
    sc = torch.empty(M, K // 32, device=x.device, dtype=torch.uint8)
    _lib.call("coati_quant_mx8", ptr(x), 1 if x.dtype == torch.float32 else 0, x.stride(0), ptr(q), q.stride(0), ptr(sc), M, K, stream())
    return q, sc


def gemm_mx8(Aq, As, Wq, Ws, bias=None, epi=EPI_BF16, aux_in=None, out=None):
    """C = epilogue(dequant(Aq, As) @ dequant(Wq, Ws)^T + bias) on the block-scaled fp8 matrix core instruction (K % 128 == 0)"""
    _need_cuda(Aq, Wq)
    M, K = Aq.shape
    N = Wq.shape[0]
    out_f32 = epi in (EPI_F32, EPI_RES_F32, EPI_ACC_F32)
    if out is None:
        out = torch.empty(M, N, device=Aq.device, dtype=torch.float32 if out_f32 else BF16)
    aux_out, ld_aux = None, 0
    if epi == EPI_GELU_GRAD:
        aux_out = torch.empty(M, N, device=Aq.device, dtype=torch.uint8)
        ld_aux = aux_out.stride(0)
    if aux_in is not None:
        ld_aux = aux_in.stride(0)
    _lib.call("coati_gemm_mx8", ptr(Aq), Aq.stride(0), ptr(As), ptr(Wq), Wq.stride(0), ptr(Ws), M, N, K, ptr(out), out.stride(0),
              ptr(bias), ptr(aux_in), ptr(aux_out), ld_aux, epi, stream())
    return (out, aux_out) if aux_out is not None else out


def dq8(q):
    """value of the 8-bit fixed-point codes the forward MLP saves for NewGELU' (csrc/common.h: q / 200 - 0.13)"""
    return q.float() / 200.0 - 0.13


def q8(d):
    """codes of values in [-0.13, 1.145] (round to nearest, saturating)"""
    return torch.clamp(torch.round(d.float() * 200.0 + 26.0), 0, 255).to(torch.uint8)


def wgrad(A, Bm, dW, dbias=None, n_out=0):
    """dW[N,K] += A[M,N]^T @ Bm[M,K]; dbias[N] += colsum(A)."""
    _need_cuda(A, Bm, dW)
    M, N = A.shape
    K = Bm.shape[1]
    _lib.call("coati_wgrad", ptr(A), 1 if A.dtype == torch.float32 else 0, A.stride(0), ptr(Bm), Bm.stride(0), M, N, K,
              ptr(dW), dW.stride(0), ptr(dbias), n_out, stream())
    return dW


def wgrad_grouped(problems, tile_size=256):
    """problems: list of (A[M,N] bf16, Bm[M,K] bf16, dW[N,K] f32, dbias[N] f32), all with the same M; one launch, no atomics
    on dW: dW += A^T @ Bm, dbias += colsum(A)."""
    import ctypes
    n = len(problems)
    for A, Bm, dW, db in problems:
        _need_cuda(A, Bm, dW, db)
        assert A.dtype == torch.bfloat16 and Bm.dtype == torch.bfloat16 and dW.dtype == torch.float32 and db.dtype == torch.float32
    M = problems[0][0].shape[0]
    PP, LL, II = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    Ns, Ks = II(*[q[0].shape[1] for q in problems]), II(*[q[1].shape[1] for q in problems])
    nbytes = int(_lib.lib().coati_wgrad_grouped_workspace_bytes(n, Ns, Ks, tile_size))
    ws = torch.empty(max(nbytes, 1), device=problems[0][0].device, dtype=torch.uint8)   # the caller owns the tile table
    _lib.call("coati_wgrad_grouped", n,
              PP(*[ptr(q[0]) for q in problems]), LL(*[q[0].stride(0) for q in problems]),
              PP(*[ptr(q[1]) for q in problems]), LL(*[q[1].stride(0) for q in problems]), M,
              Ns, Ks,
              PP(*[ptr(q[2]) for q in problems]), LL(*[q[2].stride(0) for q in problems]),
              PP(*[ptr(q[3]) for q in problems]), tile_size, ptr(ws), nbytes, stream())
    return ws


def sgemm(A, Bm, trans_a=False, trans_b=False, bias=None, alpha=1.0, out=None, accumulate=False):
    """out = alpha * op(A) @ op(Bm) (+bias) in exact fp32 (MFMA f32)."""
    _need_cuda(A, Bm)
    ars, acs = (A.stride(1), A.stride(0)) if trans_a else (A.stride(0), A.stride(1))
    brs, bcs = (Bm.stride(1), Bm.stride(0)) if trans_b else (Bm.stride(0), Bm.stride(1))
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    N = Bm.shape[0] if trans_b else Bm.shape[1]
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    _lib.call("coati_sgemm", ptr(A), ars, acs, ptr(Bm), brs, bcs, ptr(out), out.stride(0), M, N, K, ptr(bias),
              float(alpha), 1 if accumulate else 0, stream())
    return out


def layernorm_fwd(x, gamma=None, beta=None, want16=True, want32=False):
    _need_cuda(x)
    M, C = x.shape
    y16 = torch.empty(M, C, device=x.device, dtype=BF16) if want16 else None
    y32 = torch.empty(M, C, device=x.device, dtype=torch.float32) if want32 else None
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    _lib.call("coati_layernorm_fwd", ptr(x), x.stride(0), ptr(gamma), ptr(beta), ptr(y16), C, ptr(y32), C, ptr(mean),
              ptr(rstd), M, C, stream())
    return y16, y32, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma=None, dres=None, x_is_xhat=False, want_affine_grads=True, want16=False,
                  two_stage=True):
    _need_cuda(dy, x)
    M, C = x.shape
    dx = torch.empty(M, C, device=x.device, dtype=torch.float32)
    dg = torch.zeros(C, device=x.device, dtype=torch.float32) if (gamma is not None and want_affine_grads) else None
    db = torch.zeros(C, device=x.device, dtype=torch.float32) if dg is not None else None
    dx16 = torch.empty(M, C, device=x.device, dtype=BF16) if want16 else None
    partial = torch.empty(2048, 2 * C, device=x.device, dtype=torch.float32) if (dg is not None and two_stage) else None
    _lib.call("coati_layernorm_bwd", ptr(dy), 1 if dy.dtype == torch.float32 else 0, dy.stride(0), ptr(x), x.stride(0),
              1 if x_is_xhat else 0, ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx), ptr(dx16), ptr(dg), ptr(db), ptr(partial), M, C, stream())
    return (dx, dg, db, dx16) if want16 else (dx, dg, db)


def rope_tables(n_seq, head_size=16, base=10000.0, device="cuda"):
    """RotaryEmbedding cos/sin caches (reference basic_transformer.py:57-69), computed with torch on the host in the
    same op order as the reference so the tables are bit-identical, then moved to the device once."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_size, 2).float() / head_size))
    t = torch.arange(n_seq).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().contiguous().to(device), emb.sin().contiguous().to(device)


def gemm_qkv_rope(A, W, bias, T, cos, sin, hs=16):
    """qkv = [RoPE(q) | RoPE(k) | v] of A @ W^T + bias (rows are (b, t) with t = row % T); head size hs = 16 or 32."""
    _need_cuda(A, W)
    M, C = A.shape
    out = torch.empty(M, 3 * C, device=A.device, dtype=BF16)
    _lib.call("coati_gemm_qkv_rope_hs", ptr(A), A.stride(0), ptr(W), W.stride(0), ptr(bias), M, C, ptr(out), 3 * C, ptr(cos),
              ptr(sin), T, hs, stream())
    return out


def attn_fwd(qkv, B, T, n_head, hs=16):
    """causal attention on already-rotated q,k (head size 16 or 32)"""
    _need_cuda(qkv)
    C = n_head * hs
    y = torch.empty(B * T, C, device=qkv.device, dtype=BF16)
    lse = torch.empty(B, n_head, T, device=qkv.device, dtype=torch.float32)
    _lib.call("coati_attn_fwd_hs", ptr(qkv), ptr(y), ptr(lse), B, T, n_head, hs, stream())
    return y, lse


def attn_bwd(qkv, y, dy, lse, B, T, n_head, cos, sin, hs=16):
    dqkv = torch.empty_like(qkv)
    dscratch = torch.empty(B, n_head, T, device=qkv.device, dtype=torch.float32)
    _lib.call("coati_attn_bwd_hs", ptr(qkv), ptr(y), ptr(dy), ptr(lse), ptr(dscratch), ptr(dqkv), ptr(cos), ptr(sin), B, T, n_head, hs, stream())
    return dqkv


def seq_pack(tok, y=None, pad=0, rows=None):
    """row map of the packed layout of a padded [B, T] token matrix (coati_seq_pack): off [B + 1], row_src, row_t [rows], ypk, err"""
    B, T = tok.shape
    if rows is None:
        from .synthetic import packed_rows
        rows = packed_rows(tok, tok, y)[1] if y is not None else packed_rows(tok, tok)[0]
    dev = tok.device
    off = torch.empty(B + 1, device=dev, dtype=torch.int32)
    row_src = torch.empty(rows, device=dev, dtype=torch.int32)
    row_t = torch.empty(rows, device=dev, dtype=torch.int32)
    ypk = torch.empty(rows, device=dev, dtype=torch.int64) if y is not None else None
    err = torch.zeros(4, device=dev, dtype=torch.int32)
    _lib.call("coati_seq_pack", ptr(tok), ptr(y), pad, B, T, rows, ptr(off), ptr(row_src), ptr(row_t), ptr(ypk), ptr(err), stream())
    return off, row_src, row_t, ypk, err


def attn_fwd_varlen(qkv, off, B, T, n_head, hs=16):
    """causal attention on packed rows: sequence b = rows off[b] .. off[b + 1] of qkv (at most T tokens)"""
    _need_cuda(qkv)
    y = torch.zeros(qkv.shape[0], n_head * hs, device=qkv.device, dtype=BF16)
    lse = torch.zeros(B, n_head, T, device=qkv.device, dtype=torch.float32)
    _lib.call("coati_attn_fwd_varlen", ptr(qkv), ptr(y), ptr(lse), ptr(off), B, T, n_head, hs, stream())
    return y, lse


def attn_bwd_varlen(qkv, y, dy, lse, off, B, T, n_head, cos, sin, hs=16):
    dqkv = torch.zeros_like(qkv)
    dscratch = torch.empty(B, n_head, T, device=qkv.device, dtype=torch.float32)
    _lib.call("coati_attn_bwd_varlen", ptr(qkv), ptr(y), ptr(dy), ptr(lse), ptr(dscratch), ptr(dqkv), ptr(cos), ptr(sin), ptr(off), B, T, n_head, hs, stream())
    return dqkv


def embed_fwd(idx, table, injection=None, unk=7):
    B, T = idx.shape
    V, C = table.shape
    x = torch.empty(B * T, C, device=table.device, dtype=torch.float32)
    _lib.call("coati_embed_fwd", ptr(idx), ptr(table), ptr(injection), unk, ptr(x), B, T, C, V, stream())
    return x


def embed_bwd(idx, dx, V, with_injection=False, unk=7):
    B, T = idx.shape
    C = dx.shape[1]
    dt = torch.zeros(V, C, device=dx.device, dtype=torch.float32)
    di = torch.zeros(B, C, device=dx.device, dtype=torch.float32) if with_injection else None
    _lib.call("coati_embed_bwd", ptr(idx), ptr(dx), ptr(dt), ptr(di), unk, B, T, C, V, stream())
    return dt, di


def ce_fwd(a, W, target):
    """Fused lm_head + cross entropy forward.  a [M,K] bf16, W [V,K] bf16, target [M] int64 (-1 ignored).
    Returns (lse [M], scal[16]) with scal[0] = sum loss, scal[1] = count."""
    M, K = a.shape
    V = W.shape[0]
    tiles = (V + 127) // 128
    partial = torch.empty(M, tiles, 2, device=a.device, dtype=torch.float32)
    lse = torch.empty(M, device=a.device, dtype=torch.float32)
    scal = torch.zeros(16, device=a.device, dtype=torch.float32)
    _lib.call("coati_gemm_ce_partial", ptr(a), a.stride(0), ptr(W), W.stride(0), M, V, K, ptr(partial), stream())
    _lib.call("coati_ce_finish", ptr(partial), tiles, ptr(a), a.stride(0), ptr(W), W.stride(0), ptr(target), ptr(lse),
              ptr(scal), M, K, V, stream())
    return lse, scal


def ce_bwd(a, W, target, lse, scal):
    M, K = a.shape
    V = W.shape[0]
    Vpad = (V + 63) // 64 * 64
    d = torch.empty(M, Vpad, device=a.device, dtype=BF16)
    _lib.call("coati_gemm_ce_bwd", ptr(a), a.stride(0), ptr(W), W.stride(0), M, V, K, ptr(d), Vpad, Vpad, ptr(lse),
              ptr(target), ptr(scal), stream())
    return d


def attn_groups(off, B, T):
    """work list of attn_block_fwd: groups of whole consecutive sequences with <= 128 rows (off: seq_pack's offsets or None)"""
    dev = off.device if off is not None else torch.device("cuda")
    grp = torch.zeros(B + 2, device=dev, dtype=torch.int32)
    _lib.call("coati_attn_groups", ptr(off), B, T, ptr(grp), stream())
    return grp


def attn_block_fwd(x, ln_g, ln_b, Wqkv, bqkv, Wproj, bproj, cos, sin, B, T, off=None, row_src=None):
    """xmid = x + c_proj(attention(RoPE(c_attn(ln_1(x))))) in one launch; returns (xmid, a1, mean, rstd, qkv, y, lse)"""
    _need_cuda(x)
    M, C = x.shape
    dev = x.device
    grp = attn_groups(off, B, T)
    xmid = torch.zeros(M, C, device=dev, dtype=torch.float32)
    a1 = torch.zeros(M, C, device=dev, dtype=BF16)
    mean = torch.zeros(M, device=dev, dtype=torch.float32)
    rstd = torch.zeros(M, device=dev, dtype=torch.float32)
    qkv = torch.zeros(M, 3 * C, device=dev, dtype=BF16)
    y = torch.zeros(M, C, device=dev, dtype=BF16)
    lse = torch.zeros(B, 16, T, device=dev, dtype=torch.float32)
    _lib.call("coati_attn_block_fwd", ptr(x), ptr(xmid), ptr(ln_g), ptr(ln_b), ptr(mean), ptr(rstd), ptr(a1), ptr(Wqkv), ptr(bqkv),
              ptr(Wproj), ptr(bproj), ptr(qkv), ptr(y), ptr(lse), ptr(cos), ptr(sin), ptr(row_src), ptr(grp), T, M, stream())
    return xmid, a1, mean, rstd, qkv, y, lse, grp
