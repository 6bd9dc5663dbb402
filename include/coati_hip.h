/*
 * coati_hip.h -- C ABI of libcoati_hip.so, the MI355X (gfx950) implementation of COATI's contrastive +
 * autoregressive training step.
 *
 * The reference (terraytherapeutics/COATI) is pure Python on PyTorch: it has no FFI of its own.  The boundary
 * below is what a Python binding (ctypes) needs in order to replace the aten ops behind
 * the reference's own operator interface for this path:
 *
 *   e3gnn_smiles_clip_e2e.forward_dist                coati/models/encoding/clip_e2e.py:772-814
 *   e3gnn_clip.forward (+ e_gcl_sparse)               coati/models/encoding/e3gnn_clip.py:108-137,
 *                                                     coati/models/encoding/e_gcl_sparse.py:27-77,169-321
 *   RotarySmilesTransformer.xformer / encode /        coati/models/encoding/smiles_xformer.py:50-68,106-112,
 *     forward_with_replacement                          353-368,426-454
 *   RotaryBlock / RotarySelfAttention / NewGELU       coati/models/encoding/basic_transformer.py:12-28,83-174
 *   clip_loss.forward                                 coati/models/encoding/clip_e2e.py:27-47
 *   do_minibatch (loss, backward, clip, AdamW)        coati/training/train_coati.py:216-277
 *
 * Conventions
 *   - every function returns 0 on success or a negative code (COATI_EARG -1 bad argument, COATI_ESHAPE -2
 *     unsupported shape, COATI_EHIP -3 HIP error); coati_last_error() returns a thread-local message.
 *     Nothing throws across the boundary.
 *   - all pointers are DEVICE pointers unless a parameter says "host".  The caller (PyTorch) owns every buffer:
 *     parameters, gradients, optimiser state, activations/workspace.  The library never allocates device memory.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the call returns immediately.
 *   - bf16 tensors are raw uint16_t storage; "f32" = float; token / atom indices are int64_t (torch.long).
 *   - matrices are row-major with an explicit leading dimension (elements).
 */
#ifndef COATI_HIP_H
#define COATI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COATI_ABI_VERSION 5

const char* coati_last_error(void);
int coati_abi_version(void);

/* ------------------------------------------------------------------------------------------------------
 * Fine-grained operators (one HIP kernel or a short fixed sequence each).  These replace individual aten
 * ops of the reference and are what the per-op parity tests call.
 * ---------------------------------------------------------------------------------------------------- */

/* epilogue selector of coati_gemm_nt (values of enum CoatiEpi in csrc/kernels.h) */
enum {
  COATI_EPI_BF16 = 0, COATI_EPI_F32 = 1, COATI_EPI_RES_F32 = 2, COATI_EPI_GELU = 3, COATI_EPI_DGELU = 4,
  COATI_EPI_SILU = 5, COATI_EPI_DSILU = 6, COATI_EPI_ACC_F32 = 7,
  /* 8..11: internal to the engine (fused cross-entropy, GNN edge, QKV + RoPE) */
  COATI_EPI_GELU_GRAD = 12,   /* aux_out = NewGELU'(acc + bias), C = NewGELU(acc + bias): the MLP forward as the engine runs it */
  COATI_EPI_MUL_AUX = 13      /* C = acc * aux_in: its backward (input gradient of the activation) */
};

/* C[M,N] = epilogue(A[M,K] * B[N,K]^T + bias).  Replaces nn.Linear forward (F.linear) and its input-gradient,
 * fused with the activation / residual that follows it in basic_transformer.py:165-173 and e_gcl_sparse.py:130-145.
 * A is bf16 (a_f32=0) or f32 converted on load (a_f32=1); B is bf16.  K % 64 == 0. */
int coati_gemm_nt(const void* A, int a_f32, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K,
                  void* C, int64_t ldc, int n_store, const float* bias, const void* aux_in, void* aux_out,
                  int64_t ld_aux, int epi, void* stream);

#ifdef COATI_EXPERIMENTAL   /* operators of csrc/experimental/ (slower than what they would replace): only in libcoati_hip_x.so, built under COATI_AMD_EXPERIMENTAL=1 */
/* The MLP half of a transformer block as one launch (csrc/experimental/mlp64.hip): out = x + c_proj(NewGELU(c_fc(ln_2(x)))) for d = 256, hidden 1024
 * (basic_transformer.py:103-123, 165-169 forward).  x [M, 256] f32; W1 = c_fc.weight [1024, 256], W2 = c_proj.weight [256, 1024] (bf16);
 * also written: a2 = ln_2(x) (bf16), mean / rstd [M], g = NewGELU(c_fc(.)) [M, 1024] bf16, codes = NewGELU' as 8-bit fixed point [M, 1024]
 * -- what the backward reads.  Replaces nn.LayerNorm + two F.linear + NewGELU + the residual add.  24 577 .. 65 536 rows. */
int coati_mlp_fwd(const float* x, const float* gamma, const float* beta, uint16_t* a2, float* mean, float* rstd, const uint16_t* W1,
                  const float* b1, const uint16_t* W2, const float* b2, uint16_t* g, uint8_t* codes, float* out, int M, void* stream);
#endif

/* An input-gradient product whose result is the gradient w.r.t. a LayerNorm's OUTPUT, with that LayerNorm's backward in the
 * product's write-out (c_fc -> ln_2 and c_attn -> ln_1 of basic_transformer.py:162-174): dy = dY[M,K] WT[256,K]^T never visits
 * memory; dx[M,256] (f32) = dres + LayerNorm-backward(dy | x, mean, rstd, gamma) (dx may be dres: in place), dx16 (optional) its
 * bf16 copy, partial[*n_partial_rows][512] per-workgroup sums of dgamma | dbeta (add the rows up).  Replaces F.linear's input
 * gradient + nn.LayerNorm's backward.  256 columns, K % 64 == 0, 40 961 .. 57 344 rows (a packed batch; the engine runs the two
 * kernels separately elsewhere): COATI_ESHAPE otherwise.  partial must hold 256 x 512 floats.
 * chain_W [256, 256] / chain_C [M, 256] bf16 (optional, both or neither; needs dx16): a second product chained behind the LayerNorm
 * backward in the same launch, chain_C = dx16 chain_W^T -- c_proj's input gradient, which consumes ln_2's backward output
 * (basic_transformer.py:165-169 backward): the rows come back from L2 to the workgroup that has just written them. */
int coati_gemm_lnbwd(const uint16_t* dY, int64_t lda, const uint16_t* WT, int64_t ldw, int M, int K, const float* x, const float* mean,
                     const float* rstd, const float* gamma, const float* dres, float* dx, uint16_t* dx16, float* partial,
                     int32_t* n_partial_rows, const uint16_t* chain_W, uint16_t* chain_C, void* stream);

/* MXFP8 (OCP Microscaling: e4m3 elements, one E8M0 scale per 32 consecutive k) -- BASELINE.json configs[4] "fp8 MFMA GEMMs".
 * coati_quant_mx8: rows of bf16 (x_f32 = 0) or f32 x [M, K] -> q [M, K] e4m3 bytes + scales [M, K / 32] (K % 32 == 0; shared
 * exponent floor(log2 amax) - 8, saturating conversion).  coati_gemm_mx8: C[M, N] = epilogue(A W^T + bias) on
 * v_mfma_scale_f32_32x32x64_f8f6f4 (the matrix core applies the block scales), fp32 accumulation, K % 128 == 0, lda / ldw in
 * bytes, epilogues COATI_EPI_BF16 / F32 / RES_F32 / GELU_GRAD / MUL_AUX as coati_gemm_nt.  Replaces F.linear / its input
 * gradient in the d = 512 transformer block (simple_coati2/transformer_only.py:43) when the engine runs with fp8 = 1. */
int coati_quant_mx8(const void* x, int x_f32, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scales, int M, int K, void* stream);
int coati_gemm_mx8(const uint8_t* A, int64_t lda, const uint8_t* a_scales, const uint8_t* W, int64_t ldw, const uint8_t* w_scales,
                   int M, int N, int K, void* C, int64_t ldc, const float* bias, const void* aux_in, void* aux_out, int64_t ld_aux,
                   int epi, void* stream);

/* lm_head + cross-entropy without materialising logits (smiles_xformer.py:453 + train_coati.py:260-265):
 * partial[M, ceil(V/128)] receives per-tile (max, sum exp) pairs; coati_ce_finish merges them into lse[M] and
 * adds sum(lse - logit[target]) to scal[0] and the number of targets != -1 to scal[1]. */
int coati_gemm_ce_partial(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, int M, int V, int K,
                          void* partial, void* stream);
int coati_ce_finish(const void* partial, int tiles_n, const uint16_t* A, int64_t lda, const uint16_t* W,
                    int64_t ldw, const int64_t* target, float* lse, float* scal, int M, int K, int V, void* stream);
/* dlogits[M, n_store] (bf16) = (softmax(A W^T) - onehot(target)) / scal[1]; rows with target -1 are zero. */
int coati_gemm_ce_bwd(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, int M, int V, int K,
                      uint16_t* dlogits, int64_t ldd, int n_store, const float* lse, const int64_t* target,
                      const float* scal, void* stream);

/* dW[N,K] (f32, +=) = A[M,N]^T * B[M,K]; dbias[N] (+=) = column sums of A (may be NULL).  Replaces the weight /
 * bias gradient of every nn.Linear on the path.  Rows n >= n_out of dW are not written (n_out = 0 -> N). */
int coati_wgrad(const void* A, int a_f32, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K,
                float* dW, int64_t ldw, float* dbias, int n_out, void* stream);

/* The same for a LIST of problems that share M, in ONE launch and without fp32 atomics on dW: one workgroup per output tile
 * (tile_size 128 or 256; 256 needs every N and K to be a multiple of 256) streams all M rows of its tile.  bf16 A, dbias
 * required.  This is how the engine computes the 4 x n_layer Linear gradients of a transformer pass (the reference's
 * loss.backward() through RotaryBlock, basic_transformer.py:126-174).  The tile table is written into `workspace` (device
 * memory of the caller, coati_wgrad_grouped_workspace_bytes; it must stay untouched until the launch has run); nothing is
 * allocated and the stream is not synchronised. */
int64_t coati_wgrad_grouped_workspace_bytes(int n_problems, const int* N, const int* K, int tile_size);
int coati_wgrad_grouped(int n_problems, const uint16_t* const* A, const int64_t* lda, const uint16_t* const* B, const int64_t* ldb,
                        int M, const int* N, const int* K, float* const* dW, const int64_t* ldw, float* const* dbias,
                        int tile_size, void* workspace, int64_t workspace_bytes, void* stream);

/* exact-f32 GEMM with generic strides: C[M,N] = alpha * sum_k A[m*ars + k*acs] * B[k*brs + n*bcs] (+ bias[n])
 * (+ C when accumulate).  Used for the [B,256] projection heads and the InfoNCE logits (clip_e2e.py:36-37). */
int coati_sgemm(const float* A, int64_t ars, int64_t acs, const float* B, int64_t brs, int64_t bcs, float* C,
                int64_t ldc, int M, int N, int K, const float* bias, float alpha, int accumulate, void* stream);

/* nn.LayerNorm(C) / InstanceNorm1d applied over the hidden dim (gamma = beta = NULL), eps 1e-5.
 * backward: dx (f32, = dres + LN'(dy)) and optionally dx16, the same values rounded to bf16 for the next GEMMs;
 * dgamma/dbeta are ADDED to.  partial: optional [2048, 2C] f32 scratch -> deterministic two-stage reduction of the affine
 * gradients (without it they are accumulated with fp32 atomics, ~2x slower at C = 256). */
int coati_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, uint16_t* y16,
                        int64_t ld16, float* y32, int64_t ld32, float* mean, float* rstd, int M, int C,
                        void* stream);
int coati_layernorm_bwd(const void* dy, int dy_f32, int64_t lddy, const float* x, int64_t ldx, int x_is_xhat,
                        const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                        uint16_t* dx16, float* dgamma, float* dbeta, float* partial, int M, int C, void* stream);

/* QKV projection with the rotary embedding fused into the epilogue (basic_transformer.py:133-144, 83-100):
 * qkv[M, 3C] (bf16) = [RoPE(q) | RoPE(k) | v] of A W^T + bias; row m is token position m % T; head size 16. */
int coati_gemm_qkv_rope(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, int M, int C,
                        uint16_t* qkv, int64_t ldc, const float* cos_t, const float* sin_t, int T, void* stream);

/* RotarySelfAttention core (basic_transformer.py:145-150) for head size 16 on ALREADY ROTATED q,k: causal softmax(q k^T/4) v.
 * The backward returns gradients w.r.t. the un-rotated q,k (it applies the transposed rotation), i.e. w.r.t. A W^T + bias.
 * qkv [B*T, 3*nh*16] bf16, y [B*T, nh*16] bf16, lse [B, nh, T] f32, cos/sin [n_seq, 16] f32 (RotaryEmbedding
 * tables, basic_transformer.py:57-69).  dscratch: [B, nh, T] f32 scratch (row sums of dO*O) for the backward. */
int coati_attn_fwd(const uint16_t* qkv, uint16_t* y, float* lse, int B, int T, int n_head, void* stream);
int coati_attn_bwd(const uint16_t* qkv, const uint16_t* y, const uint16_t* dy, const float* lse, float* dscratch, uint16_t* dqkv,
                   const float* cos_t, const float* sin_t, int B, int T, int n_head, void* stream);
/* head-size-generic variants (head_size = 16 or 32; the un-suffixed entry points are head size 16): qkv / cache / rope
   tables as above with 16 replaced by head_size (rope tables [n_seq, head_size]) */
int coati_attn_fwd_hs(const uint16_t* qkv, uint16_t* y, float* lse, int B, int T, int n_head, int head_size, void* stream);
/* Packed rows (variable-length sequences without padding, see coati_engine_forward): coati_seq_pack derives the row map of
 * a padded token matrix tok [B, T] (y optional [B, T] targets): off [B + 1], row_src / row_t [rows_expect] int32, ypk
 * [rows_expect] int64 (optional), err |= 2 when the device-side row count differs from rows_expect.  The _varlen attention
 * entry points take seq_off = off: sequence b owns rows off[b] .. off[b + 1] of qkv / y / dy / dqkv; lse and dscratch keep
 * the padded [B, nh, T] layout. */
int coati_seq_pack(const int64_t* tok, const int64_t* y, int pad_token, int B, int T, int rows_expect, int32_t* off,
                   int32_t* row_src, int32_t* row_t, int64_t* ypk, int32_t* err, void* stream);
int coati_attn_fwd_varlen(const uint16_t* qkv, uint16_t* y, float* lse, const int32_t* seq_off, int B, int T, int n_head,
                          int head_size, void* stream);
int coati_attn_bwd_varlen(const uint16_t* qkv, const uint16_t* y, const uint16_t* dy, const float* lse, float* dscratch,
                          uint16_t* dqkv, const float* cos_t, const float* sin_t, const int32_t* seq_off, int B, int T,
                          int n_head, int head_size, void* stream);
#ifdef COATI_EXPERIMENTAL
/* The attention half of a RotaryBlock as ONE launch (csrc/experimental/attn_block.hip; basic_transformer.py:126-154, 171-172; d = 256, 16 heads of
 * 16, sequences of <= 128 rows): xmid = x + c_proj(causal_attention(RoPE(c_attn(ln_1(x))))).  x / xmid [M, 256] f32; a1 = ln_1(x)
 * [M, 256] bf16, mean / rstd [M], qkv [M, 768] bf16 (q, k rotated), y [M, 256] bf16 and lse [B, 16, T] f32 are the tensors the
 * backward reads (the same ones the three-launch path leaves).  row_src [M] (coati_seq_pack; null = padded layout).
 * coati_attn_groups builds the launch's work list on the device: grp [B + 2] int32 -- groups of whole consecutive sequences with
 * <= 128 rows (seq_off = coati_seq_pack's off, or null for the padded layout b * T). */
int coati_attn_groups(const int32_t* seq_off, int B, int T, int32_t* grp, void* stream);
int coati_attn_block_fwd(const float* x, float* xmid, const float* ln_g, const float* ln_b, float* mean, float* rstd, uint16_t* a1,
                         const uint16_t* Wqkv, const float* bqkv, const uint16_t* Wproj, const float* bproj, uint16_t* qkv, uint16_t* y,
                         float* lse, const float* cos_t, const float* sin_t, const int32_t* row_src, const int32_t* grp, int T, int M,
                         void* stream);
/* test probes: lane-half exchange semantics the kernel above relies on; its shader-clock phase trace (-DCOATI_AB_TRACE builds) */
int coati_ab_probe_swap(uint32_t* out, void* stream);
int coati_ab_trace_read(unsigned long long* out);
#endif
int coati_attn_bwd_hs(const uint16_t* qkv, const uint16_t* y, const uint16_t* dy, const float* lse, float* dscratch,
                      uint16_t* dqkv, const float* cos_t, const float* sin_t, int B, int T, int n_head, int head_size,
                      void* stream);
int coati_gemm_qkv_rope_hs(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, int M, int C,
                           uint16_t* qkv, int64_t ldc, const float* cos_t, const float* sin_t, int T, int head_size,
                           void* stream);
int coati_attn_decode_hs(const uint16_t* qkv, uint16_t* cache, uint16_t* y, int B, int n_head, int head_size, int Tmax,
                         int pos, void* stream);

/* token embedding gather with [UNK]-slot injection (basic_transformer.py:80-81, smiles_xformer.py:444-448) */
int coati_embed_fwd(const int64_t* idx, const float* table, const float* injection, int unk_token, float* x,
                    int B, int T, int C, int V, void* stream);
int coati_embed_bwd(const int64_t* idx, const float* dx, float* dtable, float* dinjection, int unk_token, int B,
                    int T, int C, int V, void* stream);
/* get_stop_token_embs (smiles_xformer.py:50-68): err[0] |= 1 when a row does not hold exactly one stop token */
int coati_find_stop(const int64_t* idx, int stop_token, int32_t* pos, int32_t* err, int B, int T, void* stream);
int coati_gather_rows(const float* x, const int32_t* pos, float* out, int B, int T, int C, void* stream);
int coati_scatter_rows_add(const float* dout, const int32_t* pos, float* dx, int B, int T, int C, void* stream);
int coati_bad_rows(const int64_t* tokens, uint8_t* bad, int B, int T, void* stream);

/* batch tail of clip_ar_xform on the device (clip_e2e.py:312-329).
   ncols[0] = (tokens.sum(0) > 0).sum();  batch_tail: tokens_out = tokens[:, :ncol] (contiguous), and if y_next_out:
   y_next_out[:, t] = tokens[:, t+1] (0 in the last column) with every id in masked_ids replaced by -1 */
int coati_batch_ncols(const int64_t* tokens, int B, int n_seq, int32_t* ncols, void* stream);
int coati_batch_tail(const int64_t* tokens, int B, int n_seq, int ncol, int64_t* tokens_out, int64_t* y_next_out,
                     const int64_t* masked_ids, int n_masked, void* stream);

/* y = x * sigmoid(x), f32 (point_clip_to_special_tokens = SiLU -> Linear, clip_e2e.py:432-435) */
int coati_silu(const float* x, float* y, int64_t n, void* stream);

/* decode-time operators: one query per (sequence, head) against the KV cache [B, nh, Tmax, k16|v16] (appends position
   pos first), and top-k sampling: token = inds[multinomial(softmax(topk(logits, k) * inv_temp))] with the caller's
   uniforms u[B]; rows flagged in stopped[] emit pad_token, rows drawing stop_token get flagged (smiles_xformer.py:305-324) */
int coati_attn_decode(const uint16_t* qkv, uint16_t* cache, uint16_t* y, int B, int n_head, int Tmax, int pos, void* stream);
int coati_topk_sample(const float* logits, int64_t ldl, int B, int V, int k, float inv_temp, const float* u,
                      int64_t* tokens_out, int32_t* stopped, int stop_token, int pad_token, void* stream);

/* E(3)-GNN pieces (e3gnn_clip.py:108-137, e_gcl_sparse.py) -- see csrc/gnn.hip for the dense-edge formulation */
int coati_gnn_embed(const int64_t* atoms, const int32_t* lut_ix, const int32_t* lut_iy, const float* W,
                    const float* b, float* h32, uint16_t* h16, int64_t ld16, float* rstd, float* mask, int BA,
                    int H, void* stream);
int coati_gnn_geom(const float* coords, const float* mask, float cutoff, float* d2, float* w, int B, int A,
                   void* stream);
/* make_neighborlist (e_gcl_sparse.py:27-77) on the device: the dense weights of coati_gnn_geom (w > 0 <=> edge) -> the
 * compacted, receiver-major edge list the engine's GNN runs on.  seg [B*A + 1]: edges received by node row r are
 * seg[r] .. seg[r+1]; n_edges[0] = E; e_bj / e_bk [E]: receiver / sender node rows (= Is*A+Js / Is*A+Ks of the reference,
 * same order); e_rev [E]: index of the reverse edge; e_d2 / e_w [E]: squared distance / cutoff weight; pos: B*A*A int scratch.
 * All edge arrays must hold B*A*A entries.  No host sync: consumers read n_edges on the device. */
int coati_gnn_compact(const float* w_dense, const float* d2_dense, int32_t* seg, int32_t* n_edges, int32_t* e_bj, int32_t* e_bk,
                      int32_t* e_rev, float* e_d2, float* e_w, int32_t* pos, int B, int A, void* stream);

/* Edge model on the compacted list (e_gcl_sparse.py:169-215, factored: W1 [h_j, h_k, d2] = Pa[j] + Pb[k] + w1c d2):
 * coati_gnn_edge_pre:    e1[e, :] = SiLU(P[bj, 0:H] + P[e_bk[e], H:2H] + e_d2[e] w1c + b1) for the edges e of every receiver row bj
 *                        (seg [BA + 1]); P [BA, 2H] bf16 = (Pa | Pb), e1 [E, H] bf16
 * coati_gnn_edge_reduce: mi[bj, :] = sum over the receiver's segment of SiLU(s2[e, :]) e_w[e]   (the cutoff-weighted message sum,
 *                        e_gcl_sparse.py:204-207, as contiguous segment sums: no scatter_add_, no atomics) */
int coati_gnn_edge_pre(const uint16_t* P, int64_t ldp, const int32_t* seg, const int32_t* e_bk, const float* e_d2, const float* w1c,
                       int64_t w1c_stride, const float* b1, uint16_t* e1, int BA, int H, void* stream);
int coati_gnn_edge_reduce(const uint16_t* s2, const int32_t* seg, const float* e_w, uint16_t* mi, int64_t ldmi, int BA, int H, void* stream);

/* symmetric InfoNCE over local rows x global columns (clip_e2e.py:35-47).  logits [R,N] f32 is overwritten by its
 * gradient scaled by gscale * inv_count[0]; rows whose label (label0 + r) is a bad row are ignored. */
int coati_infonce_rows(float* logits, int64_t ld, int R, int N, int label0, const uint8_t* bad, float* loss_sum,
                       const float* inv_count, float gscale, void* stream);

/* Barlow-Twins head (BASELINE.json configs[3]; no reference implementation exists -> parity unpinned).  The host
 * (coati_amd/barlow.py) strings these together and all-reduces the small statistics / the ExE matrix across ranks:
 *   coati_count_valid: count[0] = #rows with bad == 0, inv[0] = 1/count
 *   coati_colsum2:     out[0:E] = sum_b keep a ; out[E:2E] = sum_b keep a*(b2 ? b2 : a)
 *   coati_standardize: zt = keep (z - mu)/sqrt(var + 1e-5), rsigma written
 *   coati_barlow_dc:   C (raw Za~^T Zb~) -> dL/dC / n in place, loss[0] += L
 *   coati_standardize_bwd: batch-norm style backward of the standardisation, times `scale` */
int coati_count_valid(const uint8_t* bad, int n, float* count, float* inv, void* stream);
int coati_colsum2(const float* a, const float* b2, const uint8_t* bad, float* out, int B, int E, void* stream);
/* zc = keep (z - sum/n): first pass of the two-pass batch statistics (variance from the centred rows) */
int coati_center_rows(const float* z, const uint8_t* bad, const float* sum, const float* count, float* zc, int B, int E,
                      void* stream);
int coati_standardize(const float* z, const uint8_t* bad, const float* stats, const float* count, float* zt, float* rsigma,
                      int B, int E, void* stream);
int coati_barlow_dc(float* C, const float* count, float lam, float* loss, int E, void* stream);
int coati_standardize_bwd(const float* dzt, const float* zt, const uint8_t* bad, const float* rsigma, const float* m,
                          const float* count, float scale, float* dz, int B, int E, void* stream);

/* clip_grad_norm_ + AdamW over flat buffers (train_coati.py:145-151, 276-277) */
int coati_grad_sqnorm(const float* g, int64_t n, float* partial, int n_partial, float* out_norm, float max_norm,
                      float* out_coef, void* stream);
int coati_adamw(float* p, const float* g, float* m, float* v, uint16_t* shadow, int64_t n, float lr, float b1,
                float b2, float eps, float wd, int step, const float* coef, float gscale, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Engine: the whole step behind e3gnn_smiles_clip_e2e.forward_dist + do_minibatch, as a fixed launch
 * sequence owned by the library (no Python between kernels).  Buffers are still owned by the caller.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct coati_config {
  int32_t n_layer_xformer;  /* clip_e2e.py:357-378 kwargs */
  int32_t n_layer_e3gnn;
  int32_t n_hidden_xformer; /* C */
  int32_t n_hidden_e3nn;    /* H */
  int32_t n_embd_common;    /* E (must equal C: smiles_to_clip's LayerNorm is sized by E, clip_e2e.py:423-426) */
  int32_t n_head;           /* C / n_head must be 16 */
  int32_t n_seq;
  int32_t n_tok;            /* V */
  float msg_cutoff;         /* effective cutoff of e_gcl_sparse (always 5.0 in the reference) */
  int32_t pad_token, stop_token, unk_token;
  int32_t use_fp8;          /* 1: the four Linear layers of every transformer block run their forward and input-gradient
                               products on MXFP8 (coati_gemm_mx8); needs C % 128 == 0 and coati_engine_bind_fp8 */
  /* constructor flags of e3gnn_smiles_clip_e2e (clip_e2e.py:370-376, 405-435); grande_closed = 1 / 1 / 1, the reference's own
     do_args() defaults (train_coati.py:520-523) = 0 / 0 / 1 */
  int32_t norm_clips;        /* 1: point_to_clip / smiles_to_clip = LayerNorm -> Linear (state_dict .0 / .1); 0: plain Linear */
  int32_t token_mlp;         /* 1: point_clip_to_special_tokens = SiLU -> Linear; 0: Identity (no parameters) */
  int32_t use_point_encoder; /* 0: encode_points returns zeros (clip_e2e.py:454-463); the point encoder's and point_to_clip's
                                parameters exist in the state_dict but never receive a gradient (skipped by clip-norm / AdamW) */
  int32_t biases;            /* ABI v4.  0: the transformer blocks' four Linear layers have no bias (basic_transformer.py:113-115, 166-168 with
                                config.biases = False): the parameter table holds no such entries; LayerNorm biases and lm_head are unaffected */
  int32_t norm_embed;        /* ABI v4.  1: a LayerNorm follows the token embedding (basic_transformer.py:72-76: tok_emb = Sequential(Embedding,
                                LayerNorm), entries xformer.emb.tok_emb.0.weight / .1.weight / .1.bias); the injection overwrites its output.  The
                                reference also registers an unused xformer.norm_embed LayerNorm (smiles_xformer.py:81-82): in the table, never trained */
  int32_t torch_emb;         /* ABI v5.  1: the point encoder's node features are rows of an nn.Embedding(84, H) (entry point_encoder.emb.weight) instead
                                of Linear(one-hot period / group) (e3gnn_clip.py:49-56, 74-77, 113-115); atomic numbers above 83 read row 83 (the
                                reference asserts / raises there) */
  int32_t old_architecture;  /* ABI v5.  1 (with norm_clips): point_to_clip / smiles_to_clip = Linear -> LayerNorm (state_dict .0 = Linear, .1 =
                                LayerNorm; clip_e2e.py:409-417) instead of LayerNorm -> Linear.  Needs H == E (the reference sizes the point head's
                                LayerNorm by hidden_nf and applies it to the E-wide output) */
  int32_t residual;          /* ABI v5.  1: every node MLP of the point encoder also sees the one-hot node features h0 (e3gnn_clip.py:97-100,
                                e_gcl_sparse.py:141, 282-290): node_mlp.0.weight is [H, 2H + 28].  Not together with torch_emb */
} coati_config;

typedef struct coati_engine coati_engine;

int coati_engine_create(const coati_config* cfg, coati_engine** out);
void coati_engine_destroy(coati_engine* e);

/* parameter table: entry i = (state_dict name, element offset into the flat f32 buffer, rows, cols) */
int64_t coati_engine_param_elems(const coati_engine* e);
/* leading elements of the flat buffers that receive gradients: the parameters behind it (coord_mlp; the point encoder and
 * point_to_clip when use_point_encoder = 0) are never updated -- torch keeps no optimizer state for them (p.grad is None) */
int64_t coati_engine_trainable_elems(const coati_engine* e);
int coati_engine_n_entries(const coati_engine* e);
int coati_engine_entry(const coati_engine* e, int i, char* name, int name_cap, int64_t* offset, int32_t* rows,
                       int32_t* cols);
int64_t coati_engine_shadow_elems(const coati_engine* e);
int64_t coati_engine_workspace_bytes(const coati_engine* e, int B, int T1, int T2, int A, int Bg);
/* Grow-only buffer capacities: every later workspace_bytes / forward / encode sizes its buffers for at least (B, T1, T2, A), so that
 * batches whose T / A change from step to step (clip_ar_xform truncates each batch to its longest row, clip_e2e.py:312-315) keep the
 * same buffer addresses inside the caller's workspace and the engine's cached launch tables stay valid.  Call BEFORE
 * coati_engine_workspace_bytes; values below the current capacity are ignored. */
int coati_engine_reserve(coati_engine* e, int B, int T1, int T2, int A);

/* bind caller-owned buffers: params/grads/adam m/adam v (f32, param_elems), shadow (bf16, shadow_elems, zeroed),
 * RoPE tables [n_seq,16] f32, periodic-table LUTs [120] int32 (one-hot indices, -1 = none). */
int coati_engine_bind(coati_engine* e, float* params, float* grads, float* adam_m, float* adam_v, uint16_t* shadow,
                      const float* rope_cos, const float* rope_sin, const int32_t* lut_ix, const int32_t* lut_iy);
/* fp8 mode (cfg.use_fp8): caller-owned buffer for the MXFP8 copies of the transformer weights (e4m3 + E8M0 scales, natural
 * and transposed), coati_engine_fp8_bytes bytes; refreshed together with the bf16 shadows */
int64_t coati_engine_fp8_bytes(const coati_engine* e);
int coati_engine_bind_fp8(coati_engine* e, uint8_t* fp8_shadow, int64_t bytes);
/* rebuild every bf16 shadow (natural + transposed/packed) from the f32 parameters */
int coati_engine_refresh_shadows(coati_engine* e, void* stream);

/* forward_dist + AR loss.  raw_tokens [B,T1], tokens [B,T2], y_next [B,T2], atoms [B,A] int64; coords [B,A,3] f32;
 * use_point [B] uint8 (1 -> inject the point-cloud token; replaces `rand(B) > p_clip_emb_smi`, clip_e2e.py:802).
 * Outputs: h_e3gnn, h_smiles [B,E] f32; bad_rows [B] uint8; scal[0] = sum AR loss, scal[1] = #targets,
 * scal[6] = error flags (bit 0: a raw_tokens row without exactly one [STOP]).  Zeroes the gradient buffer.
 *
 * rows1 / rows2 > 0: PACKED ROWS.  clip_ar_xform pads every row to the batch's longest (clip_e2e.py:288-330) and the reference
 * computes the padding; under causal attention the positions behind a row's last token influence nothing that reaches a
 * loss (their targets are -1, train_coati.py:260-265; the encoder reads the [STOP] position only, smiles_xformer.py:50-68),
 * so both transformer passes then run on the concatenation of the rows' real prefixes: rows1 = sum over rows of
 * (1 + last non-[PAD] position) of raw_tokens, rows2 the same for tokens (a position also counts when its y_next is not
 * -1).  The caller computes the counts on the host (the batch assembler has the tokens there: no device -> host sync); the
 * device recomputes them and sets scal[6] bit 1 on a mismatch.  Losses, gradients, h_e3gnn / h_smiles are those of the padded
 * run; coati_engine_logits is not available after a packed forward.  rows1 = rows2 = 0: the padded layout. */
int coati_engine_forward(coati_engine* e, void* workspace, int64_t workspace_bytes, int B, int T1, int T2, int A,
                         const int64_t* raw_tokens, const int64_t* tokens, const int64_t* y_next,
                         const int64_t* atoms, const float* coords, const uint8_t* use_point, float* h_e3gnn,
                         float* h_smiles, uint8_t* bad_rows, float* scal, int train, int64_t rows1, int64_t rows2, void* stream);
/* The same forward in two calls: coati_engine_forward(..., train | 2, ...) returns behind the heads -- point encoder, encoder pass,
 * point_to_clip / smiles_to_clip / special token; h_e3gnn, h_smiles and bad_rows are final -- and coati_engine_forward_decoder
 * enqueues the rest (decoder pass with the injected token, lm_head + AR cross-entropy).  Between the two the caller may put the
 * contrastive head (the embedding all-gather, coati_engine_infonce, the reduce-scatter) on ANOTHER stream: it only needs the
 * embeddings, so it runs underneath the decoder pass instead of in front of the backward (clip_e2e.py:772-814 computes the
 * embeddings first as well; train_coati.py:256-258 gathers them after the whole forward). */
int coati_engine_forward_decoder(coati_engine* e, void* stream);
/* Inference encoders alone: e3gnn_smiles_clip_e2e.encode_tokens (clip_e2e.py:448-452) when raw_tokens + h_smiles are
 * given, .encode_points (clip_e2e.py:454-463) when atoms + coords + h_e3gnn are given (either pair may be null).  Only
 * the requested tower runs; workspace as for coati_engine_forward with T2 = 1.  scal[6] bit 0: a row without exactly
 * one [STOP] (smiles_xformer.py:63-66). */
int coati_engine_encode(coati_engine* e, void* workspace, int64_t workspace_bytes, int B, int T1, int A,
                        const int64_t* raw_tokens, const int64_t* atoms, const float* coords, float* h_smiles,
                        float* h_e3gnn, float* scal, void* stream);

/* logits [B*T2, ldl] f32 of the last forward (API parity with forward_dist's third return value) */
int coati_engine_logits(coati_engine* e, float* logits, int64_t ldl, void* stream);

/* InfoNCE over local rows: S_loc/C_loc [B,E], S_all/C_all [Bg,E], bad_all [Bg]; rank rows start at row0.
 * Writes dS_all, dC_all [Bg,E] (to be reduce-scattered when Bg > B), adds the two directional loss sums to
 * scal[2], scal[3], writes the valid-row count to scal[4].  gscale multiplies the gradients. */
int coati_engine_infonce(coati_engine* e, const float* S_loc, const float* C_loc, const float* S_all,
                         const float* C_all, const uint8_t* bad_all, int B, int Bg, int row0, float gscale,
                         float* dS_all, float* dC_all, float* scal, void* stream);

/* backward of the last forward.  dh_smiles / dh_e3gnn [B,E]: gradient of the contrastive term w.r.t. the two
 * embeddings.  stage: 0 = everything; 1 = lm_head + decoder pass + heads; 2 = encoder pass; 3 = point encoder
 * (1,2,3 in that order == 0; lets the caller overlap bucketed gradient all-reduces with the remaining stages);
 * 4, 5 = stage 2 in two halves (layers [L/2, L) incl. ln_f, then [0, L/2) incl. the embeddings): 1,4,5,3 == 0. */
int coati_engine_backward(coati_engine* e, const float* dh_smiles, const float* dh_e3gnn, int stage, void* stream);

/* clip_grad_norm_(max_norm) + AdamW + shadow refresh.  scal[5] receives the pre-clip gradient norm. */
int coati_engine_optimizer_step(coati_engine* e, float lr, float beta1, float beta2, float eps, float weight_decay,
                                float max_norm, int step, float* scal, void* stream);
/* Data-parallel hosts: replace this rank's error word of the step (bit 0: a row without [STOP], smiles_xformer.py:63-66; bit 1: packed
 * row counts differ from the tokens) by the one reduced over all ranks BEFORE coati_engine_optimizer_step, whose AdamW kernel drops the
 * update when the word is non-zero: every rank then skips or none does.  word_dev: one int32 in device memory. */
int coati_engine_set_error_word(coati_engine* e, const int32_t* word_dev, void* stream);

/* per-site kernel timing with HIP events on the launch stream (bench.py roofline leg) */
int coati_engine_prof_select(coati_engine* e, int site);              /* -1 disables */
int coati_engine_prof_add_site(coati_engine* e, int site);           /* after prof_select: time this site's launches as well */
/* keep != 0: time the selected site while the step runs as the product runs it (point encoder concurrent on the side stream);
 * 0 (default after every prof_select): the point encoder is serialised onto the launch stream, a site's events bracket its kernels alone */
int coati_engine_prof_keep_overlap(coati_engine* e, int keep);
/* paused != 0: no events until resumed; the selection, its counters and the overlap setting stay (sampling a subset of the steps) */
int coati_engine_prof_pause(coati_engine* e, int paused);
int coati_engine_prof_collect(coati_engine* e, double* total_ms, int64_t* launches, double* flops_per_launch);
/* algorithmic HBM bytes per launch (operands read once, results written once) of the site collected last */
int coati_engine_prof_last_bytes(coati_engine* e, double* bytes_per_launch);
int coati_engine_site_count(void);

/* ---- inference decode: KV-cached generation, one position per call -------------------------------
   replaces RotarySmilesTransformer.generate_top_k_with_inj_batch's per-token full-prefix recompute
   (smiles_xformer.py:272-351).  workspace: caller-owned device memory (KV cache [L,B,nh,Tmax,32] bf16 + per-step
   activations).  decode_step consumes tokens[B]; rows equal to the [UNK] id read injection[B, C] instead of the
   embedding table (smiles_xformer.py:444-448); logits [B, n_tok] f32 (optional). */
int64_t coati_engine_decode_workspace_bytes(coati_engine* e, int B, int Tmax);
int coati_engine_decode_begin(coati_engine* e, void* workspace, int64_t ws_bytes, int B, int Tmax);
int coati_engine_decode_step(coati_engine* e, const int64_t* tokens, const float* injection, float* logits, int64_t ldl,
                             void* stream);
int coati_engine_decode_pos(coati_engine* e);
/* the same step captured into HIP graphs (one hipGraphLaunch instead of ~115 kernel launches; pays off at small batch).
   graph_build: after decode_begin, on an explicit stream.  graph_step: tokens[B] / injection[B, C] (device) are copied
   into the session's fixed input buffers; *logits_out points at the session's logits [B, n_tok] (row stride *ldl_out),
   valid until the next step.  Eager and graph steps may be mixed. */
int coati_engine_decode_graph_build(coati_engine* e, void* stream);
int coati_engine_decode_graph_step(coati_engine* e, const int64_t* tokens, const float* injection, float** logits_out,
                                   int64_t* ldl_out, void* stream);
const char* coati_engine_site_name(int site);

/* ---- host-side trie tokenizer (reference tokenizers/trie.py:39-214, trie_tokenizer.py:48-109) ---------
   special[i] has id special_ids[i] (null: i), smiles[j] has id smiles_ids[j] (null: n_special + j).
   Leftmost-longest matching, special tokens first. */
typedef struct coati_tokenizer coati_tokenizer;
int coati_tokenizer_create(const char* const* special, const int32_t* special_ids, int n_special, const char* const* smiles,
                           const int32_t* smiles_ids, int n_smiles, coati_tokenizer** out);
void coati_tokenizer_destroy(coati_tokenizer* tk);
/* token count (may exceed cap; nothing past cap is written) or -(1 + byte offset) of the first piece with no id */
long long coati_tokenizer_encode(const coati_tokenizer* tk, const char* text, long long n_bytes, int32_t* ids_out, int cap);
/* pre_tokenize: byte ranges [begin, end) and ids (-1 = not in the vocabulary) of the pieces; returns their number */
long long coati_tokenizer_pieces(const coati_tokenizer* tk, const char* text, long long n_bytes, int64_t* begin, int64_t* end,
                                 int32_t* id, int cap);
/* rows -> out[n_rows, n_seq] int64 zero-padded, len[i] = tokens / -1 unknown piece / -2 longer than n_seq; threaded */
int coati_tokenizer_encode_batch(const coati_tokenizer* tk, const char* const* rows, int n_rows, int n_seq, int64_t* out,
                                 int32_t* len, int n_threads);

/* ---- data-parallel exchange for hosts without torch.distributed -----------------------------------------
   Stream-ordered calls into RCCL (resolved with dlopen at the first call: no link-time dependency; $COATI_RCCL_LIB overrides the
   library name).  One communicator per process = per GPU, bound to the HIP device that is current at coati_comm_init.  They are
   the three collectives of the step: the embedding all-gather (reference AllGatherFunction.forward,
   coati/models/autograd_funs/autograd_funs.py:10-14), the reduce-scatter of the embedding gradients (.backward, :16-21) and the
   gradient-bucket all-reduce (DistributedDataParallel, coati/training/train_coati.py:71-76).  dtype: 0 = f32, 1 = bf16.
   The Python host of this repository uses torch.distributed (backend "nccl" = the same library) instead. */
#define COATI_COMM_ID_BYTES 128
typedef struct coati_comm coati_comm;
/* rank 0 draws the id and hands it to the other ranks by the host's own means (file, socket, MPI, ...) */
int coati_comm_unique_id(void* id_out, int id_bytes);
int coati_comm_init(const void* unique_id, int rank, int world, coati_comm** out);
int coati_comm_rank(const coati_comm* c);
int coati_comm_world(const coati_comm* c);
int coati_comm_destroy(coati_comm* c);
/* recv[world * rows, cols] <- the ranks' send[rows, cols], rank-major */
int coati_allgather_rows(coati_comm* c, const void* send, void* recv, int64_t rows, int64_t cols, int dtype, void* stream);
/* recv[rows, cols] <- this rank's row block of the SUM over ranks of send[world * rows, cols] */
int coati_reducescatter_rows(coati_comm* c, const void* send, void* recv, int64_t rows, int64_t cols, int dtype, void* stream);
/* buf[n] <- sum (average != 0: mean) over ranks, in place */
int coati_allreduce_bucket(coati_comm* c, void* buf, int64_t n, int dtype, int average, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COATI_HIP_H */
