// Internal launcher interface shared by the .hip translation units of libcoati_hip.so.
// Everything here takes raw device pointers + a hipStream_t; nothing allocates device memory.
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// GEMM family (gemm.hip)
// ------------------------------------------------------------------------------------------------
enum CoatiEpi {
  EPI_BF16 = 0,        // C(bf16) = acc + bias
  EPI_F32 = 1,         // C(f32)  = acc + bias
  EPI_RES_F32 = 2,     // C(f32)  = aux_in(f32) + acc + bias
  EPI_GELU = 3,        // aux_out(bf16) = acc + bias ; C(bf16) = NewGELU(acc + bias)
  EPI_DGELU = 4,       // C(bf16) = acc * NewGELU'(aux_in(bf16))
  EPI_SILU = 5,        // aux_out(bf16) = acc + bias ; C(bf16) = SiLU(acc + bias)
  EPI_DSILU = 6,       // C(bf16) = acc * SiLU'(aux_in(bf16))
  EPI_ACC_F32 = 7,     // C(f32) += acc
  EPI_CE_PARTIAL = 8,  // per (row, 128-col tile): (max, sum exp(v-max)) -> partial[row][tile]
  EPI_CE_BWD = 9,      // C(bf16) = (exp(acc - lse[row]) - [col==target[row]]) / count   (0 if target<0)
  EPI_EDGE_DPRE = 10,  // GNN: C(bf16) = acc * SiLU'(Pa[bj] + Pb[bk] + d2*w1c + b1)
  EPI_QKV_ROPE = 11,   // C(bf16) = RoPE(acc + bias) on the q and k column blocks (cols < 2*rope_C), plain on v
  EPI_GELU_GRAD = 12,  // aux_out(u8 fixed point, common.h packq8) = NewGELU'(acc + bias) ; C(bf16) = NewGELU(acc + bias): the backward is EPI_MUL_AUX
  EPI_MUL_AUX = 13,    // C(bf16) = acc * dequant(aux_in(u8 fixed point))
  EPI_LNBWD = 14,      // N = C, ring GEMM only: dy = acc is the gradient w.r.t. a LayerNorm's output; C(f32) = aux_in(f32) + LayerNorm-backward(dy | lnb_x, lnb_mean,
                       // lnb_rstd, lnb_gamma) (aux_in may be C: the residual-stream gradient in place), aux_out(bf16, optional) = its bf16 copy,
                       // lnb_partial[workgroup][2N] = the workgroup's dgamma | dbeta sums (added up by launch_ln_finish_batched)
  EPI_COUNT = 15
};

struct GemmArgs {
  const int* m_dev;     // optional: the number of rows that exist, read on the DEVICE (<= M; M then only sizes the grid):
                        // row counts that are data dependent (the E(3)-GNN's compacted edge list) need no host sync
  const void* A;        // [M,K] row-major, bf16 or f32 (a_f32)
  long long lda;
  const bf16_t* B;      // [N,K] row-major bf16 (NT: C = A * B^T)
  long long ldb;
  int M, N, K;
  void* C;
  long long ldc;
  int n_store;          // columns [N, n_store) of C are written as zero (0 -> = N)
  const float* bias;    // [N] or null
  const void* aux_in;
  void* aux_out;
  long long ld_aux;
  // cross-entropy epilogues
  const float* lse;         // [M]
  const long long* target;  // [M], -1 = ignore
  const float* scal;        // scal[1] = number of non-ignored targets
  float2* partial;          // [M, tiles_n]
  int partial_tile;         // EPI_CE_PARTIAL: 0 / 128 = one entry per 128 columns (tiled kernel); 64 = the buffer holds one entry
                            // per 64 columns, so the row-block kernel may be used (gemm_ce_tile_width tells which one ran)
  // GNN edge epilogue
  const bf16_t* P;          // [B*A, 2H] (Pa | Pb)
  long long ldp;
  const float* d2;          // [B*A*A]
  const float* w1c;         // w1c[c] = w1c[c * w1c_stride]
  long long w1c_stride;
  const float* b1;
  int natom;
  int H;
  const int* e_bj;          // compacted edge list: row -> receiver node row / sender node row (null: dense grid, row = (b*A + j)*A + k)
  const int* e_bk;
  // rotary epilogue: row m is token t = m % rope_T; tables [n_seq, rope_hs] f32; head size rope_hs = 16 or 32 (0 -> 16)
  const float* rope_cos;
  const float* rope_sin;
  int rope_T;
  int rope_C;
  int rope_hs;
  const int* rope_pos;   // decode: when set, EVERY row sits at token position *rope_pos (device memory; graph replay)
  const int* rope_row_t; // packed rows: when set, row m sits at token position rope_row_t[m] (instead of m % rope_T)
  // fp8 mode (gemm_mx8.hip only): also emit the MXFP8 copy of the bf16 output (e4m3 [M, ld_q8] + one scale per 32 columns,
  // [M, N / 32]) for the next MXFP8 product; N % 32 == 0, bf16-output epilogues
  unsigned char* q8_out;
  unsigned char* q8_scales;
  long long ld_q8;
  // LayerNorm fused into the A load (row-block kernel, K = 256; gemm_rb256_ln_fusable): the operand is LN(ln_x) and is
  // computed while the A slab is loaded; A / lda then name the bf16 buffer that RECEIVES the normalised rows (the weight
  // gradient reads it later), ln_mean / ln_rstd the per-row statistics for the backward
  const float* ln_x;        // [M, K] f32, null = plain bf16 A
  long long ln_ldx;
  const float* ln_gamma;    // [K]
  const float* ln_beta;     // [K]
  float* ln_mean;           // [M]
  float* ln_rstd;           // [M]
  // EPI_LNBWD: the LayerNorm whose OUTPUT gradient this product computes
  const float* lnb_x;       // [M, N] f32: the LayerNorm's input rows
  long long lnb_ldx;
  const float* lnb_mean;    // [M]
  const float* lnb_rstd;    // [M]
  const float* lnb_gamma;   // [N]
  float* lnb_partial;       // [workgroups][2 N]
  // EPI_LNBWD, optional: a SECOND product chained behind the LayerNorm backward in the same launch -- chain_C (bf16 [M, 256]) =
  // dx16 chain_W^T with chain_W [256, 256] bf16 (the c_proj input gradient that consumes ln_2's backward output: the rows are
  // re-read from L2 by the workgroup that has just written them); needs aux_out
  const bf16_t* chain_W;
  long long chain_ldw;
  bf16_t* chain_C;
  long long chain_ldc;
};

int launch_gemm_nt(const GemmArgs& a, int a_f32, int epi, hipStream_t s);
// row-block kernel for K = 256 (gemm_rb.hip); launch_gemm_nt dispatches to it when supported
bool gemm_rb256_supported(const GemmArgs& a, int a_f32, int epi);
// EPI_LNBWD: true when the ring GEMM takes this shape with the LayerNorm backward in its write-out (gemm_ring.hip); *nwg = the
// number of partial rows the launch will leave in lnb_partial
bool gemm_ring_lnbwd_supported(const GemmArgs& a, int* nwg);
// true when launch_gemm_nt(a, 0, epi) accepts the ln_* fields (LayerNorm fused into the operand load)
bool gemm_rb256_ln_fusable(const GemmArgs& a, int epi);
// column width of the EPI_CE_PARTIAL entries launch_gemm_nt writes for these arguments (64 or 128)
int gemm_ce_tile_width(const GemmArgs& a);
int launch_gemm_rb256(const GemmArgs& a, int epi, hipStream_t s);
// the same on 16-row slabs / 13-16 waves per workgroup (gemm_rb16.hip): packed batch sizes (36 865 .. 65 536 rows)
bool gemm_rb16_supported(const GemmArgs& a, int a_f32, int epi);
// weight-resident persistent form for N <= 256 with a device-side row count (the GNN's edge products)
bool gemm_rb16_resident_supported(const GemmArgs& a, int a_f32, int epi);
int launch_gemm_rb16_resident(const GemmArgs& a, int epi, hipStream_t s);
int launch_gemm_rb16(const GemmArgs& a, int epi, hipStream_t s);
#ifdef COATI_EXPERIMENTAL   // csrc/experimental/: parity-green, slower than what they would replace; not in the default library (build.py)
// the same products on 32-row slabs in the TRANSPOSED form (gemm_t32.hip, round 5): 24 577 .. 65 536 rows; taken before the 16-row slabs
// the MLP half of a block as one launch (mlp64.hip): out = x + c_proj(NewGELU(c_fc(ln_2(x)))), d = 256, hidden 1024
struct Mlp64Args {
  const float* x;          // [M, ldx] f32: the residual stream in front of ln_2
  long long ldx;
  const float* gamma;      // ln_2
  const float* beta;
  bf16_t* a2;              // [M, 256] bf16: ln_2(x) (the c_fc weight gradient's operand)
  float* mean;             // [M]
  float* rstd;             // [M]
  const bf16_t* W1;        // c_fc.weight [1024, 256]
  const float* b1;         // [1024]
  const bf16_t* W2;        // c_proj.weight [256, 1024]
  const float* b2;         // [256]
  bf16_t* g;               // [M, 1024] bf16: NewGELU(c_fc(.)) (the c_proj weight gradient's operand)
  unsigned char* codes;    // [M, 1024]: NewGELU' as 8-bit fixed point (common.h)
  float* out;              // [M, ldo] f32
  long long ldo;
  int M;
};
bool mlp64_fwd_supported(int M, int C, int hidden);
int launch_mlp64_fwd(const Mlp64Args& a, hipStream_t s);
bool gemm_t32_supported(const GemmArgs& a, int a_f32, int epi);
int launch_gemm_t32(const GemmArgs& a, int epi, hipStream_t s);
#endif
// ring kernel for N = 256, long K (gemm_ring.hip)
bool gemm_ring256_supported(const GemmArgs& a, int a_f32, int epi);
int launch_gemm_ring256(const GemmArgs& a, int epi, hipStream_t s);
// MXFP8 (gemm_mx8.hip): e4m3 operands + one E8M0 scale per 32 k, v_mfma_scale_f32_32x32x64_f8f6f4.  a.A / a.B are the e4m3
// bytes (a.lda / a.ldb in bytes), sa / sb the scale bytes [rows, K / 32]; K % 128 == 0; epilogues BF16, F32, RES_F32, GELU_GRAD,
// MUL_AUX, QKV_ROPE.  launch_quant_mx8 quantises rows of bf16 / f32 (blocks of 32 along k, shared exponent floor(log2 amax) - 8).
int launch_gemm_mx8(const GemmArgs& a, const unsigned char* sa, const unsigned char* sb, int epi, hipStream_t s);
int launch_quant_mx8(const void* x, int x_f32, long long ldx, unsigned char* q, long long ldq, unsigned char* scales, int M, int K, hipStream_t s);

// dW[N,K] (f32, atomic +=) = A[M,N]^T * B[M,K];  dbias[N] (atomic +=) = colsum(A) if non-null
struct WgradArgs {
  const void* A;   // [M,N] bf16 or f32
  long long lda;
  const bf16_t* B; // [M,K] bf16
  long long ldb;
  int M, N, K;
  float* dW;
  long long ldw;
  float* dbias;
  int n_out;   // rows of dW that exist (0 -> N); columns of A beyond it must be zero
  const int* m_dev = nullptr;   // optional: rows that exist, read on the device (<= M)
};
int launch_wgrad(const WgradArgs& a, int a_f32, hipStream_t s);
// Grouped, atomics-free form (gemm.hip, wgrad_dma_table_kernel): one workgroup per 128 x 128 output tile of a list of
// problems (bf16 A, bias), each streaming all of M.  The table lives in device memory (one entry per workgroup).
struct WgradTile {
  WgradArgs p;
  int tiles_k;   // 128-column tiles along K of this problem
  int tile;      // tile index inside the problem (tile_n * tiles_k + tile_k)
  int c_begin = 0, c_end = 0;   // split tables only: the 64-row chunks [c_begin, c_end) of M this entry streams
};
int launch_wgrad_table(const WgradTile* dev_table, int n_tiles, hipStream_t s, int tile_size = 128, int M_rt = 0);   // M_rt > 0: rows of this launch (<= the M of the table entries)   // every entry of a table has the same tile size
bool wgrad_table_tile256_ok(const WgradArgs& a);
// Grouped form for SMALL row counts (the E(3)-GNN's node-level Linears: 16 384 rows): the entries of a problem's 128 x 128 tiles
// are repeated over n_splits slices of M, every entry adds its partial tile with fp32 atomics (dbias may be null)
int launch_wgrad_split_table(const WgradTile* dev_table, int n_entries, hipStream_t s);
#ifdef __cplusplus
#include <vector>
int wgrad_table_append(std::vector<WgradTile>& tab, const WgradArgs& a, int tile_size = 128);   // host: appends the tiles of one problem
int wgrad_table_append_split(std::vector<WgradTile>& tab, const WgradArgs& a, int n_splits);    // host: tiles x M slices of one problem
#endif

// C[M,N] (f32) = sum_k A(m,k) B(k,n) [+ bias[n]] [+ C];  A(m,k) = A[m*ars + k*acs], B(k,n) = B[k*brs + n*bcs]
// exact-f32 MFMA (v_mfma_f32_32x32x2_f32).  alpha scales the product.
// several independent small f32 problems in one launch (gemm.hip): add problems, then launch
#define SGEMM_BATCH_MAX 6
struct SgemmProb {
  const float *A, *B, *bias;
  float* C;
  long long ars, acs, brs, bcs, ldc;
  int M, N, K, accumulate, ksplit, tx, ty, tz, wg0;
  float alpha;
};
struct SgemmBatch {
  int n = 0;
  SgemmProb p[SGEMM_BATCH_MAX];
};
int sgemm_batch_add(SgemmBatch& b, const float* A, long long ars, long long acs, const float* B, long long brs, long long bcs,
                    float* C, long long ldc, int M, int N, int K, const float* bias, float alpha, int accumulate);
int launch_sgemm_batch(const SgemmBatch& b, hipStream_t s);
int launch_sgemm(const float* A, long long ars, long long acs, const float* B, long long brs, long long bcs,
                 float* C, long long ldc, int M, int N, int K, const float* bias, float alpha, int accumulate,
                 hipStream_t s);

// ------------------------------------------------------------------------------------------------
// normalisation (norm.hip)
// ------------------------------------------------------------------------------------------------
// y = (x - mean) * rstd [* gamma + beta];  x f32 [M,C];  y16 (bf16, ld16) and/or y32 (f32, ld32) optional.
int launch_layernorm_fwd(const float* x, long long ldx, const float* gamma, const float* beta, bf16_t* y16,
                         long long ld16, float* y32, long long ld32, float* mean, float* rstd, int M, int C,
                         hipStream_t s);
// dx = [dres +] rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma.  dy bf16 or f32.
// xhat is recomputed from (x, mean, rstd) or, if x_is_xhat, x already holds xhat (affine-free norms).
int launch_layernorm_bwd(const void* dy, int dy_f32, long long lddy, const float* x, long long ldx, int x_is_xhat,
                         const float* mean, const float* rstd, const float* gamma, const float* dres,
                         float* dx, bf16_t* dx16, float* dgamma, float* dbeta, float* partial, int M, int C, hipStream_t s);
// partial: optional [2048, 2C] f32 scratch -> deterministic two-stage dgamma/dbeta reduction (else fp32 atomics)
// The same without the second stage: the per-workgroup [gamma | beta] partial sums stay in `partial` (COATI_LN_PARTIAL_ROWS x
// 2C floats at most; *nblk_out rows are valid) until launch_ln_finish_batched adds them up -- one launch for all the
// LayerNorms of a transformer pass instead of one 5-us launch each.
int launch_layernorm_bwd_deferred(const void* dy, int dy_f32, long long lddy, const float* x, long long ldx, int x_is_xhat,
                                  const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                                  bf16_t* dx16, float* partial, int* nblk_out, int M, int C, hipStream_t s);
#define COATI_LN_MAX_SLOTS 72
struct LnFinishBatch {
  int n;                                   // slots in use
  long long dg_off[COATI_LN_MAX_SLOTS];    // offsets of dgamma / dbeta of slot i from the gradient base pointer
  long long db_off[COATI_LN_MAX_SLOTS];
  int nblk[COATI_LN_MAX_SLOTS];            // valid partial rows of slot i (0 = the launch's common count): the stand-alone kernel
                                           // leaves one row per workgroup of ITS grid, the GEMM-fused backward one per ring workgroup
};
int launch_ln_finish_batched(const float* partial, long long slot_stride, int nblk, float* grad_base, const LnFinishBatch& b,
                             int C, hipStream_t s);
#define COATI_LN_PARTIAL_ROWS 2048

// ------------------------------------------------------------------------------------------------
// attention, head size 16 (attention.hip)
// ------------------------------------------------------------------------------------------------
// seq_off [B + 1] (optional): packed rows -- sequence b owns rows seq_off[b] .. seq_off[b + 1] of qkv / y / dy / dqkv and has
// that many tokens (<= T); lse / dscratch keep the padded [B, nh, T] layout
// seq_ord (packed rows, head size 16): the launch order of the sequences, longest first (launch_seq_pack)
int launch_attn_fwd(const bf16_t* qkv, bf16_t* y, float* lse, int B, int T, int n_head, int head_size, hipStream_t s,
                    const int* seq_off = nullptr, const int* seq_ord = nullptr);
int launch_attn_bwd(const bf16_t* qkv, const bf16_t* y, const bf16_t* dy, const float* lse, float* dscratch, bf16_t* dqkv,
                    const float* cos, const float* sin, int B, int T, int n_head, int head_size, hipStream_t s,
                    const int* seq_off = nullptr, const int* seq_ord = nullptr);
// head size 16, T <= 128 on 16-row causal granularity (attention16.hip); launch_attn_fwd / _bwd route there
int launch_attn16_fwd(const bf16_t* qkv, bf16_t* y, float* lse, int B, int T, int n_head, hipStream_t s, const int* seq_off, const int* seq_ord = nullptr);
int launch_attn16_bwd(const bf16_t* qkv, const bf16_t* y, const bf16_t* dy, const float* lse, bf16_t* dqkv, const float* cos_t,
                      const float* sin_t, int B, int T, int n_head, hipStream_t s, const int* seq_off, const int* seq_ord = nullptr);

#ifdef COATI_EXPERIMENTAL
// The attention half of a block as one sequence-stationary kernel (attn_block.hip): xmid = x + c_proj(attention(RoPE(c_attn(ln_1(x))))),
// d = 256, 16 heads of 16, sequences of <= 128 rows.  Saves what the backward reads: a1 = ln_1(x), mean / rstd, qkv (q, k rotated), y, lse.
struct AttnBlockArgs {
  const float* x;          // [M, 256] residual stream in
  float* xmid;             // [M, 256] out
  const float* ln_g;       // ln_1 weight / bias [256]
  const float* ln_b;
  float* mean;             // [M] ln_1 statistics
  float* rstd;
  bf16_t* a1;              // [M, 256] ln_1(x)
  const bf16_t* Wqkv;      // [768, 256] bf16
  const float* bqkv;       // [768]
  const bf16_t* Wproj;     // [256, 256] bf16
  const float* bproj;      // [256]
  bf16_t* qkv;             // [M, 768] out: q, k rotated
  bf16_t* y;               // [M, 256] out
  float* lse;              // [B, 16, Tl]
  const float* cos_t;      // [n_seq, 16]
  const float* sin_t;
  const int* row_src;      // [M] slot b * Tl + t of row m (null: padded layout, the row index itself)
  const int* grp;          // launch_attn_groups: grp[0] = number of groups, grp[1 + g] = first row of group g (g = 0 .. groups)
  int Tl, M;
};
bool attn_block_fwd_supported(int B, int T, int C, int n_head);
// grp [B + 2] ints: groups of whole consecutive sequences with <= 128 rows each (seq_off [B + 1] or null = the padded layout b * T)
int launch_attn_groups(const int* seq_off, int B, int T, int* grp, hipStream_t s);
int launch_attn_block_fwd(const AttnBlockArgs& a, hipStream_t s);
int launch_ab_probe_swap(unsigned* out, hipStream_t s);
#endif

// ------------------------------------------------------------------------------------------------
// embedding / token kernels (embed.hip)
// ------------------------------------------------------------------------------------------------
// row_src / off: packed rows (launch_seq_pack below); null = the padded [B, T] layout
// mode 1 / 2: only the [UNK] rows are written -- the injected vector / zeros (norm_embed models, see embed.hip)
int launch_embed_fwd(const long long* idx, const float* table, const float* injection, int unk_token, float* x,
                     int B, int T, int C, int V, hipStream_t s, const int* row_src = nullptr, int rows = 0, int mode = 0);
// inj_only: only the [UNK] rows' gradient is collected (into dinjection)
int launch_embed_bwd(const long long* idx, const float* dx, float* dtable, float* dinjection, int unk_token,
                     int B, int T, int C, int V, hipStream_t s, const int* off = nullptr, int inj_only = 0);
// Packed rows: the transformer passes run on the concatenation of every row's real prefix (embed.hip).  off [B + 1],
// row_src / row_t [rows_expect] ints, ypk [rows_expect] (optional: the packed targets); err |= 2 when the device-side total
// differs from rows_expect (the count the caller computed on the host)
int launch_seq_pack(const long long* tok, const long long* y, int pad_token, int B, int T, int rows_expect, int* off,
                    int* row_src, int* row_t, long long* ypk, int* err, hipStream_t s, int* ord = nullptr);
// pos[b] = position of the single stop token of row b; err[0] |= 1 if some row has != 1 stop tokens
int launch_find_stop(const long long* idx, int stop_token, int* pos, int* err, int B, int T, hipStream_t s);
int launch_gather_rows(const float* x, const int* pos, float* out, int B, int T, int C, hipStream_t s, const int* off = nullptr);
// dx[b, pos[b], :] += dout[b, :]
int launch_scatter_rows_add(const float* dout, const int* pos, float* dx, int B, int T, int C, hipStream_t s, const int* off = nullptr);
int launch_scatter_rows_bf16(const bf16_t* src, const int* pos, bf16_t* dst, int B, int T, int C, hipStream_t s, const int* off = nullptr);
// bad[b] = sum_t tokens[b,t] < 1
int launch_bad_rows(const long long* tokens, unsigned char* bad, int B, int T, hipStream_t s);
// inference decode (decode.hip)
// pos_dev: optional device pointer holding the position (overrides pos): lets a captured graph be replayed step after step
int launch_attn_decode(const bf16_t* qkv, bf16_t* cache, bf16_t* y, int B, int n_head, int head_size, int Tmax, int pos,
                       const int* pos_dev, hipStream_t s);
int launch_add_int(int* x, int v, int set, hipStream_t s);
int launch_topk_sample(const float* logits, long long ldl, int B, int V, int k, float inv_temp, const float* u,
                       long long* tok_out, int* stopped, int stop_token, int pad_token, hipStream_t s);
// batch tail (batch.hip)
int launch_batch_ncols(const long long* tok, int B, int S, int* ncols, hipStream_t s);
int launch_batch_tail(const long long* tok, int B, int S, int ncol, long long* tok_out, long long* y_out,
                      const long long* masked, int n_masked, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// fused lm_head cross-entropy pieces (lmhead.hip)
// ------------------------------------------------------------------------------------------------
// merges the per-tile partials into lse[M]; adds sum(lse - logit[target]) and the target count into scal[0], scal[1]
int launch_ce_finish(const float2* partial, int tiles_n, const bf16_t* a, long long lda, const bf16_t* W,
                     long long ldw, const long long* target, float* lse, float* scal, int M, int C, int V,
                     hipStream_t s);

// ------------------------------------------------------------------------------------------------
// E(3)-GNN kernels (gnn.hip)
// ------------------------------------------------------------------------------------------------
int launch_gnn_embed(const long long* atoms, const int* lut_ix, const int* lut_iy, const float* W, const float* b,
                     float* h32, bf16_t* h16, long long ld16, float* rstd, float* mask, int BA, int H,
                     hipStream_t s, int* err = nullptr);
int launch_gnn_node_res_silu(const float* u32, const long long* atoms, const int* lut_ix, const int* lut_iy, const float* W3c, long long ldw,
                             bf16_t* upre, bf16_t* t16, int BA, int H, hipStream_t s);
int launch_gnn_onehot_wgrad(const long long* atoms, const int* lut_ix, const int* lut_iy, const bf16_t* du, float* dW, long long ldw, int BA,
                            int H, hipStream_t s);
int launch_gnn_embed_bwd(const long long* atoms, const int* lut_ix, const int* lut_iy, const float* de,
                         float* dW, float* db, int BA, int H, hipStream_t s);
int launch_gnn_geom(const float* coords, const float* mask, float cutoff, float* d2, float* w, int B, int A,
                    hipStream_t s);
// Compacted edge list (the reference's neighbour list, e_gcl_sparse.py:27-77, built on the device, no host sync):
// edges in receiver-major order; seg[bj] .. seg[bj+1] = the edges received by node row bj; e_rev[e] = the edge (k -> j) of
// edge e = (j -> k) (the edge set is symmetric); n_edges[0] = E.  pos is a B*A*A int scratch.
int launch_gnn_compact(const float* w_dense, const float* d2_dense, int* seg, int* n_edges, int* e_bj, int* e_bk, int* e_rev,
                       float* e_d2, float* e_w, int* pos, int B, int A, hipStream_t s);
int launch_gnn_edge_pre_c(const bf16_t* P, long long ldp, const int* seg, const int* e_bk, const float* e_d2, const float* w1c,
                          long long w1c_stride, const float* b1, bf16_t* e1, int BA, int H, hipStream_t s);
int launch_gnn_edge_reduce_c(const bf16_t* s2, const int* seg, const float* e_w, bf16_t* mi, long long ldmi, int BA, int H, hipStream_t s);
// the three forward edge steps (edge_pre_c, the second edge Linear, edge_reduce_c) as one weight-resident launch (gemm_rb16.hip), H = 256
int launch_gnn_edge_fwd_fused(const bf16_t* P, long long ldp, const int* seg, const int* e_bk, const float* e_d2, const float* e_w,
                              const float* w1c, long long w1c_stride, const float* b1, const bf16_t* W3, long long ldw, const float* b3,
                              bf16_t* e1, bf16_t* s2, bf16_t* mi, long long ldmi, int BA, int H, hipStream_t s);
int launch_gnn_edge_reduce_bwd_c(const bf16_t* dmi, long long lddmi, const bf16_t* s2, const int* seg, const float* e_w,
                                 bf16_t* ds2, int BA, int H, hipStream_t s);
int launch_gnn_edge_pre_bwd_c(const bf16_t* dpre, const int* seg, const int* e_rev, const float* e_d2, bf16_t* dP, long long lddp,
                              float* dw1c, long long dw1c_stride, float* db1, int BA, int H, hipStream_t s);
int launch_gnn_readout(const float* o, const float* mask, float* hp, int B, int A, int H, hipStream_t s);
int launch_gnn_readout_bwd(const float* dhp, const float* mask, bf16_t* dout, int B, int A, int H, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// small fp32 elementwise + contrastive heads (loss.hip)
// ------------------------------------------------------------------------------------------------
int launch_silu_fwd(const float* x, float* y, long long n, hipStream_t s);
int launch_silu_bwd(const float* x, const float* dy, float* dx, long long n, int accumulate, hipStream_t s);
int launch_select_rows(const unsigned char* use_a, const float* a, const float* b, float* out, int B, int C,
                       hipStream_t s);
// da[b] += use_a[b] ? dout[b] : 0 ; db[b] += use_a[b] ? 0 : dout[b]
int launch_select_rows_bwd(const unsigned char* use_a, const float* dout, float* da, float* db, int B, int C,
                           hipStream_t s);
int launch_axpy(const float* x, float* y, float alpha, long long n, hipStream_t s);  // y += alpha*x
// rows: logits[R, N] (f32, in place -> dlogits).  label of row r = label0 + r, ignored if bad[label].
// sum of row losses -> scal_out[0] (+=); dlogits = (softmax - onehot) * (*gscale_ptr-free) handled by `gscale`.
int launch_infonce_rows2(float* logits, float* logits2, long long ld, int R, int N, int label0, const unsigned char* bad,
                         float* loss_sum, float* loss_sum2, const float* inv_count, float gscale, hipStream_t s);   // both directions in one launch
int launch_infonce_rows(float* logits, long long ld, int R, int N, int label0, const unsigned char* bad,
                        float* loss_sum, const float* inv_count, float gscale, hipStream_t s);
int launch_count_valid(const unsigned char* bad, int n, float* out_count, float* out_inv, hipStream_t s);
// Barlow-Twins head pieces (loss.hip)
int launch_colsum2(const float* a, const float* b2, const unsigned char* bad, float* out, int B, int E, hipStream_t s);
int launch_center_rows(const float* z, const unsigned char* bad, const float* sum, const float* count, float* zc, int B, int E, hipStream_t s);
int launch_standardize(const float* z, const unsigned char* bad, const float* stats, const float* count, float* zt, float* rsigma,
                       int B, int E, hipStream_t s);
int launch_barlow_dc(float* C, const float* count, float lam, float* loss, int E, hipStream_t s);
int launch_standardize_bwd(const float* dzt, const float* zt, const unsigned char* bad, const float* rsigma, const float* m,
                           const float* count, float scale, float* dz, int B, int E, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// optimiser (optim.hip)
// ------------------------------------------------------------------------------------------------
int launch_grad_sqnorm(const float* g, long long n, float* partial, int n_partial, float* out_norm, float max_norm,
                       float* out_coef, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* shadow, long long n, float lr, float b1,
                 float b2, float eps, float wd, int step, const float* coef, float gscale, hipStream_t s, const int* skip = nullptr);
int launch_cast_bf16(const float* src, bf16_t* dst, long long n, hipStream_t s);
// dst[c*ld_dst + r] = bf16(src[r*ld_src + c]) for r<rows, c<cols; pad rows [cols, cols_pad) x... see optim.hip
int launch_transpose_cast(const float* src, long long ld_src, bf16_t* dst, long long ld_dst, int rows, int cols,
                          hipStream_t s);
int launch_pack_rows_cast(const float* src, long long ld_src, bf16_t* dst, long long ld_dst, int rows, int cols,
                          hipStream_t s);

// one-launch rebuild of the transposed / packed bf16 weight shadows (optim.hip); jobs + prefix tile counts live on the device
struct ShadowJob {
  long long src_off, dst_off;   // element offsets into the f32 parameter buffer / the bf16 shadow buffer
  int ld_src, ld_dst, rows, cols, transpose;
};
int launch_shadow_jobs(const ShadowJob* jobs, const int* tile_start, int n_jobs, int n_tiles, const float* P, bf16_t* S,
                       hipStream_t s);
