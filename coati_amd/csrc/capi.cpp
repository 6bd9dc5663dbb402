// C ABI wrappers of the fine-grained operators (declared in include/coati_hip.h).
#include <string.h>

#include "../../include/coati_hip.h"
#include <vector>
#include "kernels.h"

#define S_(x) ((hipStream_t)(x))
#define LL(x) (reinterpret_cast<const long long*>(x))

extern "C" {

int coati_gemm_nt(const void* A, int a_f32, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K,
                  void* C, int64_t ldc, int n_store, const float* bias, const void* aux_in, void* aux_out,
                  int64_t ld_aux, int epi, void* stream) {
  COATI_CHECK_ARG((epi >= EPI_BF16 && epi <= EPI_ACC_F32) || epi == EPI_GELU_GRAD || epi == EPI_MUL_AUX, "coati_gemm_nt: epilogue %d is not a plain epilogue", epi);
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc; a.n_store = n_store;
  a.bias = bias; a.aux_in = aux_in; a.aux_out = aux_out; a.ld_aux = ld_aux;
  return launch_gemm_nt(a, a_f32, epi, S_(stream));
}

#ifdef COATI_EXPERIMENTAL
int coati_mlp_fwd(const float* x, const float* gamma, const float* beta, uint16_t* a2, float* mean, float* rstd, const uint16_t* W1,
                  const float* b1, const uint16_t* W2, const float* b2, uint16_t* g, uint8_t* codes, float* out, int M, void* stream) {
  COATI_CHECK_SHAPE(M >= 1 && M <= 65536, "mlp_fwd: M=%d out of range", M);
  Mlp64Args a;
  a.x = x; a.ldx = 256; a.gamma = gamma; a.beta = beta; a.a2 = a2; a.mean = mean; a.rstd = rstd; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2;
  a.g = g; a.codes = codes; a.out = out; a.ldo = 256; a.M = M;
  return launch_mlp64_fwd(a, S_(stream));
}
#endif

int coati_gemm_lnbwd(const uint16_t* dY, int64_t lda, const uint16_t* WT, int64_t ldw, int M, int K, const float* x, const float* mean,
                     const float* rstd, const float* gamma, const float* dres, float* dx, uint16_t* dx16, float* partial,
                     int32_t* n_partial_rows, const uint16_t* chain_W, uint16_t* chain_C, void* stream) {
  COATI_CHECK_ARG(dY && WT && x && mean && rstd && gamma && dres && dx && partial && n_partial_rows, "gemm_lnbwd: null argument");
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = dY; a.lda = lda; a.B = WT; a.ldb = ldw; a.M = M; a.N = 256; a.K = K; a.C = dx; a.ldc = 256; a.aux_in = dres; a.ld_aux = 256;
  a.aux_out = dx16; a.lnb_x = x; a.lnb_ldx = 256; a.lnb_mean = mean; a.lnb_rstd = rstd; a.lnb_gamma = gamma; a.lnb_partial = partial;
  COATI_CHECK_ARG((chain_W == nullptr) == (chain_C == nullptr) && (chain_W == nullptr || dx16 != nullptr), "gemm_lnbwd: the chained product needs chain_W, chain_C and dx16");
  a.chain_W = chain_W; a.chain_ldw = 256; a.chain_C = chain_C; a.chain_ldc = 256;
  int nwg = 0;
  COATI_CHECK_SHAPE(gemm_ring_lnbwd_supported(a, &nwg), "gemm_lnbwd: unsupported shape M=%d K=%d (256 columns, 40 961 .. 57 344 rows, K %% 64 == 0)", M, K);
  *n_partial_rows = nwg;
  return launch_gemm_nt(a, 0, EPI_LNBWD, S_(stream));
}

int coati_quant_mx8(const void* x, int x_f32, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scales, int M, int K, void* stream) {
  return launch_quant_mx8(x, x_f32, ldx, q, ldq, scales, M, K, S_(stream));
}
int coati_gemm_mx8(const uint8_t* A, int64_t lda, const uint8_t* a_scales, const uint8_t* W, int64_t ldw, const uint8_t* w_scales,
                   int M, int N, int K, void* C, int64_t ldc, const float* bias, const void* aux_in, void* aux_out, int64_t ld_aux,
                   int epi, void* stream) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = reinterpret_cast<const bf16_t*>(W); a.ldb = ldw; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc;
  a.bias = bias; a.aux_in = aux_in; a.aux_out = aux_out; a.ld_aux = ld_aux;
  COATI_CHECK_ARG(epi != EPI_QKV_ROPE, "gemm_mx8: the rotary epilogue is an engine-internal call");
  return launch_gemm_mx8(a, a_scales, w_scales, epi, S_(stream));
}
int coati_gemm_ce_partial(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, int M, int V, int K,
                          void* partial, void* stream) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = W; a.ldb = ldw; a.M = M; a.N = V; a.K = K; a.partial = reinterpret_cast<float2*>(partial);
  return launch_gemm_nt(a, 0, EPI_CE_PARTIAL, S_(stream));
}

int coati_ce_finish(const void* partial, int tiles_n, const uint16_t* A, int64_t lda, const uint16_t* W,
                    int64_t ldw, const int64_t* target, float* lse, float* scal, int M, int K, int V, void* stream) {
  return launch_ce_finish(reinterpret_cast<const float2*>(partial), tiles_n, A, lda, W, ldw, LL(target), lse, scal, M, K, V, S_(stream));
}

int coati_gemm_ce_bwd(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, int M, int V, int K,
                      uint16_t* dlogits, int64_t ldd, int n_store, const float* lse, const int64_t* target,
                      const float* scal, void* stream) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = W; a.ldb = ldw; a.M = M; a.N = V; a.K = K; a.C = dlogits; a.ldc = ldd; a.n_store = n_store;
  a.lse = lse; a.target = LL(target); a.scal = scal;
  return launch_gemm_nt(a, 0, EPI_CE_BWD, S_(stream));
}

int coati_wgrad(const void* A, int a_f32, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K,
                float* dW, int64_t ldw, float* dbias, int n_out, void* stream) {
  WgradArgs a;
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.M = M; a.N = N; a.K = K; a.dW = dW; a.ldw = ldw; a.dbias = dbias; a.n_out = n_out;
  return launch_wgrad(a, a_f32, S_(stream));
}

int64_t coati_wgrad_grouped_workspace_bytes(int n_problems, const int* N, const int* K, int tile_size) {
  if (n_problems <= 0 || !N || !K || (tile_size != 128 && tile_size != 256)) return 0;
  int64_t tiles = 0;
  for (int i = 0; i < n_problems; ++i) tiles += (int64_t)cdiv(N[i], tile_size) * cdiv(K[i], tile_size);
  return tiles * (int64_t)sizeof(WgradTile);
}

int coati_wgrad_grouped(int n_problems, const uint16_t* const* A, const int64_t* lda, const uint16_t* const* B, const int64_t* ldb,
                        int M, const int* N, const int* K, float* const* dW, const int64_t* ldw, float* const* dbias, int tile_size,
                        void* workspace, int64_t workspace_bytes, void* stream) {
  COATI_CHECK_ARG(n_problems > 0 && A && lda && B && ldb && N && K && dW && ldw && dbias && workspace, "wgrad_grouped: null argument");
  std::vector<WgradTile> tab;
  for (int i = 0; i < n_problems; ++i) {
    WgradArgs a;
    a.A = A[i]; a.lda = lda[i]; a.B = B[i]; a.ldb = ldb[i]; a.M = M; a.N = N[i]; a.K = K[i]; a.dW = dW[i]; a.ldw = ldw[i]; a.dbias = dbias[i]; a.n_out = 0;
    COATI_TRY(wgrad_table_append(tab, a, tile_size));
  }
  COATI_CHECK_ARG((int64_t)(tab.size() * sizeof(WgradTile)) <= workspace_bytes, "wgrad_grouped: workspace too small (%zu tiles)", tab.size());
  hipStream_t s = S_(stream);
  // the tile table goes into the CALLER's device workspace (the library allocates nothing); `tab` lives on this frame, so the
  // upload is waited for before returning (this stand-alone entry point is not on the training step's path: the engine
  // caches its tables, engine.cpp xformer_wgrad_group)
  if (hipMemcpyAsync(workspace, tab.data(), tab.size() * sizeof(WgradTile), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    coati_set_error("wgrad_grouped: table upload failed");
    return COATI_EHIP;
  }
  return launch_wgrad_table(reinterpret_cast<const WgradTile*>(workspace), (int)tab.size(), s, tile_size);
}

int coati_sgemm(const float* A, int64_t ars, int64_t acs, const float* B, int64_t brs, int64_t bcs, float* C,
                int64_t ldc, int M, int N, int K, const float* bias, float alpha, int accumulate, void* stream) {
  return launch_sgemm(A, ars, acs, B, brs, bcs, C, ldc, M, N, K, bias, alpha, accumulate, S_(stream));
}

int coati_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, uint16_t* y16,
                        int64_t ld16, float* y32, int64_t ld32, float* mean, float* rstd, int M, int C, void* stream) {
  return launch_layernorm_fwd(x, ldx, gamma, beta, y16, ld16, y32, ld32, mean, rstd, M, C, S_(stream));
}
int coati_layernorm_bwd(const void* dy, int dy_f32, int64_t lddy, const float* x, int64_t ldx, int x_is_xhat,
                        const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                        uint16_t* dx16, float* dgamma, float* dbeta, float* partial, int M, int C, void* stream) {
  return launch_layernorm_bwd(dy, dy_f32, lddy, x, ldx, x_is_xhat, mean, rstd, gamma, dres, dx, dx16, dgamma, dbeta, partial, M, C, S_(stream));
}

int coati_attn_fwd(const uint16_t* qkv, uint16_t* y, float* lse, int B, int T, int n_head, void* stream) {
  return launch_attn_fwd(qkv, y, lse, B, T, n_head, 16, S_(stream));
}
int coati_attn_fwd_hs(const uint16_t* qkv, uint16_t* y, float* lse, int B, int T, int n_head, int head_size, void* stream) {
  return launch_attn_fwd(qkv, y, lse, B, T, n_head, head_size, S_(stream));
}
int coati_seq_pack(const int64_t* tok, const int64_t* y, int pad_token, int B, int T, int rows_expect, int32_t* off,
                   int32_t* row_src, int32_t* row_t, int64_t* ypk, int32_t* err, void* stream) {
  return launch_seq_pack(LL(tok), LL(y), pad_token, B, T, rows_expect, off, row_src, row_t, reinterpret_cast<long long*>(ypk), err, S_(stream));
}
int coati_attn_fwd_varlen(const uint16_t* qkv, uint16_t* y, float* lse, const int32_t* seq_off, int B, int T, int n_head,
                          int head_size, void* stream) {
  COATI_CHECK_ARG(seq_off, "attn_fwd_varlen: null seq_off");
  return launch_attn_fwd(qkv, y, lse, B, T, n_head, head_size, S_(stream), seq_off);
}
int coati_attn_bwd_varlen(const uint16_t* qkv, const uint16_t* y, const uint16_t* dy, const float* lse, float* dscratch,
                          uint16_t* dqkv, const float* cos_t, const float* sin_t, const int32_t* seq_off, int B, int T,
                          int n_head, int head_size, void* stream) {
  COATI_CHECK_ARG(seq_off, "attn_bwd_varlen: null seq_off");
  return launch_attn_bwd(qkv, y, dy, lse, dscratch, dqkv, cos_t, sin_t, B, T, n_head, head_size, S_(stream), seq_off);
}
#ifdef COATI_EXPERIMENTAL
int coati_attn_groups(const int32_t* seq_off, int B, int T, int32_t* grp, void* stream) {
  return launch_attn_groups(seq_off, B, T, grp, S_(stream));
}
int coati_attn_block_fwd(const float* x, float* xmid, const float* ln_g, const float* ln_b, float* mean, float* rstd, uint16_t* a1,
                         const uint16_t* Wqkv, const float* bqkv, const uint16_t* Wproj, const float* bproj, uint16_t* qkv, uint16_t* y,
                         float* lse, const float* cos_t, const float* sin_t, const int32_t* row_src, const int32_t* grp, int T, int M,
                         void* stream) {
  AttnBlockArgs a;
  a.x = x; a.xmid = xmid; a.ln_g = ln_g; a.ln_b = ln_b; a.mean = mean; a.rstd = rstd; a.a1 = a1; a.Wqkv = Wqkv; a.bqkv = bqkv;
  a.Wproj = Wproj; a.bproj = bproj; a.qkv = qkv; a.y = y; a.lse = lse; a.cos_t = cos_t; a.sin_t = sin_t; a.row_src = row_src;
  a.grp = grp; a.Tl = T; a.M = M;
  return launch_attn_block_fwd(a, S_(stream));
}
int coati_ab_probe_swap(uint32_t* out, void* stream) { return launch_ab_probe_swap(out, S_(stream)); }
#endif
int coati_gemm_qkv_rope(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, int M, int C,
                        uint16_t* qkv, int64_t ldc, const float* cos_t, const float* sin_t, int T, void* stream) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = W; a.ldb = ldw; a.M = M; a.N = 3 * C; a.K = C; a.C = qkv; a.ldc = ldc; a.bias = bias;
  a.rope_cos = cos_t; a.rope_sin = sin_t; a.rope_T = T; a.rope_C = C; a.rope_hs = 16;
  return launch_gemm_nt(a, 0, EPI_QKV_ROPE, S_(stream));
}
int coati_gemm_qkv_rope_hs(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, int M, int C,
                           uint16_t* qkv, int64_t ldc, const float* cos_t, const float* sin_t, int T, int head_size, void* stream) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = W; a.ldb = ldw; a.M = M; a.N = 3 * C; a.K = C; a.C = qkv; a.ldc = ldc; a.bias = bias;
  a.rope_cos = cos_t; a.rope_sin = sin_t; a.rope_T = T; a.rope_C = C; a.rope_hs = head_size;
  return launch_gemm_nt(a, 0, EPI_QKV_ROPE, S_(stream));
}
int coati_attn_bwd(const uint16_t* qkv, const uint16_t* y, const uint16_t* dy, const float* lse, float* dscratch, uint16_t* dqkv,
                   const float* cos_t, const float* sin_t, int B, int T, int n_head, void* stream) {
  return launch_attn_bwd(qkv, y, dy, lse, dscratch, dqkv, cos_t, sin_t, B, T, n_head, 16, S_(stream));
}
int coati_attn_bwd_hs(const uint16_t* qkv, const uint16_t* y, const uint16_t* dy, const float* lse, float* dscratch, uint16_t* dqkv,
                      const float* cos_t, const float* sin_t, int B, int T, int n_head, int head_size, void* stream) {
  return launch_attn_bwd(qkv, y, dy, lse, dscratch, dqkv, cos_t, sin_t, B, T, n_head, head_size, S_(stream));
}

int coati_embed_fwd(const int64_t* idx, const float* table, const float* injection, int unk_token, float* x, int B,
                    int T, int C, int V, void* stream) {
  return launch_embed_fwd(LL(idx), table, injection, unk_token, x, B, T, C, V, S_(stream));
}
int coati_embed_bwd(const int64_t* idx, const float* dx, float* dtable, float* dinjection, int unk_token, int B, int T,
                    int C, int V, void* stream) {
  return launch_embed_bwd(LL(idx), dx, dtable, dinjection, unk_token, B, T, C, V, S_(stream));
}
int coati_find_stop(const int64_t* idx, int stop_token, int32_t* pos, int32_t* err, int B, int T, void* stream) {
  return launch_find_stop(LL(idx), stop_token, pos, err, B, T, S_(stream));
}
int coati_gather_rows(const float* x, const int32_t* pos, float* out, int B, int T, int C, void* stream) {
  return launch_gather_rows(x, pos, out, B, T, C, S_(stream));
}
int coati_scatter_rows_add(const float* dout, const int32_t* pos, float* dx, int B, int T, int C, void* stream) {
  return launch_scatter_rows_add(dout, pos, dx, B, T, C, S_(stream));
}
int coati_bad_rows(const int64_t* tokens, uint8_t* bad, int B, int T, void* stream) {
  return launch_bad_rows(LL(tokens), bad, B, T, S_(stream));
}
int coati_silu(const float* x, float* y, int64_t n, void* stream) { return launch_silu_fwd(x, y, n, S_(stream)); }
int coati_attn_decode(const uint16_t* qkv, uint16_t* cache, uint16_t* y, int B, int n_head, int Tmax, int pos, void* stream) {
  return launch_attn_decode(qkv, cache, y, B, n_head, 16, Tmax, pos, nullptr, S_(stream));
}
int coati_attn_decode_hs(const uint16_t* qkv, uint16_t* cache, uint16_t* y, int B, int n_head, int head_size, int Tmax, int pos, void* stream) {
  return launch_attn_decode(qkv, cache, y, B, n_head, head_size, Tmax, pos, nullptr, S_(stream));
}
int coati_topk_sample(const float* logits, int64_t ldl, int B, int V, int k, float inv_temp, const float* u,
                      int64_t* tokens_out, int32_t* stopped, int stop_token, int pad_token, void* stream) {
  return launch_topk_sample(logits, ldl, B, V, k, inv_temp, u, reinterpret_cast<long long*>(tokens_out), stopped, stop_token, pad_token, S_(stream));
}
int coati_batch_ncols(const int64_t* tokens, int B, int n_seq, int32_t* ncols, void* stream) {
  return launch_batch_ncols(LL(tokens), B, n_seq, ncols, S_(stream));
}
int coati_batch_tail(const int64_t* tokens, int B, int n_seq, int ncol, int64_t* tokens_out, int64_t* y_next_out,
                     const int64_t* masked_ids, int n_masked, void* stream) {
  return launch_batch_tail(LL(tokens), B, n_seq, ncol, reinterpret_cast<long long*>(tokens_out),
                           reinterpret_cast<long long*>(y_next_out), LL(masked_ids), n_masked, S_(stream));
}

int coati_gnn_embed(const int64_t* atoms, const int32_t* lut_ix, const int32_t* lut_iy, const float* W, const float* b,
                    float* h32, uint16_t* h16, int64_t ld16, float* rstd, float* mask, int BA, int H, void* stream) {
  return launch_gnn_embed(LL(atoms), lut_ix, lut_iy, W, b, h32, h16, ld16, rstd, mask, BA, H, S_(stream));
}
int coati_gnn_geom(const float* coords, const float* mask, float cutoff, float* d2, float* w, int B, int A, void* stream) {
  return launch_gnn_geom(coords, mask, cutoff, d2, w, B, A, S_(stream));
}
int coati_gnn_compact(const float* w_dense, const float* d2_dense, int32_t* seg, int32_t* n_edges, int32_t* e_bj, int32_t* e_bk,
                      int32_t* e_rev, float* e_d2, float* e_w, int32_t* pos, int B, int A, void* stream) {
  return launch_gnn_compact(w_dense, d2_dense, seg, n_edges, e_bj, e_bk, e_rev, e_d2, e_w, pos, B, A, S_(stream));
}

int coati_gnn_edge_pre(const uint16_t* P, int64_t ldp, const int32_t* seg, const int32_t* e_bk, const float* e_d2, const float* w1c,
                       int64_t w1c_stride, const float* b1, uint16_t* e1, int BA, int H, void* stream) {
  return launch_gnn_edge_pre_c(P, ldp, seg, e_bk, e_d2, w1c, w1c_stride, b1, e1, BA, H, S_(stream));
}
int coati_gnn_edge_reduce(const uint16_t* s2, const int32_t* seg, const float* e_w, uint16_t* mi, int64_t ldmi, int BA, int H, void* stream) {
  return launch_gnn_edge_reduce_c(s2, seg, e_w, mi, ldmi, BA, H, S_(stream));
}

int coati_infonce_rows(float* logits, int64_t ld, int R, int N, int label0, const uint8_t* bad, float* loss_sum,
                       const float* inv_count, float gscale, void* stream) {
  return launch_infonce_rows(logits, ld, R, N, label0, bad, loss_sum, inv_count, gscale, S_(stream));
}

int coati_count_valid(const uint8_t* bad, int n, float* count, float* inv, void* stream) {
  return launch_count_valid(bad, n, count, inv, S_(stream));
}
int coati_colsum2(const float* a, const float* b2, const uint8_t* bad, float* out, int B, int E, void* stream) {
  return launch_colsum2(a, b2, bad, out, B, E, S_(stream));
}
int coati_center_rows(const float* z, const uint8_t* bad, const float* sum, const float* count, float* zc, int B, int E, void* stream) {
  return launch_center_rows(z, bad, sum, count, zc, B, E, S_(stream));
}
int coati_standardize(const float* z, const uint8_t* bad, const float* stats, const float* count, float* zt, float* rsigma,
                      int B, int E, void* stream) {
  return launch_standardize(z, bad, stats, count, zt, rsigma, B, E, S_(stream));
}
int coati_barlow_dc(float* C, const float* count, float lam, float* loss, int E, void* stream) {
  return launch_barlow_dc(C, count, lam, loss, E, S_(stream));
}
int coati_standardize_bwd(const float* dzt, const float* zt, const uint8_t* bad, const float* rsigma, const float* m,
                          const float* count, float scale, float* dz, int B, int E, void* stream) {
  return launch_standardize_bwd(dzt, zt, bad, rsigma, m, count, scale, dz, B, E, S_(stream));
}

int coati_grad_sqnorm(const float* g, int64_t n, float* partial, int n_partial, float* out_norm, float max_norm,
                      float* out_coef, void* stream) {
  return launch_grad_sqnorm(g, n, partial, n_partial, out_norm, max_norm, out_coef, S_(stream));
}
int coati_adamw(float* p, const float* g, float* m, float* v, uint16_t* shadow, int64_t n, float lr, float b1, float b2,
                float eps, float wd, int step, const float* coef, float gscale, void* stream) {
  return launch_adamw(p, g, m, v, shadow, n, lr, b1, b2, eps, wd, step, coef, gscale, S_(stream));
}

}  // extern "C"
