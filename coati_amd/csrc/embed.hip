// Token-side gather/scatter kernels (HBM-bound row copies; one 64-lane wave per token row, float4 lanes).
//   embed_fwd        x[b,t,:] = tok_emb[idx[b,t]]  or  injection[b] where idx == [UNK]
//                    (reference basic_transformer.py:80-81, smiles_xformer.py:444-448)
//   embed_bwd        scatter-add of dx into the embedding table / the injected row
//   find_stop        position of the single [STOP] per row (smiles_xformer.py:50-68)
//   gather/scatter   row pick at the stop position and its transpose
//   bad_rows         augmented_tokens.sum(-1) < 1 (clip_e2e.py:813)
#include "kernels.h"

// mode 0: the gather (with the injection when given).  norm_embed models (the embedding is followed by a LayerNorm and the injection
// overwrites its OUTPUT, basic_transformer.py:72-76, smiles_xformer.py:442-448) run the gather without injection, the LayerNorm, and
// then mode 1: ONLY the [UNK] rows are written (the injected vector); mode 2: ONLY the [UNK] rows are written, with zeros (the
// backward clears the gradient rows the injection replaced before the LayerNorm backward sums over the rows)
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ idx, const float* __restrict__ table,
                                                        const float* __restrict__ injection, int unk, float* __restrict__ x,
                                                        int M, int T, int C, int V, const int* __restrict__ row_src, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  if (m >= M) return;
  const int ms = row_src ? row_src[m] : m;   // packed rows: output row m is slot ms = b * T + t of the padded [B, T] token matrix
  long long tok = idx[ms];
  const float* src;
  if (mode != 0) {
    if (tok != unk) return;
    float* dstz = x + (long long)m * C;
    if (mode == 2) {
      for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(dstz + c) = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    src = injection + (long long)(ms / T) * C;
  } else if (injection != nullptr && tok == unk) {
    src = injection + (long long)(ms / T) * C;
  } else {
    if (tok < 0) tok = 0;
    if (tok >= V) tok = V - 1;
    src = table + tok * C;
  }
  float* dst = x + (long long)m * C;
  for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
}

int launch_embed_fwd(const long long* idx, const float* table, const float* injection, int unk_token, float* x,
                     int B, int T, int C, int V, hipStream_t s, const int* row_src, int rows, int mode) {
  COATI_CHECK_ARG(idx && x && (mode != 0 || table) && (mode != 1 || injection) && mode >= 0 && mode <= 2, "embed_fwd: null operand / bad mode");
  COATI_CHECK_SHAPE(B > 0 && T > 0 && C % 4 == 0 && V > 0, "embed_fwd: unsupported shape");
  const int M = row_src ? rows : B * T;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, idx, table, injection, unk_token, x, M, T, C, V, row_src, mode);
  COATI_LAUNCH_CHECK("embed_fwd");
  return COATI_OK;
}

// One wave walks EMB_RUN consecutive molecules at ONE token position t.  The special tokens sit at fixed positions
// ([CLIP]/[SMILES] at t = 0, [UNK] at 1, ...), so consecutive rows of a walk usually target the same table row: the
// wave keeps a running sum in registers and flushes it (atomics) only when the destination changes.  The 1024-way
// same-address contention on the special-token rows becomes B / EMB_RUN-way (measured 243 -> see profiles).
#define EMB_RUN 32
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ idx, const float* __restrict__ dx,
                                                        float* __restrict__ dtable, float* __restrict__ dinj, int unk,
                                                        int B, int T, int C, int V, const int* __restrict__ off, int inj_only) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x;
  const int b0 = (blockIdx.y * 4 + wave) * EMB_RUN;
  if (b0 >= B) return;
  const int b1 = (b0 + EMB_RUN < B) ? b0 + EMB_RUN : B;
  // lane -> channels lane, lane + 64, ...: every atomic instruction then covers 64 CONSECUTIVE floats (two full 128-B
  // lines).  With a float4 per lane the four atomics of a lane each touched a quarter of 8 lines: 13x slower.
  for (int c0 = 0; c0 < C; c0 += 256) {
    float* cur = nullptr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto flush = [&]() {
      // rows behind the [STOP] token (all the padding) carry exactly-zero gradient under causal attention:
      // adding 0.0f is a no-op, so skipping it is exact and removes the [PAD]-row atomic hot spot.
      if (cur == nullptr) return;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + lane + 64 * i;
        if (c < C && acc[i] != 0.f) atomicAdd(cur + c, acc[i]);
      }
    };
    for (int b = b0; b < b1; ++b) {
      long long m = (long long)b * T + t;
      long long tok = idx[m];
      if (off != nullptr) {   // packed rows: molecule b owns rows off[b] .. off[b + 1]; positions behind its last token do not exist
        const int o = off[b];
        if (t >= off[b + 1] - o) continue;
        m = o + t;
      }
      float* dst;
      if (inj_only && tok != unk) continue;      // (norm_embed: the table's share goes through the LayerNorm backward first)
      if (dinj != nullptr && tok == unk) {
        dst = dinj + (long long)b * C;
      } else {
        if (tok < 0) tok = 0;
        if (tok >= V) tok = V - 1;
        dst = dtable + tok * C;
      }
      float g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + lane + 64 * i;
        g[i] = c < C ? dx[m * C + c] : 0.f;
      }
      if (dst != cur) {   // wave-uniform
        flush();
        cur = dst;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = g[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += g[i];
      }
    }
    flush();
  }
}

int launch_embed_bwd(const long long* idx, const float* dx, float* dtable, float* dinjection, int unk_token,
                     int B, int T, int C, int V, hipStream_t s, const int* off, int inj_only) {
  COATI_CHECK_ARG(idx && dx && (dtable || inj_only) && (!inj_only || dinjection), "embed_bwd: null operand");
  COATI_CHECK_SHAPE(B > 0 && T > 0 && C % 4 == 0 && V > 0, "embed_bwd: unsupported shape");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(T, cdiv(B, 4 * EMB_RUN)), dim3(256), 0, s, idx, dx, dtable, dinjection, unk_token, B, T, C, V, off, inj_only);
  COATI_LAUNCH_CHECK("embed_bwd");
  return COATI_OK;
}

// one wave per row: the row's tokens in coalesced 8-B loads, the matches as ballots (first position = lowest set bit of the first
// non-empty ballot, count = popcounts); one thread per row walked T dependent strided loads (22 us for 1024 x 80)
__global__ __launch_bounds__(256) void find_stop_kernel(const long long* __restrict__ idx, int stop, int* __restrict__ pos, int* __restrict__ err,
                                                        int B, int T) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int n = 0, p = 0;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    const bool hit = t < T && idx[(long long)b * T + t] == stop;
    const unsigned long long m = __ballot(hit);
    if (m != 0ull) {
      if (n == 0) p = t0 + __builtin_ctzll(m);
      n += __builtin_popcountll(m);
    }
  }
  if (lane == 0) {
    pos[b] = p;
    if (n != 1) atomicOr(err, 1);
  }
}

int launch_find_stop(const long long* idx, int stop_token, int* pos, int* err, int B, int T, hipStream_t s) {
  COATI_CHECK_ARG(idx && pos && err, "find_stop: null operand");
  hipLaunchKernelGGL(find_stop_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, idx, stop_token, pos, err, B, T);
  COATI_LAUNCH_CHECK("find_stop");
  return COATI_OK;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ pos,
                                                          float* __restrict__ out, int B, int T, int C, const int* __restrict__ off) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* src = x + ((off ? (long long)off[b] : (long long)b * T) + pos[b]) * C;
  for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(out + (long long)b * C + c) = *reinterpret_cast<const float4*>(src + c);
}

int launch_gather_rows(const float* x, const int* pos, float* out, int B, int T, int C, hipStream_t s, const int* off) {
  COATI_CHECK_ARG(x && pos && out, "gather_rows: null operand");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, x, pos, out, B, T, C, off);
  COATI_LAUNCH_CHECK("gather_rows");
  return COATI_OK;
}

__global__ __launch_bounds__(256) void scatter_rows_add_kernel(const float* __restrict__ dout, const int* __restrict__ pos,
                                                               float* __restrict__ dx, int B, int T, int C, const int* __restrict__ off) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float* dst = dx + ((off ? (long long)off[b] : (long long)b * T) + pos[b]) * C;
  for (int c = lane * 4; c < C; c += 256) {
    float4 d = *reinterpret_cast<float4*>(dst + c);
    const float4 g = *reinterpret_cast<const float4*>(dout + (long long)b * C + c);
    d.x += g.x; d.y += g.y; d.z += g.z; d.w += g.w;
    *reinterpret_cast<float4*>(dst + c) = d;
  }
}

// dst[(row of sequence b at position pos[b])] = src[b] (bf16 rows; dst was zeroed by the caller)
__global__ __launch_bounds__(256) void scatter_rows_bf16_kernel(const bf16_t* __restrict__ src, const int* __restrict__ pos,
                                                                bf16_t* __restrict__ dst, int B, int T, int C, const int* __restrict__ off) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  bf16_t* d = dst + ((off ? (long long)off[b] : (long long)b * T) + pos[b]) * C;
  for (int c = lane * 8; c < C; c += 512) *reinterpret_cast<uint4*>(d + c) = *reinterpret_cast<const uint4*>(src + (long long)b * C + c);
}
int launch_scatter_rows_bf16(const bf16_t* src, const int* pos, bf16_t* dst, int B, int T, int C, hipStream_t s, const int* off) {
  COATI_CHECK_ARG(src && pos && dst && C % 8 == 0, "scatter_rows_bf16: null operand / C % 8");
  hipLaunchKernelGGL(scatter_rows_bf16_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, src, pos, dst, B, T, C, off);
  COATI_LAUNCH_CHECK("scatter_rows_bf16");
  return COATI_OK;
}

int launch_scatter_rows_add(const float* dout, const int* pos, float* dx, int B, int T, int C, hipStream_t s, const int* off) {
  COATI_CHECK_ARG(dout && pos && dx, "scatter_rows_add: null operand");
  hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, dout, pos, dx, B, T, C, off);
  COATI_LAUNCH_CHECK("scatter_rows_add");
  return COATI_OK;
}

// ------------------------------------------------------------------------------------------------------
// Packed rows ("ragged" batches).  clip_ar_xform pads every row of a batch to the longest one (clip_e2e.py:288-330), and the
// reference computes the padding: under causal attention a position behind a row's last token influences no earlier
// position, its targets are -1 (ignored by the AR loss, train_coati.py:260-265) and the encoder reads the [STOP] position
// only (smiles_xformer.py:50-68) -- so those positions contribute EXACTLY zero to both losses and to every gradient.  The
// engine therefore runs the transformer passes on the concatenation of the rows' real prefixes:
//   len[b]     = 1 + last position whose token is not [PAD] or whose target is not -1
//   off[b]     = exclusive prefix sum, off[B] = number of packed rows (must equal the count the caller computed on the host)
//   row_src[m] = b * T + t : slot of packed row m in the padded [B, T] matrices;  row_t[m] = t (rotary position)
//   ypk[m]     = y_next[b, t]
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seq_len_kernel(const long long* __restrict__ tok, const long long* __restrict__ y, int pad,
                                                      int* __restrict__ len, int B, int T) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int last = 0;
  for (int t = lane; t < T; t += 64) {
    const bool live = tok[(long long)b * T + t] != pad || (y != nullptr && y[(long long)b * T + t] >= 0);
    if (live) last = t + 1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(last, o, 64); last = v > last ? v : last; }
  if (lane == 0) len[b] = last;
}
// exclusive scan of len[0..B) in place -> off[0..B], off[B] = total; err |= 2 when the total differs from the host's count
// ord (optional): the sequences in order of DESCENDING 16-row block count (a counting sort over 0 .. 8+ blocks; ties in arbitrary order) --
// the launch schedule of the attention kernels (attention16.hip): workgroups are dispatched in index order, so the long sequences
// go first and the short ones fill the tail of the launch.  A schedule only: no result depends on it.
__global__ __launch_bounds__(1024) void seq_scan_kernel(int* __restrict__ off, int B, int expect, int* __restrict__ err, int* __restrict__ ord) {
  __shared__ int part[1024];
  __shared__ int cnt[10], cur[10];
  const int t = threadIdx.x, per = (B + 1023) / 1024;
  const int lo = t * per, hi = lo + per < B ? lo + per : B;
  if (t < 10) cnt[t] = 0;
  __syncthreads();
  int sum = 0;
  for (int i = lo; i < hi; ++i) {
    const int l = off[i];
    sum += l;
    if (ord != nullptr) { const int nb = (l + 15) >> 4; atomicAdd(&cnt[9 - (nb < 9 ? nb : 9)], 1); }
  }
  part[t] = sum;
  __syncthreads();
  if (t == 0) { int run = 0; for (int k = 0; k < 10; ++k) { cur[k] = run; run += cnt[k]; } }
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;
  for (int i = lo; i < hi; ++i) {
    const int l = off[i];
    off[i] = run;
    run += l;
    if (ord != nullptr) { const int nb = (l + 15) >> 4; ord[atomicAdd(&cur[9 - (nb < 9 ? nb : 9)], 1)] = i; }
  }
  if (t == 1023) {
    off[B] = part[1023];
    if (part[1023] != expect) atomicOr(err, 2);
  }
}
__global__ void seq_fill_kernel(const int* __restrict__ off, const long long* __restrict__ y, int* __restrict__ row_src,
                                int* __restrict__ row_t, long long* __restrict__ ypk, int B, int T, int cap) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)B * T) return;
  // rows [total, cap) exist only when the host's count was too LARGE (flagged by the scan): they still have to index inside the
  // matrices (row 0, position 0, no target) -- written here instead of by three memsets in front of every pack
  {
    const long long m = (long long)off[B] + g;
    if (m < cap) {
      row_src[m] = 0;
      row_t[m] = 0;
      if (ypk != nullptr) ypk[m] = -1;
    }
  }
  const int b = (int)(g / T), t = (int)(g - (long long)b * T);
  const int o = off[b];
  if (t >= off[b + 1] - o) return;
  const int m = o + t;
  if (m >= cap) return;   // (only when the host's count was too small: flagged by the scan)
  row_src[m] = (int)g;
  row_t[m] = t;
  if (ypk != nullptr) ypk[m] = y[g];
}
int launch_seq_pack(const long long* tok, const long long* y, int pad_token, int B, int T, int rows_expect, int* off,
                    int* row_src, int* row_t, long long* ypk, int* err, hipStream_t s, int* ord) {
  COATI_CHECK_ARG(tok && off && row_src && row_t && err && (ypk == nullptr || y != nullptr), "seq_pack: null operand");
  COATI_CHECK_SHAPE(B > 0 && T > 0 && rows_expect > 0 && rows_expect <= (long long)B * T, "seq_pack: bad row count %d for %d x %d", rows_expect, B, T);
  hipLaunchKernelGGL(seq_len_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, tok, y, pad_token, off, B, T);
  hipLaunchKernelGGL(seq_scan_kernel, dim3(1), dim3(1024), 0, s, off, B, rows_expect, err, ord);
  hipLaunchKernelGGL(seq_fill_kernel, dim3(cdiv((long long)B * T, 256)), dim3(256), 0, s, off, y, row_src, row_t, ypk, B, T, rows_expect);
  COATI_LAUNCH_CHECK("seq_pack");
  return COATI_OK;
}

__global__ void bad_rows_kernel(const long long* __restrict__ tok, unsigned char* __restrict__ bad, int B, int T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  long long sum = 0;
  for (int t = 0; t < T; ++t) sum += tok[(long long)b * T + t];
  bad[b] = sum < 1 ? 1 : 0;
}

int launch_bad_rows(const long long* tokens, unsigned char* bad, int B, int T, hipStream_t s) {
  COATI_CHECK_ARG(tokens && bad, "bad_rows: null operand");
  hipLaunchKernelGGL(bad_rows_kernel, dim3(cdiv(B, 128)), dim3(128), 0, s, tokens, bad, B, T);
  COATI_LAUNCH_CHECK("bad_rows");
  return COATI_OK;
}

// ------------------------------------------------------------------------------------------------------
// cross-entropy finish (lmhead): merge per-tile (max, sumexp) partials into lse[row]; target logit by a
// direct bf16 dot product (same operands the MFMA saw); accumulate sum(lse - logit_t) and the count.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_finish_kernel(const float2* __restrict__ partial, int tiles_n,
                                                        const bf16_t* __restrict__ a, long long lda,
                                                        const bf16_t* __restrict__ W, long long ldw,
                                                        const long long* __restrict__ target, float* __restrict__ lse,
                                                        float* __restrict__ scal, int M, int C, int V) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float loss_acc = 0.f, cnt_acc = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    float mx = -INFINITY;
    for (int t = lane; t < tiles_n; t += 64) mx = fmaxf(mx, partial[(long long)row * tiles_n + t].x);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int t = lane; t < tiles_n; t += 64) {
      const float2 p = partial[(long long)row * tiles_n + t];
      sm += p.y * __expf(p.x - mx);
    }
    sm = wave_sum(sm);
    const float l = mx + __logf(sm);
    if (lane == 0) lse[row] = l;
    const long long tgt = target[row];
    if (tgt >= 0 && tgt < V) {
      float dot = 0.f;
      for (int c = lane * 4; c < C; c += 256) {
        const uint2 ua = *reinterpret_cast<const uint2*>(a + (long long)row * lda + c);
        const uint2 uw = *reinterpret_cast<const uint2*>(W + tgt * ldw + c);
        dot += bflo(ua.x) * bflo(uw.x) + bfhi(ua.x) * bfhi(uw.x) + bflo(ua.y) * bflo(uw.y) + bfhi(ua.y) * bfhi(uw.y);
      }
      dot = wave_sum(dot);
      loss_acc += l - dot;
      cnt_acc += 1.f;
    }
  }
  if (lane == 0) { red[0][wave] = loss_acc; red[1][wave] = cnt_acc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(scal + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(scal + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

int launch_ce_finish(const float2* partial, int tiles_n, const bf16_t* a, long long lda, const bf16_t* W,
                     long long ldw, const long long* target, float* lse, float* scal, int M, int C, int V,
                     hipStream_t s) {
  COATI_CHECK_ARG(partial && a && W && target && lse && scal, "ce_finish: null operand");
  COATI_CHECK_SHAPE(C % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, "ce_finish: alignment");
  int blocks = cdiv(M, 4);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(ce_finish_kernel, dim3(blocks), dim3(256), 0, s, partial, tiles_n, a, lda, W, ldw, target, lse, scal, M, C, V);
  COATI_LAUNCH_CHECK("ce_finish");
  return COATI_OK;
}
