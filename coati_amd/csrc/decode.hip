// Inference decode step: KV-cached causal attention for ONE new token per sequence at
// head size 16, and top-k sampling of the next token (reference smiles_xformer.py:272-351 recomputes the whole prefix
// for every generated token; here the rotated keys and the values of earlier positions live in an HBM cache).
//
// Cache layout: [B][n_head][Tmax][k | v] bf16 (head size 16 or 32) -> one (b, head) sequence is a contiguous run of
// 64 / 128-B records, a wave streams it with one record per lane per pass.  HBM-bound: 4 * hs bytes per cached token
// per head per step.
#include "kernels.h"


template <int N>
__device__ __forceinline__ void load_bf16(const bf16_t* p, float* x) {
#pragma unroll
  for (int i = 0; i < N / 8; ++i) unpack8(*reinterpret_cast<const uint4*>(p + 8 * i), x + 8 * i);
}

// qkv: [B, 3C] bf16 of the new token (q, k already rotated by the QKV GEMM epilogue); y: [B, C] bf16.
// One wave per (b, head).  Appends (k, v) at position pos, attends to positions 0..pos.
template <int DHS>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ cache,
                                                          bf16_t* __restrict__ y, int B, int n_head, int Tmax, int pos_arg,
                                                          const int* __restrict__ pos_dev) {
  const int pos = pos_dev ? *pos_dev : pos_arg;
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= B * n_head) return;
  const int b = item / n_head, h = item - b * n_head;
  const int C = n_head * DHS;
  const bf16_t* row = qkv + (long long)b * 3 * C + h * DHS;
  float q[DHS], kn[DHS], vn[DHS];
  constexpr int REC = 2 * DHS, CH = DHS / 8;   // record = [k | v] halfs; 16-B chunks per operand
  load_bf16<DHS>(row, q);
  load_bf16<DHS>(row + C, kn);
  load_bf16<DHS>(row + 2 * C, vn);
  bf16_t* seq = cache + ((long long)item * Tmax) * REC;
  if (lane < 2 * CH) {   // append the new record
    const bf16_t* src = (lane < CH) ? row + C + lane * 8 : row + 2 * C + (lane - CH) * 8;
    *reinterpret_cast<uint4*>(seq + (long long)pos * REC + lane * 8) = *reinterpret_cast<const uint4*>(src);
  }
  // scores of this lane's keys (t = lane, lane + 64, ...); the newest key comes from registers, not from the cache
  float m = -INFINITY;
  float sc[4];
  float kv[4][DHS];   // values of this lane's keys
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = lane + 64 * i;
    sc[i] = -INFINITY;
    if (t <= pos) {
      float k[DHS];
      if (t == pos) {
#pragma unroll
        for (int d = 0; d < DHS; ++d) { k[d] = kn[d]; kv[i][d] = vn[d]; }
      } else {
        load_bf16<DHS>(seq + (long long)t * REC, k);
        load_bf16<DHS>(seq + (long long)t * REC + DHS, kv[i]);
      }
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DHS; ++d) s += q[d] * k[d];
      sc[i] = s * (DHS == 16 ? 0.25f : 0.17677669529663687f);   // 1 / sqrt(hs)
      m = fmaxf(m, sc[i]);
    } else {
#pragma unroll
      for (int d = 0; d < DHS; ++d) kv[i][d] = 0.f;
    }
  }
  m = wave_max(m);
  float l = 0.f, acc[DHS];
#pragma unroll
  for (int d = 0; d < DHS; ++d) acc[d] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float p = (sc[i] == -INFINITY) ? 0.f : __expf(sc[i] - m);
    l += p;
#pragma unroll
    for (int d = 0; d < DHS; ++d) acc[d] += p * kv[i][d];
  }
  l = wave_sum(l);
  const float inv = 1.0f / l;
#pragma unroll
  for (int d = 0; d < DHS; ++d) acc[d] = wave_sum(acc[d]) * inv;
  if (lane == 0) {
    bf16_t* dst = y + (long long)b * C + h * DHS;
#pragma unroll
    for (int i = 0; i < DHS / 8; ++i) *reinterpret_cast<uint4*>(dst + 8 * i) = pack8(acc + 8 * i);
  }
}

__global__ void add_int_kernel(int* x, int v, int set) { *x = set ? v : *x + v; }
// set != 0: *x = v; else *x += v   (the decode position lives in device memory for graph replay)
int launch_add_int(int* x, int v, int set, hipStream_t s) {
  hipLaunchKernelGGL(add_int_kernel, dim3(1), dim3(1), 0, s, x, v, set);
  COATI_LAUNCH_CHECK("add_int");
  return COATI_OK;
}

int launch_attn_decode(const bf16_t* qkv, bf16_t* cache, bf16_t* y, int B, int n_head, int head_size, int Tmax, int pos,
                       const int* pos_dev, hipStream_t s) {
  COATI_CHECK_ARG(qkv && cache && y, "attn_decode: null operand");
  COATI_CHECK_SHAPE(B > 0 && n_head > 0 && Tmax > 0 && Tmax <= 256 && pos >= 0 && pos < Tmax && (head_size == 16 || head_size == 32),
                    "attn_decode: unsupported shape B=%d nh=%d hs=%d Tmax=%d pos=%d", B, n_head, head_size, Tmax, pos);
  if (head_size == 16)
    hipLaunchKernelGGL(attn_decode_kernel<16>, dim3(cdiv(B * n_head, 4)), dim3(256), 0, s, qkv, cache, y, B, n_head, Tmax, pos, pos_dev);
  else
    hipLaunchKernelGGL(attn_decode_kernel<32>, dim3(cdiv(B * n_head, 4)), dim3(256), 0, s, qkv, cache, y, B, n_head, Tmax, pos, pos_dev);
  COATI_LAUNCH_CHECK("attn_decode");
  return COATI_OK;
}

// ---- top-k sampling (smiles_xformer.py:305-313) -----------------------------------------------------------------------
//   logits_topk, inds = topk(logits[b], k);  probs = softmax(logits_topk * inv_temp);  token = inds[multinomial(probs)]
// One workgroup per row, the row's order-preserving integer keys in LDS.  The k-th largest key is found with a 4-pass
// radix select (8 bits per pass, one histogram bin per thread), the <= TOPK_MAX survivors are compacted (ties at the
// threshold in index order), sorted (value descending, index ascending -- the order of a stable descending sort) and
// sampled by inverse CDF with the caller's uniform u[b] in [0, 1).  stopped rows emit pad_token; a row that draws
// stop_token is marked stopped (reference :314-324).  Integer/compare work on a 40-KB row: ~10 us per row-block.
#define TOPK_MAX 128
__device__ __forceinline__ unsigned f2key(float f) {   // larger float <-> larger unsigned (NaN sorts high, like torch)
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void topk_sample_kernel(const float* __restrict__ logits, long long ldl, int V, int k,
                                                          float inv_temp, const float* __restrict__ u,
                                                          long long* __restrict__ tok_out, int* __restrict__ stopped,
                                                          int stop_token, int pad_token) {
  extern __shared__ unsigned keys[];   // [V]
  __shared__ int hist[256];
  __shared__ unsigned s_prefix;
  __shared__ int s_need, s_ngt, s_neq;
  __shared__ unsigned top_k[TOPK_MAX];
  __shared__ int top_i[TOPK_MAX];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (stopped && stopped[b]) {
    if (tid == 0) tok_out[b] = pad_token;
    return;
  }
  for (int i = tid; i < V; i += 256) keys[i] = f2key(logits[(long long)b * ldl + i]);
  if (tid == 0) { s_prefix = 0u; s_need = k; }
  __syncthreads();
  // radix select of the k-th largest key: after pass p the top 8*(p+1) bits of the threshold are known
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (int i = tid; i < V; i += 256) {
      const unsigned key = keys[i];
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int need = s_need, bin = 255;
      for (; bin > 0; --bin) {
        if (hist[bin] >= need) break;
        need -= hist[bin];
      }
      s_need = need;                              // rank of the threshold inside its bin
      s_prefix = prefix | ((unsigned)bin << shift);
    }
    __syncthreads();
  }
  const unsigned tau = s_prefix;                  // the k-th largest key; s_need = how many keys == tau belong to the top k
  if (tid == 0) { s_ngt = 0; s_neq = 0; }
  __syncthreads();
  // survivors: every key > tau (any order), then the first s_need keys == tau in index order
  for (int i = tid; i < V; i += 256) {
    if (keys[i] > tau) {
      const int slot = atomicAdd(&s_ngt, 1);
      top_k[slot] = keys[i];
      top_i[slot] = i;
    }
  }
  __syncthreads();
  {
    // the first s_need keys == tau in INDEX order: threads own contiguous index segments, exclusive scan of their counts
    const int seg = (V + 255) / 256, i0 = tid * seg, i1 = (i0 + seg < V) ? i0 + seg : V;
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += (keys[i] == tau) ? 1 : 0;
    hist[tid] = cnt;
    __syncthreads();
    int before = 0;
    for (int t = 0; t < tid; ++t) before += hist[t];
    const int base = s_ngt, need = s_need;
    if (cnt > 0 && before < need) {
      int pos = before;
      for (int i = i0; i < i1 && pos < need; ++i)
        if (keys[i] == tau) { top_k[base + pos] = tau; top_i[base + pos] = i; ++pos; }
    }
  }
  __syncthreads();
  // sort the k survivors: key descending, index ascending (rank by counting; k <= 128)
  unsigned myk = 0;
  int myi = 0, rank = 0;
  if (tid < k) {
    myk = top_k[tid];
    myi = top_i[tid];
    for (int j = 0; j < k; ++j) {
      const unsigned kj = top_k[j];
      const int ij = top_i[j];
      rank += (kj > myk || (kj == myk && ij < myi)) ? 1 : 0;
    }
  }
  __syncthreads();
  if (tid < k) { top_k[rank] = myk; top_i[rank] = myi; }
  __syncthreads();
  if (tid == 0) {
    const float mx = key2f(top_k[0]) * inv_temp;
    float z = 0.f;
    for (int r = 0; r < k; ++r) z += __expf(key2f(top_k[r]) * inv_temp - mx);
    const float target = (u ? u[b] : 0.f) * z;
    float c = 0.f;
    int pick = k - 1;
    for (int r = 0; r < k; ++r) {
      c += __expf(key2f(top_k[r]) * inv_temp - mx);
      if (target < c) { pick = r; break; }
    }
    const int tok = top_i[pick];
    tok_out[b] = tok;
    if (stopped && tok == stop_token) stopped[b] = 1;
  }
}

int launch_topk_sample(const float* logits, long long ldl, int B, int V, int k, float inv_temp, const float* u,
                       long long* tok_out, int* stopped, int stop_token, int pad_token, hipStream_t s) {
  COATI_CHECK_ARG(logits && tok_out, "topk_sample: null operand");
  COATI_CHECK_SHAPE(B > 0 && V > 0 && k > 0 && k <= TOPK_MAX && k <= V && (size_t)V * 4 <= 120 * 1024,
                    "topk_sample: unsupported shape B=%d V=%d k=%d", B, V, k);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    if (e != hipSuccess) {
      coati_set_error("topk_sample: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(topk_sample_kernel, dim3(B), dim3(256), (size_t)V * 4, s, logits, ldl, V, k, inv_temp, u, tok_out, stopped, stop_token, pad_token);
  COATI_LAUNCH_CHECK("topk_sample");
  return COATI_OK;
}
