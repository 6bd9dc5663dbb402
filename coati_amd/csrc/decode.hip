// Inference decode step: KV-cached causal attention for ONE new token per sequence at
// head size 16, and top-k sampling of the next token (reference smiles_xformer.py:272-351 recomputes the whole prefix
// for every generated token; here the rotated keys and the values of earlier positions live in an HBM cache).
//
// Cache layout: [B][n_head][Tmax][k16 | v16] bf16 -> one (b, head) sequence is a contiguous run of 64-B records, a wave
// streams it with one 64-B record per lane per pass.  HBM-bound: 64 B per cached token per head per step.
#include "kernels.h"

#define DHS 16

__device__ __forceinline__ void load_bf16x16(const bf16_t* p, float* x) {
  const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
  unpack8(a, x);
  unpack8(b, x + 8);
}

// qkv: [B, 3C] bf16 of the new token (q, k already rotated by the QKV GEMM epilogue); y: [B, C] bf16.
// One wave per (b, head).  Appends (k, v) at position pos, attends to positions 0..pos.
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ cache,
                                                          bf16_t* __restrict__ y, int B, int n_head, int Tmax, int pos) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= B * n_head) return;
  const int b = item / n_head, h = item - b * n_head;
  const int C = n_head * DHS;
  const bf16_t* row = qkv + (long long)b * 3 * C + h * DHS;
  float q[DHS], kn[DHS], vn[DHS];
  load_bf16x16(row, q);
  load_bf16x16(row + C, kn);
  load_bf16x16(row + 2 * C, vn);
  bf16_t* seq = cache + ((long long)item * Tmax) * 32;
  if (lane < 4) {   // append the new record: 4 x 16 B
    const bf16_t* src = (lane < 2) ? row + C + lane * 8 : row + 2 * C + (lane - 2) * 8;
    *reinterpret_cast<uint4*>(seq + (long long)pos * 32 + lane * 8) = *reinterpret_cast<const uint4*>(src);
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
        load_bf16x16(seq + (long long)t * 32, k);
        load_bf16x16(seq + (long long)t * 32 + 16, kv[i]);
      }
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DHS; ++d) s += q[d] * k[d];
      sc[i] = s * 0.25f;   // 1 / sqrt(16)
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
    uint4 o0, o1;
    o0.x = pack2bf(acc[0], acc[1]); o0.y = pack2bf(acc[2], acc[3]); o0.z = pack2bf(acc[4], acc[5]); o0.w = pack2bf(acc[6], acc[7]);
    o1.x = pack2bf(acc[8], acc[9]); o1.y = pack2bf(acc[10], acc[11]); o1.z = pack2bf(acc[12], acc[13]); o1.w = pack2bf(acc[14], acc[15]);
    bf16_t* dst = y + (long long)b * C + h * DHS;
    *reinterpret_cast<uint4*>(dst) = o0;
    *reinterpret_cast<uint4*>(dst + 8) = o1;
  }
}

int launch_attn_decode(const bf16_t* qkv, bf16_t* cache, bf16_t* y, int B, int n_head, int Tmax, int pos, hipStream_t s) {
  COATI_CHECK_ARG(qkv && cache && y, "attn_decode: null operand");
  COATI_CHECK_SHAPE(B > 0 && n_head > 0 && Tmax > 0 && Tmax <= 256 && pos >= 0 && pos < Tmax,
                    "attn_decode: unsupported shape B=%d nh=%d Tmax=%d pos=%d", B, n_head, Tmax, pos);
  hipLaunchKernelGGL(attn_decode_kernel, dim3(cdiv(B * n_head, 4)), dim3(256), 0, s, qkv, cache, y, B, n_head, Tmax, pos);
  COATI_LAUNCH_CHECK("attn_decode");
  return COATI_OK;
}

// ---- top-k sampling (smiles_xformer.py:305-313) -----------------------------------------------------------------------
//   logits_topk, inds = topk(logits[b], k);  probs = softmax(logits_topk * inv_temp);  token = inds[multinomial(probs)]
// One workgroup per row; the row lives in LDS; k rounds of a block-wide arg-max (ties -> the lower index, like a stable
// descending sort), then an inverse-CDF draw with the caller's uniform u[b] in [0, 1).  stopped rows emit pad_token;
// a row that draws stop_token is marked stopped (reference :314-324).
#define TOPK_MAX 128
__global__ __launch_bounds__(256) void topk_sample_kernel(const float* __restrict__ logits, long long ldl, int V, int k,
                                                          float inv_temp, const float* __restrict__ u,
                                                          long long* __restrict__ tok_out, int* __restrict__ stopped,
                                                          int stop_token, int pad_token) {
  extern __shared__ float row[];   // [V]
  __shared__ float wv[4];
  __shared__ int wi[4];
  __shared__ float top_v[TOPK_MAX];
  __shared__ int top_i[TOPK_MAX];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (stopped && stopped[b]) {
    if (tid == 0) tok_out[b] = pad_token;
    return;
  }
  for (int i = tid; i < V; i += 256) row[i] = logits[(long long)b * ldl + i];
  __syncthreads();
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
      const float v = row[i];
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { wv[wave] = bv; wi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float v = wv[0];
      int i = wi[0];
      for (int w = 1; w < 4; ++w)
        if (wv[w] > v || (wv[w] == v && wi[w] < i)) { v = wv[w]; i = wi[w]; }
      top_v[r] = v;
      top_i[r] = i;
      if (i >= 0 && i < V) row[i] = -INFINITY;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float mx = top_v[0] * inv_temp;
    float z = 0.f;
    for (int r = 0; r < k; ++r) z += __expf(top_v[r] * inv_temp - mx);
    const float target = (u ? u[b] : 0.f) * z;
    float c = 0.f;
    int pick = k - 1;
    for (int r = 0; r < k; ++r) {
      c += __expf(top_v[r] * inv_temp - mx);
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
