// Causal rotary self-attention for head size 16 (reference basic_transformer.py:83-100, 126-154),
// forward and backward, one 64-lane wave per (batch row, head).
//
// Head size 16 = exactly one K step of v_mfma_f32_32x32x16_bf16, so the kernels are softmax / LDS bound,
// not MFMA bound.  Everything a (b, head) problem needs (T <= 256 tokens x 16 dims) sits in LDS:
//   * q, k are rotated (RoPE, fp32 maths) while being staged and rounded to bf16 once;
//   * scores are computed TRANSPOSED (S^T = K Q^T) so each lane owns one query column and the softmax
//     row statistics are lane-local (+ one cross-half shuffle);
//   * P^T leaves the MFMA accumulator in exactly the layout the next MFMA wants as its B operand, provided
//     the A operand (V^T, K^T, ...) is read with the same key permutation -- no cross-lane traffic for P;
//   * the [T,T] score matrix is never materialised; the backward recomputes P from the saved log-sum-exp.
// Layout: qkv [B*T, 3C] bf16 (q | k | v, head h at columns h*16..h*16+15), y / dy [B*T, C], lse [B, nh, T].
#include "kernels.h"

#define HS 16
#define SCALE 0.25f   // 1/sqrt(16)

__device__ __forceinline__ void load16(const bf16_t* p, float* x) {
  const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
  unpack8(a, x);
  unpack8(b, x + 8);
}
__device__ __forceinline__ void rope16(float* x, const float* cs, const float* sn) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a = x[i], b = x[i + 8];
    x[i] = a * cs[i] - b * sn[i];
    x[i + 8] = b * cs[i] + a * sn[i];
  }
}
__device__ __forceinline__ void load_cs(const float* tab, int t, float* c) {
  const float4 a = *reinterpret_cast<const float4*>(tab + t * HS), b = *reinterpret_cast<const float4*>(tab + t * HS + 4);
  c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
}

// stage one [T,16] operand: optional RoPE, row-major image (pitch 16) and/or transposed image (pitch tp)
template <bool ROPE>
__device__ __forceinline__ void stage16(const bf16_t* src, long long stride, int T, int Tp, const float* cos_t,
                                        const float* sin_t, bf16_t* rm, bf16_t* tr, int tp, int lane) {
  for (int t = lane; t < Tp; t += 64) {
    float x[16];
    if (t < T) {
      load16(src + (long long)t * stride, x);
      if (ROPE) {
        float cs[8], sn[8];
        load_cs(cos_t, t, cs);
        load_cs(sin_t, t, sn);
        rope16(x, cs, sn);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = 0.f;
    }
    const uint4 lo = pack8(x), hi = pack8(x + 8);
    if (rm) {
      *reinterpret_cast<uint4*>(rm + t * HS) = lo;
      *reinterpret_cast<uint4*>(rm + t * HS + 8) = hi;
    }
    if (tr) {
      const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        tr[(2 * i) * tp + t] = (bf16_t)(w[i] & 0xffffu);
        tr[(2 * i + 1) * tp + t] = (bf16_t)(w[i] >> 16);
      }
    }
  }
}

// A/B fragment of a row-major [*,16] image: lane (r = lane&31, half = lane>>5) -> row blk*32+r, dims half*8..+7
__device__ __forceinline__ bf16x8 rfrag(const bf16_t* rm, int blk, int lane) {
  return *reinterpret_cast<const bf16x8*>(rm + (blk * 32 + (lane & 31)) * HS + (lane >> 5) * 8);
}
// A fragment of a transposed [16, tp] image with the accumulator's key permutation:
// lane (d = lane&31, h = lane>>5), slot j <-> column base + 4h + (j&3) + 8*(j>>2); rows d >= 16 are zero.
__device__ __forceinline__ bf16x8 tfrag(const bf16_t* tr, int tp, int base, int lane) {
  const int d = lane & 31, h = lane >> 5;
  const bf16_t* p = tr + (d & 15) * tp + base + 4 * h;
  uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 8);
  if (d >= 16) { lo = make_uint2(0, 0); hi = make_uint2(0, 0); }
  const uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ bf16x8 pfrag(const float* p) {  // 8 accumulator values -> bf16 B fragment
  const uint4 u = pack8(p);
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
__device__ __forceinline__ int arow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ y,
                                                      float* __restrict__ lse, const float* __restrict__ cos_t,
                                                      const float* __restrict__ sin_t, int T, int n_head) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x / n_head, hh = blockIdx.x - b * n_head;
  const int C = n_head * HS, Tp = (T + 31) & ~31, tp = Tp + 8;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Ks = Qs + Tp * HS;
  bf16_t* Vt = Ks + Tp * HS;
  const long long stride = 3LL * C;
  const bf16_t* base = qkv + (long long)b * T * stride + hh * HS;
  stage16<true>(base, stride, T, Tp, cos_t, sin_t, Qs, nullptr, 0, lane);
  stage16<true>(base + C, stride, T, Tp, cos_t, sin_t, Ks, nullptr, 0, lane);
  stage16<false>(base + 2 * C, stride, T, Tp, nullptr, nullptr, nullptr, Vt, tp, lane);
  __syncthreads();

  const int nblk = Tp >> 5, half = lane >> 5;
  for (int qb = 0; qb < nblk; ++qb) {
    const bf16x8 qf = rfrag(Qs, qb, lane);
    const int q = qb * 32 + (lane & 31);
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o = zero16();
    for (int kb = 0; kb <= qb; ++kb) {
      f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(Ks, kb, lane), qf, zero16(), 0, 0, 0);
      float p[16];
      float mloc = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + arow(r, lane);
        p[r] = (key <= q) ? s[r] * SCALE : -INFINITY;
        mloc = fmaxf(mloc, p[r]);
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __expf(m_run - m_new);
      float lsum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __expf(p[r] - m_new);
        lsum += p[r];
      }
      lsum += __shfl_xor(lsum, 32, 64);
      l_run = l_run * alpha + lsum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= alpha;
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Vt, tp, kb * 32, lane), pfrag(p), o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Vt, tp, kb * 32 + 16, lane), pfrag(p + 8), o, 0, 0, 0);
    }
    if (q < T) {
      const float inv = 1.0f / l_run;
      bf16_t* yr = y + ((long long)b * T + q) * C + hh * HS + 4 * half;
      *reinterpret_cast<uint2*>(yr) = make_uint2(pack2bf(o[0] * inv, o[1] * inv), pack2bf(o[2] * inv, o[3] * inv));
      *reinterpret_cast<uint2*>(yr + 8) = make_uint2(pack2bf(o[4] * inv, o[5] * inv), pack2bf(o[6] * inv, o[7] * inv));
      if (half == 0) lse[((long long)b * n_head + hh) * T + q] = m_run + __logf(l_run);
    }
  }
}

int launch_attn_fwd(const bf16_t* qkv, bf16_t* y, float* lse, const float* cos_t, const float* sin_t, int B, int T,
                    int n_head, hipStream_t s) {
  COATI_CHECK_ARG(qkv && y && lse && cos_t && sin_t, "attn_fwd: null operand");
  COATI_CHECK_SHAPE(B > 0 && T > 0 && T <= 256 && n_head > 0, "attn_fwd: unsupported shape B=%d T=%d nh=%d", B, T, n_head);
  const int Tp = (T + 31) & ~31;
  const size_t lds = (size_t)(2 * Tp * HS + HS * (Tp + 8)) * 2;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * n_head), dim3(64), lds, s, qkv, y, lse, cos_t, sin_t, T, n_head);
  COATI_LAUNCH_CHECK("attn_fwd");
  return COATI_OK;
}

// ---------------------------------------------------------------------------------------------------
// backward.  Sweep 1 (per query block): dQ.  Sweep 2 (per key block): dK, dV.  Both recompute P from lse.
// dS = P * (dP - D) * scale with D[q] = sum_d dO[q,d] O[q,d].  No atomics: fully deterministic.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_grad_cols(bf16_t* dst, const f32x16& g, int half, bool rope_inv, const float* cos_t,
                                                const float* sin_t, int t) {
  // lane holds dims d = 4*half + j (regs 0..3) and 8 + 4*half + j (regs 4..7): the RoPE pairs (d, d+8)
  float a[4], c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = g[j]; c[j] = g[4 + j]; }
  if (rope_inv) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float cs = cos_t[t * HS + 4 * half + j], sn = sin_t[t * HS + 4 * half + j];
      const float ga = a[j], gc = c[j];
      a[j] = ga * cs + gc * sn;
      c[j] = gc * cs - ga * sn;
    }
  }
  *reinterpret_cast<uint2*>(dst + 4 * half) = make_uint2(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]));
  *reinterpret_cast<uint2*>(dst + 8 + 4 * half) = make_uint2(pack2bf(c[0], c[1]), pack2bf(c[2], c[3]));
}

__global__ __launch_bounds__(64) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ y,
                                                      const bf16_t* __restrict__ dy, const float* __restrict__ lse,
                                                      bf16_t* __restrict__ dqkv, const float* __restrict__ cos_t,
                                                      const float* __restrict__ sin_t, int T, int n_head) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x / n_head, hh = blockIdx.x - b * n_head;
  const int C = n_head * HS, Tp = (T + 31) & ~31, tp = Tp + 8;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Ks = Qs + Tp * HS;
  bf16_t* Vs = Ks + Tp * HS;
  bf16_t* Gs = Vs + Tp * HS;          // dO row-major
  bf16_t* Qt = Gs + Tp * HS;
  bf16_t* Kt = Qt + HS * tp;
  bf16_t* Gt = Kt + HS * tp;          // dO transposed
  float* Ls = reinterpret_cast<float*>(Gt + HS * tp);
  float* Ds = Ls + Tp;
  const long long stride = 3LL * C;
  const bf16_t* base = qkv + (long long)b * T * stride + hh * HS;
  const bf16_t* ybase = y + (long long)b * T * C + hh * HS;
  const bf16_t* gbase = dy + (long long)b * T * C + hh * HS;
  stage16<true>(base, stride, T, Tp, cos_t, sin_t, Qs, Qt, tp, lane);
  stage16<true>(base + C, stride, T, Tp, cos_t, sin_t, Ks, Kt, tp, lane);
  stage16<false>(base + 2 * C, stride, T, Tp, nullptr, nullptr, Vs, nullptr, 0, lane);
  stage16<false>(gbase, (long long)C, T, Tp, nullptr, nullptr, Gs, Gt, tp, lane);
  for (int t = lane; t < Tp; t += 64) {
    float d = 0.f, l = INFINITY;
    if (t < T) {
      float o[16], g[16];
      load16(ybase + (long long)t * C, o);
      load16(gbase + (long long)t * C, g);
#pragma unroll
      for (int i = 0; i < 16; ++i) d += o[i] * g[i];
      l = lse[((long long)b * n_head + hh) * T + t];
    }
    Ds[t] = d;
    Ls[t] = l;
  }
  __syncthreads();

  const int nblk = Tp >> 5, half = lane >> 5;
  bf16_t* const dbase = dqkv + (long long)b * T * stride + hh * HS;

  // ---- sweep 1: dQ^T[d][q] = sum_keys K^T[d][key] dS^T[key][q] -------------------------------------
  for (int qb = 0; qb < nblk; ++qb) {
    const bf16x8 qf = rfrag(Qs, qb, lane), gf = rfrag(Gs, qb, lane);
    const int q = qb * 32 + (lane & 31);
    const float lq = Ls[q], dq_ = Ds[q];
    f32x16 acc = zero16();
    for (int kb = 0; kb <= qb; ++kb) {
      const f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(Ks, kb, lane), qf, zero16(), 0, 0, 0);
      const f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(Vs, kb, lane), gf, zero16(), 0, 0, 0);
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + arow(r, lane);
        const float p = (key <= q) ? __expf(s[r] * SCALE - lq) : 0.f;
        ds[r] = p * (dp[r] - dq_) * SCALE;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Kt, tp, kb * 32, lane), pfrag(ds), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Kt, tp, kb * 32 + 16, lane), pfrag(ds + 8), acc, 0, 0, 0);
    }
    if (q < T) store_grad_cols(dbase + (long long)q * stride, acc, half, true, cos_t, sin_t, q);
  }

  // ---- sweep 2: dK^T[d][key] = sum_q Q^T[d][q] dS[q][key],  dV^T[d][key] = sum_q dO^T[d][q] P[q][key] ---
  for (int kb = 0; kb < nblk; ++kb) {
    const bf16x8 kf = rfrag(Ks, kb, lane), vf = rfrag(Vs, kb, lane);
    const int key = kb * 32 + (lane & 31);
    f32x16 dk = zero16(), dv = zero16();
    for (int qb = kb; qb < nblk; ++qb) {
      const f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(Qs, qb, lane), kf, zero16(), 0, 0, 0);
      const f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag(Gs, qb, lane), vf, zero16(), 0, 0, 0);
      float p[16], ds[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int q0 = qb * 32 + 8 * g4 + 4 * half;
        const float4 l4 = *reinterpret_cast<const float4*>(Ls + q0);
        const float4 d4 = *reinterpret_cast<const float4*>(Ds + q0);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = g4 * 4 + j, q = q0 + j;
          p[r] = (key <= q) ? __expf(s[r] * SCALE - lv[j]) : 0.f;
          ds[r] = p[r] * (dp[r] - dvv[j]) * SCALE;
        }
      }
      dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Gt, tp, qb * 32, lane), pfrag(p), dv, 0, 0, 0);
      dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Gt, tp, qb * 32 + 16, lane), pfrag(p + 8), dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Qt, tp, qb * 32, lane), pfrag(ds), dk, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag(Qt, tp, qb * 32 + 16, lane), pfrag(ds + 8), dk, 0, 0, 0);
    }
    if (key < T) {
      store_grad_cols(dbase + (long long)key * stride + C, dk, half, true, cos_t, sin_t, key);
      store_grad_cols(dbase + (long long)key * stride + 2 * C, dv, half, false, cos_t, sin_t, key);
    }
  }
}

int launch_attn_bwd(const bf16_t* qkv, const bf16_t* y, const bf16_t* dy, const float* lse, bf16_t* dqkv,
                    const float* cos_t, const float* sin_t, int B, int T, int n_head, hipStream_t s) {
  COATI_CHECK_ARG(qkv && y && dy && lse && dqkv && cos_t && sin_t, "attn_bwd: null operand");
  COATI_CHECK_SHAPE(B > 0 && T > 0 && T <= 256 && n_head > 0, "attn_bwd: unsupported shape B=%d T=%d nh=%d", B, T, n_head);
  const int Tp = (T + 31) & ~31;
  const size_t lds = (size_t)(4 * Tp * HS + 3 * HS * (Tp + 8)) * 2 + (size_t)2 * Tp * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) {
      coati_set_error("attn_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * n_head), dim3(64), lds, s, qkv, y, dy, lse, dqkv, cos_t, sin_t, T, n_head);
  COATI_LAUNCH_CHECK("attn_bwd");
  return COATI_OK;
}
