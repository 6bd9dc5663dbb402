// Causal rotary self-attention for head size 16 (grande / "closed": d = 256, 16 heads), sequences of T <= 128 rows, on 16-ROW
// causal granularity -- reference basic_transformer.py:126-154 (`RotarySelfAttention.forward`: att = softmax(mask(q k^T / sqrt(hs))) v).
//
// Round 6.  attention.hip works on 32 x 32 blocks of v_mfma_f32_32x32x16_bf16: for the training batch's lengths (16 .. 80 rows) it
// evaluated 2.4 x the score elements the causal triangle holds (the diagonal block runs all 1024 exponentials with half of them masked,
// a ragged tail is rounded up to 32 rows) and threw half of every PV result away (16 of the 32 accumulator rows are dims that do not
// exist).  Head size 16 is exactly ONE v_mfma_f32_16x16x16_bf16 in K (scores) and in M (the 16 dims of P V): here a sequence is
// ceil(T / 16) blocks, only the nb (nb + 1) / 2 blocks of the causal triangle are ever issued, the mask touches the 16 x 16
// diagonal blocks only, and no accumulator row is dead.
//
//  * one 64-lane wave per (sequence, head), 4 heads = 4 waves per workgroup: the workgroup moves whole 128-B row segments of
//    qkv [rows, 3C], y / dy [rows, C]; every operand is staged ONCE, all global loads of the prologue issued before the first LDS
//    write (attn_img.h), as row-major [T][16] bf16 images with the 16-B-chunk bank swizzle;
//  * forward: S^T = K Q^T per (key block, query block) -- lane = query (l & 15), registers = keys 4 (l >> 4) + r -- so the softmax
//    statistics are lane-local plus two cross-row swaps (v_permlane16_swap, v_permlane32_swap).  A query block's scores against ALL its
//    key blocks (<= 8 x 4 registers) are in registers at once: plain two-pass softmax, no online rescaling.  P^T leaves the
//    accumulators as the B operand of O^T += V^T P^T as it is; V^T comes out of the row-major image with ds_read_b64_tr_b16; two key
//    blocks share one v_mfma_f32_16x16x32_bf16;
//  * backward: ONE sweep over the (query block, key block) pairs gives dQ, dK, dV (P recomputed from the saved log-sum-exp).  The K
//    and V fragments of every key block (both orientations of K) stay in registers for the whole sweep, dK / dV of every key block
//    too; per pair the LDS traffic is the 16 x 16 dS tile written as bf16 and read back transposed (the B operand of dQ^T += K^T dS^T);
//  * results are written over the images they replace (O / dQ over Q's rows, dK over K, dV over dO -- the same 16 x 16 bf16 shape)
//    and leave the workgroup after ONE barrier as whole 128-B row segments.
//  * one workgroup per item, on purpose: a persistent workgroup that prefetches its next item into registers while it computes the current
//    one was built and measured 17 % slower in three variants (profiles/r06_attn_persistent.txt; the kernel is in the git history).
// q and k arrive rotated (the QKV product applies RoPE in its write-out); dq and dk are rotated back here.
// Layout: qkv [B*T, 3C] bf16 (q | k | v, head h at columns 16 h ..), y / dy [B*T, C], lse [B, nh, Tl] (padded pitch in both row layouts).
#include "attn_img.h"

typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef short v8s16 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s16 lds_v4s16;

#define A16_SCALE 0.25f
#define A16_SCALE_LOG2E 0.36067376022224085f

__device__ __forceinline__ f32x4 a16_zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 mfma16(v4s16 a, v4s16 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16x2(v4s16 a0, v4s16 a1, v4s16 b0, v4s16 b1, f32x4 c) {   // two 16-deep reduction blocks in one issue
  const v8s16 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ v4s16 a16_pack(const f32x4& v) {
  const uint2 u = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
  return __builtin_bit_cast(v4s16, u);
}
// max / sum over the four 16-lane rows of the wave (lanes l, l ^ 16, l ^ 32, l ^ 48): the swap instructions exchange rows between their two
// operands; with both operands the same register every lane ends up holding its own value and a partner row's
__device__ __forceinline__ float rows_max(float v) {
  coati_v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float rows_sum(float v) {
  coati_v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// Offsets (in bf16 elements, inside one 16-row block of an image: a block starts on a multiple of 16 rows, which does not change the
// swizzle) of the lane's piece of
//   the plain fragment: row l & 15, dims 4 (l >> 4) .. + 3   (A operand with rows as MFMA rows / B operand with rows as MFMA columns;
//                       also where the lane's 4 results of one row go when a result block overwrites the image)
//   the transpose read: 16-lane group g fetches rows 4 g .. 4 g + 3, lane i of the group supplies the address of row 4 g + (i >> 2),
//                       dims 4 (i & 3) .. + 3 and receives column i of the four rows: X^T[d = i][rows 4 g + r]
__device__ __forceinline__ int a16_foff(int lane) {
  const int r = lane & 15, g = lane >> 4;
  return r * 16 + img_chunk<16>(r, g >> 1) * 8 + (g & 1) * 4;
}
__device__ __forceinline__ int a16_toff(int lane) {
  const int i = lane & 15, r = 4 * (lane >> 4) + (i >> 2);
  return r * 16 + img_chunk<16>(r, (i & 3) >> 1) * 8 + (i & 1) * 4;
}
__device__ __forceinline__ v4s16 a16_frag(const bf16_t* img, int blk, int foff) { return *reinterpret_cast<const v4s16*>(img + blk * 256 + foff); }
__device__ __forceinline__ v4s16 a16_tfrag(const bf16_t* img, int blk, int toff) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16*)(img + blk * 256 + toff));
}

// ---- staging: one operand of a sequence (rows of 4 heads x 16 dims = 128 B) <-> the 4 heads' images -------------------------------
// Task (row t, 16-B chunk c of the 128 B) -> head c >> 1's image, row t, chunk (c & 1) swizzled.  A block count that leaves half a round
// of tasks (16 NB rows x 8 chunks is an odd multiple of 128 for odd NB) takes it with ALL 256 threads on 8-B pieces: every thread runs
// the same compile-time number of loads, all of them issued before the first LDS write -- ONE memory round trip per prologue.  (The
// round-5 helper predicated the last round on the thread index; the compiler sank those loads into the conditional store block behind
// the first s_waitcnt: a second, serialised round trip for every sequence of 1, 3 or 5 blocks.)  Row addresses are 32-bit byte
// offsets from the sequence's first row (a scalar base): rows of a sequence span < 128 x 3 C x 2 B.
template <int TP>
struct A16Stage {
  static constexpr int NF = TP * 8 / 256;
  static constexpr bool HALF = (TP * 8 % 256) != 0;
  uint4 v[NF > 0 ? NF : 1];
  uint2 h;
};
template <int TP>
__device__ __forceinline__ void a16_stage_load(A16Stage<TP>& r, const bf16_t* src, unsigned stride_b, int T, int heads_here, int tid) {
  const char* base = reinterpret_cast<const char*>(src);
#pragma unroll
  for (int i = 0; i < A16Stage<TP>::NF; ++i) {
    const int task = tid + 256 * i, t = task >> 3, c = task & 7;
    const unsigned tc = min(t, T - 1), cc = (c >> 1) < heads_here ? c : 0;
    r.v[i] = *reinterpret_cast<const uint4*>(base + (tc * stride_b + cc * 16));
  }
  if constexpr (A16Stage<TP>::HALF) {
    const int task = A16Stage<TP>::NF * 256 + (tid >> 1), t = task >> 3, c = task & 7;
    const unsigned tc = min(t, T - 1), cc = (c >> 1) < heads_here ? c : 0;
    r.h = *reinterpret_cast<const uint2*>(base + (tc * stride_b + cc * 16 + (tid & 1) * 8));
  }
}
template <int TP>
__device__ __forceinline__ void a16_stage_store(const A16Stage<TP>& r, int T, unsigned char* smem, unsigned pw, int image, int heads_here, int tid) {
#pragma unroll
  for (int i = 0; i < A16Stage<TP>::NF; ++i) {
    const int task = tid + 256 * i, t = task >> 3, c = task & 7, w = c >> 1;
    const unsigned keep = (t < T && w < heads_here) ? 0xffffffffu : 0u;   // mask, not select: the load stays unpredicated
    *reinterpret_cast<uint4*>(smem + w * pw + image * (TP * 32) + t * 32 + img_chunk<16>(t, c & 1) * 16) =
        make_uint4(r.v[i].x & keep, r.v[i].y & keep, r.v[i].z & keep, r.v[i].w & keep);
  }
  if constexpr (A16Stage<TP>::HALF) {
    const int task = A16Stage<TP>::NF * 256 + (tid >> 1), t = task >> 3, c = task & 7, w = c >> 1;
    const unsigned keep = (t < T && w < heads_here) ? 0xffffffffu : 0u;
    *reinterpret_cast<uint2*>(smem + w * pw + image * (TP * 32) + t * 32 + img_chunk<16>(t, c & 1) * 16 + (tid & 1) * 8) = make_uint2(r.h.x & keep, r.h.y & keep);
  }
}
// images -> global: image `image` of the 4 heads' regions as whole 128-B row segments (the inverse of a16_stage_store)
template <int TP>
__device__ __forceinline__ void a16_unstage(const unsigned char* smem, unsigned pw, int image, bf16_t* dst, unsigned stride_b, int T, int heads_here, int tid) {
  char* base = reinterpret_cast<char*>(dst);
#pragma unroll
  for (int i = 0; i < A16Stage<TP>::NF; ++i) {
    const int task = tid + 256 * i, t = task >> 3, c = task & 7, w = c >> 1;
    if (t < T && w < heads_here)
      *reinterpret_cast<uint4*>(base + ((unsigned)t * stride_b + c * 16)) = *reinterpret_cast<const uint4*>(smem + w * pw + image * (TP * 32) + t * 32 + img_chunk<16>(t, c & 1) * 16);
  }
  if constexpr (A16Stage<TP>::HALF) {
    const int task = A16Stage<TP>::NF * 256 + (tid >> 1), t = task >> 3, c = task & 7, w = c >> 1;
    if (t < T && w < heads_here)
      *reinterpret_cast<uint2*>(base + ((unsigned)t * stride_b + c * 16 + (tid & 1) * 8)) =
          *reinterpret_cast<const uint2*>(smem + w * pw + image * (TP * 32) + t * 32 + img_chunk<16>(t, c & 1) * 16 + (tid & 1) * 8);
  }
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// the compute phase of one (sequence, 4 heads) item on its staged images (between the two workgroup barriers of the body)
template <int NB>
__device__ __forceinline__ void att16_fwd_compute(unsigned char* smem, float* __restrict__ lse, int Tl, int n_head, int b, int hq, int T) {
  constexpr int Tp = 16 * NB;
  constexpr unsigned pw = 3 * Tp * 32 + ATT_PW_PAD;   // images: Q (O takes its place block by block), K, V
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = hq * 4 + wave;
  if (hh < n_head) {
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem + wave * pw);
    const bf16_t* Ks = Qs + Tp * 16;
    const bf16_t* Vs = Ks + Tp * 16;
    const int ql = lane & 15, g = lane >> 4;
    const int foff = a16_foff(lane), toff = a16_toff(lane);
    float* lrow = lse + ((long long)b * n_head + hh) * Tl;
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
      const v4s16 qf = a16_frag(Qs, qb, foff);
      f32x4 s[NB];
#pragma unroll
      for (int kb = 0; kb <= qb; ++kb) s[kb] = mfma16(a16_frag(Ks, kb, foff), qf, a16_zero());
      // causal mask: the diagonal block only (keys behind the sequence's end lie in the last block's masked half for every real query)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r > ql) s[qb][r] = -INFINITY;
      float mk[NB];
#pragma unroll
      for (int kb = 0; kb <= qb; ++kb) mk[kb] = fmaxf(fmaxf(s[kb][0], s[kb][1]), fmaxf(s[kb][2], s[kb][3]));
      float m = mk[0];
#pragma unroll
      for (int kb = 1; kb <= qb; ++kb) m = fmaxf(m, mk[kb]);
      m = rows_max(m);
      const float mc = m * A16_SCALE_LOG2E;
      float lk[NB];
      v4s16 pk[NB];
#pragma unroll
      for (int kb = 0; kb <= qb; ++kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], A16_SCALE_LOG2E, -mc));
        lk[kb] = (s[kb][0] + s[kb][1]) + (s[kb][2] + s[kb][3]);
        pk[kb] = a16_pack(s[kb]);
      }
      float l = lk[0];
#pragma unroll
      for (int kb = 1; kb <= qb; ++kb) l += lk[kb];
      l = rows_sum(l);
      f32x4 o = a16_zero();
#pragma unroll
      for (int kb = 0; kb + 1 <= qb; kb += 2) o = mfma16x2(a16_tfrag(Vs, kb, toff), a16_tfrag(Vs, kb + 1, toff), pk[kb], pk[kb + 1], o);
      if ((qb & 1) == 0) o = mfma16(a16_tfrag(Vs, qb, toff), pk[qb], o);
      const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] *= inv;
      // O over Q's rows of this block (read above; same wave, LDS operations of a wave complete in order)
      *reinterpret_cast<v4s16*>(Qs + qb * 256 + foff) = a16_pack(o);
      const int q = qb * 16 + ql;
      if (g == 0 && q < T) lrow[q] = m * A16_SCALE + __logf(l);
    }
  }
}

template <int NB>
__device__ __forceinline__ void att16_fwd_body(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ y, float* __restrict__ lse,
                                               int Tl, int n_head, int b, int hq, int T, long long row0) {
  constexpr int Tp = 16 * NB;
  constexpr unsigned pw = 3 * Tp * 32 + ATT_PW_PAD;   // images: Q (O takes its place block by block), K, V
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = hq * 4 + wave;
  const int heads_here = (n_head - hq * 4) < 4 ? (n_head - hq * 4) : 4;
  const int C = n_head * 16;
  const bf16_t* base = qkv + row0 * (3LL * C) + hq * 64;
  {
    A16Stage<Tp> rq, rk, rv;
    a16_stage_load<Tp>(rq, base, 6 * C, T, heads_here, threadIdx.x);
    a16_stage_load<Tp>(rk, base + C, 6 * C, T, heads_here, threadIdx.x);
    a16_stage_load<Tp>(rv, base + 2 * C, 6 * C, T, heads_here, threadIdx.x);
    a16_stage_store<Tp>(rq, T, smem, pw, 0, heads_here, threadIdx.x);
    a16_stage_store<Tp>(rk, T, smem, pw, 1, heads_here, threadIdx.x);
    a16_stage_store<Tp>(rv, T, smem, pw, 2, heads_here, threadIdx.x);
  }
  __syncthreads();
#ifndef A16_PROBE_NOCOMPUTE
  att16_fwd_compute<NB>(smem, lse, Tl, n_head, b, hq, T);
#endif
  __syncthreads();
#ifdef A16_PROBE_NOSTORE
  if (Tl < 0)
#endif
  a16_unstage<Tp>(smem, pw, 0, y + row0 * C + hq * 64, 2 * C, T, heads_here, threadIdx.x);
}

template <int NBMAX>
__global__ __launch_bounds__(256) void att16_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ y, float* __restrict__ lse,
                                                        int Tl, int n_head, int quads, const int* __restrict__ seq_off, const int* __restrict__ seq_ord) {
  const int bi = blockIdx.x / quads, hq = blockIdx.x - bi * quads;
  const int b = seq_ord != nullptr ? seq_ord[bi] : bi;      // launch order: the long sequences first (embed.hip seq_scan_kernel)
  ATT_SEQ(0);
  const int nb = (T + 15) >> 4;   // uniform over the workgroup: it runs the body compiled for ITS sequence's block count
#define A16_CASE(N)                                                                   \
  if constexpr (NBMAX >= N)                                                           \
    if (nb == N) {                                                                    \
      att16_fwd_body<N>(qkv, y, lse, Tl, n_head, b, hq, T, row0);                     \
      return;                                                                         \
    }
  A16_CASE(1) A16_CASE(2) A16_CASE(3) A16_CASE(4) A16_CASE(5) A16_CASE(6) A16_CASE(7) A16_CASE(8)
#undef A16_CASE
}

size_t att16_fwd_lds(int T) { return (size_t)4 * ((size_t)3 * 16 * cdiv(T, 16) * 32 + ATT_PW_PAD); }

int launch_attn16_fwd(const bf16_t* qkv, bf16_t* y, float* lse, int B, int T, int n_head, hipStream_t s, const int* seq_off, const int* seq_ord) {
  if (seq_off == nullptr) seq_ord = nullptr;   // the padded layout has one length
  COATI_CHECK_SHAPE(T > 0 && T <= 128, "attn16_fwd: T=%d", T);
  const int quads = cdiv(n_head, 4), nb = cdiv(T, 16);
  size_t lds = att16_fwd_lds(T);
#ifdef A16_PROBE_LDS_EXTRA
  lds += A16_PROBE_LDS_EXTRA;
#endif
  if (nb <= 3) hipLaunchKernelGGL((att16_fwd_kernel<3>), dim3(B * quads), dim3(256), lds, s, qkv, y, lse, T, n_head, quads, seq_off, seq_ord);
  else if (nb <= 5) hipLaunchKernelGGL((att16_fwd_kernel<5>), dim3(B * quads), dim3(256), lds, s, qkv, y, lse, T, n_head, quads, seq_off, seq_ord);
  else hipLaunchKernelGGL((att16_fwd_kernel<8>), dim3(B * quads), dim3(256), lds, s, qkv, y, lse, T, n_head, quads, seq_off, seq_ord);
  COATI_LAUNCH_CHECK("attn16_fwd");
  return COATI_OK;
}

// ---------------------------------------------------------------------------------------------------
// backward.  P = exp(S scale - lse), dS = P (dP - D) scale with dP = dO V^T, D[q] = sum_d dO[q, d] O[q, d];
//   dV^T[d][key] = sum_q dO^T[d][q] P[q][key]   dK^T[d][key] = sum_q Q^T[d][q] dS[q][key]   dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
// Pairs are evaluated with lane = key (l & 15), registers = queries 4 (l >> 4) + r: P and dS are then the B operands of the dV and dK
// products as they leave the accumulators; dQ wants dS^T -- through the wave's 16 x 16 LDS tile.
// ---------------------------------------------------------------------------------------------------
#define A16_DST_PITCH 20   // bf16 per row of the dS tile (40 B)
template <int TP> __host__ __device__ constexpr size_t a16_bwd_pw() { return (size_t)3 * TP * 32 + (size_t)2 * TP * 4 + 2 * 16 * A16_DST_PITCH * 2 + ATT_PW_PAD; }

// inverse rotation of a gradient block held as (row l & 15, dims 4 g + r): the RoPE pair of dim d < 8 is d + 8 = the same register of
// lane l ^ 32; cos / sin tables [T][16] hold the 8 frequencies twice (cat(freqs, freqs), basic_transformer.py:57-69)
__device__ __forceinline__ f32x4 a16_rope_inv(const f32x4& v, const float* cos_t, const float* sin_t, int t, int lane) {
  const int g = lane >> 4;
  const float4 cs = *reinterpret_cast<const float4*>(cos_t + t * 16 + 4 * g);
  const float4 sn = *reinterpret_cast<const float4*>(sin_t + t * 16 + 4 * g);
  const float csv[4] = {cs.x, cs.y, cs.z, cs.w}, snv[4] = {sn.x, sn.y, sn.z, sn.w};
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const coati_v2u x = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[r]), __float_as_uint(v[r]), false, false);
    const float partner = __uint_as_float(lane < 32 ? x.y : x.x);
    // d < 8: g_d cos + g_{d+8} sin ; d >= 8: g_d cos - g_{d-8} sin
    o[r] = fmaf(v[r], csv[r], (g < 2 ? partner : -partner) * snv[r]);
  }
  return o;
}

__device__ __forceinline__ float a16_dot4(const uint2& a, const uint2& b) {   // 4 bf16 pairs, one fixed order
  return fmaf(bfhi(a.y), bfhi(b.y), fmaf(bflo(a.y), bflo(b.y), fmaf(bfhi(a.x), bfhi(b.x), bflo(a.x) * bflo(b.x))));
}

template <int NB>
__device__ __forceinline__ void att16_bwd_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy,
                                               const float* __restrict__ lse, bf16_t* __restrict__ dqkv, const float* __restrict__ cos_t,
                                               const float* __restrict__ sin_t, int Tl, int n_head, int b, int hq, int T, long long row0) {
  constexpr int Tp = 16 * NB;
  constexpr unsigned pw = (unsigned)a16_bwd_pw<Tp>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = hq * 4 + wave;
  const int heads_here = (n_head - hq * 4) < 4 ? (n_head - hq * 4) : 4;
  const int C = n_head * 16;
  const long long stride = 3LL * C;
  const bool active = hh < n_head;
  const int hc = active ? hh : 0;
  const int kl = lane & 15, g = lane >> 4;
  const bf16_t* qbase = qkv + row0 * stride + hq * 64;
  unsigned char* my = smem + (size_t)wave * pw;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(my);
  bf16_t* Ks = Qs + Tp * 16;
  bf16_t* Gs = Ks + Tp * 16;
  float* Ls = reinterpret_cast<float*>(Gs + Tp * 16);
  float* Ds = Ls + Tp;
  bf16_t* dsT = reinterpret_cast<bf16_t*>(Ds + Tp);
  v4s16 vf[NB];
  {
    // one memory round trip: the three staged operands, O (for D), this head's log-sum-exp rows and V's fragments
    A16Stage<Tp> rq, rk, rg, ro;
    a16_stage_load<Tp>(rq, qbase, 6 * C, T, heads_here, threadIdx.x);
    a16_stage_load<Tp>(rk, qbase + C, 6 * C, T, heads_here, threadIdx.x);
    a16_stage_load<Tp>(rg, dy + row0 * C + hq * 64, 2 * C, T, heads_here, threadIdx.x);
    a16_stage_load<Tp>(ro, y + row0 * C + hq * 64, 2 * C, T, heads_here, threadIdx.x);
    float lrow[(Tp + 63) / 64];
#pragma unroll
    for (int i = 0; i < (Tp + 63) / 64; ++i) {
      const int t = lane + 64 * i;
      lrow[i] = lse[((long long)b * n_head + hc) * Tl + min(t, T - 1)];
    }
    const char* vsrc = reinterpret_cast<const char*>(qkv + row0 * stride + 2 * C + hc * 16);
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const int row = kb * 16 + kl;
      uint2 u = *reinterpret_cast<const uint2*>(vsrc + ((unsigned)min(row, T - 1) * (unsigned)(6 * C) + 8 * g));
      if (row >= T) u = make_uint2(0, 0);
      vf[kb] = __builtin_bit_cast(v4s16, u);
    }
    a16_stage_store<Tp>(rq, T, smem, pw, 0, heads_here, threadIdx.x);
    a16_stage_store<Tp>(rk, T, smem, pw, 1, heads_here, threadIdx.x);
    a16_stage_store<Tp>(rg, T, smem, pw, 2, heads_here, threadIdx.x);
    {  // D[t] = sum_d dO[t, d] O[t, d]: the thread that staged a piece of dO holds the same piece of O; the lanes of a head's row add up
      float* const d0 = reinterpret_cast<float*>(smem + 3 * Tp * 32) + Tp;   // head 0's Ds
#pragma unroll
      for (int i = 0; i < A16Stage<Tp>::NF; ++i) {
        const int task = threadIdx.x + 256 * i, t = task >> 3, c = task & 7, w = c >> 1;
        // (a row's D must not depend on which round staged it: the same 4-term sums in the same order in both forms)
        float d = a16_dot4(make_uint2(rg.v[i].x, rg.v[i].y), make_uint2(ro.v[i].x, ro.v[i].y)) + a16_dot4(make_uint2(rg.v[i].z, rg.v[i].w), make_uint2(ro.v[i].z, ro.v[i].w));
        d = (t < T && w < heads_here) ? d : 0.f;
        d += __shfl_xor(d, 1, 64);
        if ((c & 1) == 0) *reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(d0) + w * pw + 4 * t) = d;
      }
      if constexpr (A16Stage<Tp>::HALF) {
        const int task = A16Stage<Tp>::NF * 256 + (threadIdx.x >> 1), t = task >> 3, c = task & 7, w = c >> 1;
        float d = a16_dot4(rg.h, ro.h);
        d = (t < T && w < heads_here) ? d : 0.f;
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        if ((c & 1) == 0 && (threadIdx.x & 1) == 0) *reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(d0) + w * pw + 4 * t) = d;
      }
    }
#pragma unroll
    for (int i = 0; i < (Tp + 63) / 64; ++i) {   // log-sum-exp pre-multiplied by log2(e); rows behind the end: +inf -> P = 0
      const int t = lane + 64 * i;
      if (t < Tp) Ls[t] = (t < T) ? lrow[i] * LOG2E : INFINITY;
    }
  }
  __syncthreads();
#ifndef A16_PROBE_NOCOMPUTE
  if (active) {
    const int foff = a16_foff(lane), toff = a16_toff(lane);
    f32x4 dk[NB], dv[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) dk[kb] = dv[kb] = a16_zero();
    // the dS tile: two of them, written alternately; the dQ product of a pair is issued one pair later, so that the tile's write -> transposed
    // read round trip hides behind the next pair's products and exponentials.  (LDS operations of ONE wave complete in issue order: the
    // read needs no wait on the write.)
    bf16_t* const tw = dsT + kl * A16_DST_PITCH + 4 * g;                                  // this lane's write: row key, queries 4 g ..
    const bf16_t* const tr = dsT + (4 * g + (kl >> 2)) * A16_DST_PITCH + 4 * (kl & 3);    // transpose read: rows (keys) 4 g .., column q = kl
    constexpr int TILE = 16 * A16_DST_PITCH;
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
      const v4s16 qf = a16_frag(Qs, qb, foff), gf = a16_frag(Gs, qb, foff);
      const v4s16 qT = a16_tfrag(Qs, qb, toff), gT = a16_tfrag(Gs, qb, toff);
      const float4 l4 = *reinterpret_cast<const float4*>(Ls + qb * 16 + 4 * g);
      const float4 d4 = *reinterpret_cast<const float4*>(Ds + qb * 16 + 4 * g);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
      f32x4 dq = a16_zero();
#pragma unroll
      for (int kb = 0; kb <= qb; ++kb) {
        const f32x4 s = mfma16(qf, a16_frag(Ks, kb, foff), a16_zero());    // S[q = 4 g + r][key = kl]
        const f32x4 dp = mfma16(gf, vf[kb], a16_zero());
        if (kb > 0) dq = mfma16(a16_tfrag(Ks, kb - 1, toff), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16*)(tr + ((kb - 1) & 1) * TILE)), dq);
        f32x4 p, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], A16_SCALE_LOG2E, -lv[r]));
          if (kb == qb && kl > 4 * g + r) p[r] = 0.f;
          ds[r] = p[r] * (dp[r] - dd[r]);   // the softmax scale is applied once to the finished dQ / dK blocks
        }
        const v4s16 pp = a16_pack(p), pd = a16_pack(ds);
        *reinterpret_cast<v4s16*>(tw + (kb & 1) * TILE) = pd;
        dv[kb] = mfma16(gT, pp, dv[kb]);
        dk[kb] = mfma16(qT, pd, dk[kb]);
      }
      dq = mfma16(a16_tfrag(Ks, qb, toff), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16*)(tr + (qb & 1) * TILE)), dq);
      const int q = qb * 16 + kl;
#pragma unroll
      for (int r = 0; r < 4; ++r) dq[r] *= A16_SCALE;
      *reinterpret_cast<v4s16*>(Qs + qb * 256 + foff) = a16_pack(a16_rope_inv(dq, cos_t, sin_t, q < T ? q : 0, lane));
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const int key = kb * 16 + kl;
#pragma unroll
      for (int r = 0; r < 4; ++r) dk[kb][r] *= A16_SCALE;
      *reinterpret_cast<v4s16*>(Ks + kb * 256 + foff) = a16_pack(a16_rope_inv(dk[kb], cos_t, sin_t, key < T ? key : 0, lane));
      *reinterpret_cast<v4s16*>(Gs + kb * 256 + foff) = a16_pack(dv[kb]);
    }
  }
#endif
  __syncthreads();
#ifdef A16_PROBE_NOSTORE
  if (Tl < 0) {
#endif
  bf16_t* const dbase = dqkv + row0 * stride + hq * 64;
  a16_unstage<Tp>(smem, pw, 0, dbase, 6 * C, T, heads_here, threadIdx.x);
  a16_unstage<Tp>(smem, pw, 1, dbase + C, 6 * C, T, heads_here, threadIdx.x);
  a16_unstage<Tp>(smem, pw, 2, dbase + 2 * C, 6 * C, T, heads_here, threadIdx.x);
#ifdef A16_PROBE_NOSTORE
  }
#endif
}

template <int NBMAX>
__global__ __launch_bounds__(256, NBMAX <= 5 ? 4 : 2) void att16_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy,
                                                        const float* __restrict__ lse, bf16_t* __restrict__ dqkv, const float* __restrict__ cos_t,
                                                        const float* __restrict__ sin_t, int Tl, int n_head, int quads, const int* __restrict__ seq_off, const int* __restrict__ seq_ord) {
  const int bi = blockIdx.x / quads, hq = blockIdx.x - bi * quads;
  const int b = seq_ord != nullptr ? seq_ord[bi] : bi;
  ATT_SEQ(0);
  const int nb = (T + 15) >> 4;
#define A16_CASE(N)                                                                                  \
  if constexpr (NBMAX >= N)                                                                          \
    if (nb == N) {                                                                                   \
      att16_bwd_body<N>(qkv, y, dy, lse, dqkv, cos_t, sin_t, Tl, n_head, b, hq, T, row0);            \
      return;                                                                                        \
    }
  A16_CASE(1) A16_CASE(2) A16_CASE(3) A16_CASE(4) A16_CASE(5) A16_CASE(6) A16_CASE(7) A16_CASE(8)
#undef A16_CASE
}

int launch_attn16_bwd(const bf16_t* qkv, const bf16_t* y, const bf16_t* dy, const float* lse, bf16_t* dqkv, const float* cos_t,
                      const float* sin_t, int B, int T, int n_head, hipStream_t s, const int* seq_off, const int* seq_ord) {
  COATI_CHECK_SHAPE(T > 0 && T <= 128, "attn16_bwd: T=%d", T);
  if (seq_off == nullptr) seq_ord = nullptr;
  const int quads = cdiv(n_head, 4), nb = cdiv(T, 16);
  size_t lds = (size_t)4 * ((size_t)3 * 16 * nb * 32 + (size_t)2 * 16 * nb * 4 + 2 * 16 * A16_DST_PITCH * 2 + ATT_PW_PAD);
#ifdef A16_PROBE_LDS_EXTRA
  lds += A16_PROBE_LDS_EXTRA;
#endif
  if (nb <= 3) hipLaunchKernelGGL((att16_bwd_kernel<3>), dim3(B * quads), dim3(256), lds, s, qkv, y, dy, lse, dqkv, cos_t, sin_t, T, n_head, quads, seq_off, seq_ord);
  else if (nb <= 5) hipLaunchKernelGGL((att16_bwd_kernel<5>), dim3(B * quads), dim3(256), lds, s, qkv, y, dy, lse, dqkv, cos_t, sin_t, T, n_head, quads, seq_off, seq_ord);
  else hipLaunchKernelGGL((att16_bwd_kernel<8>), dim3(B * quads), dim3(256), lds, s, qkv, y, dy, lse, dqkv, cos_t, sin_t, T, n_head, quads, seq_off, seq_ord);
  COATI_LAUNCH_CHECK("attn16_bwd");
  return COATI_OK;
}
