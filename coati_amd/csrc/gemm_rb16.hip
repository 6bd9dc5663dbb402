// Row-block GEMM on 16-row slabs (round 3): the K = 256 products of the transformer at PACKED batch sizes.
//
// Why: the 32-row form (gemm_rb.hip) holds 168 VGPRs per wave, i.e. 2-3 waves per SIMD, and its phases are latency chains of ONE
// wave -- 32 MFMAs, then ~500 dependent-ish epilogue instructions per tile (tools/probes/rb_trace.py: the NewGELU epilogue costs
// the same 5 k cycles per tile with 2 waves per SIMD as with 3).  A wave's own MFMAs and VALU work do not overlap, other waves'
// do (tools/probes/overlap_probe.hip): the kernel lacks waves, not issue slots.  Here a wave owns 16 rows on
// v_mfma_f32_16x16x32_bf16: half the A slab (32 VGPRs), a quarter of the accumulators (16), half the epilogue per tile -- 128
// VGPRs, 9-16 waves per workgroup, up to 4 waves per SIMD.  A packed batch brings ~200 rows per CU = 13 slabs: one workgroup per CU,
// one round.  Same structure otherwise: A-stationary slab in registers (LayerNorm evaluated inside the slab load for the QKV /
// FC1 products, basic_transformer.py:165-173), 64-column weight tiles L2 -> LDS by global_load_lds, double-buffered, one
// workgroup barrier per tile.  The product is issued TRANSPOSED (weight rows as the MFMA's row operand, the slab as its column
// operand) with the tile's weight rows permuted among the MFMA blocks so that a lane ends with 8 / 16 CONSECUTIVE columns of one
// output row: the epilogue (gemm_epi.h) runs straight out of the accumulators -- no LDS transpose, no per-wave LDS at all.
// What bounds it (round 4, tools/rb16_ablate.py -> profiles/r04_rb16_ablate.txt; 50 000 rows, N = 1024, bf16 out: 41 us): with no
// weight stream 38, with no stores 32, with one operand read per tile instead of eight 36, with all three gone 26 -- against 13.6 us
// of MFMA issue on the SIMD that carries 4 of the 13 waves.  The NewGELU + codes write-out: 70 us, of which 22 are the activation's
// VALU work (56 with the stores gone, 34 with the math gone too) and 14 the store stream.  An MFMA wave and a VALU (or LDS-read)
// wave on ONE SIMD do not overlap on this chip (tools/probes/coissue_probe.hip -> profiles/r04_coissue_probe.txt: MFMA 489 us,
// plain fma 263, both on one SIMD 693; exp2 258 -> 753; packed f32 fma runs at HALF rate, 487 -> 928), so the phases of a tile add
// up whichever wave they sit in: static wave priorities (the waves of a SIMD taking turns: -DR16_PRIO=1) and a raised MFMA phase
// (=2) change nothing, and neither does a workgroup of 12 instead of 13 waves (49 152 rows).
// Delayed stores -- a tile's packed outputs kept in registers across the tile barrier and stored at the top of the next tile, BEHIND
// that tile's weight DMA, so that the wait after the MFMA phase becomes vmcnt(2 or 3) and never covers a store acknowledgement
// (bare s_barrier instead of __syncthreads(), the saved-code loads of EPI_MUL_AUX from inline assembly: otherwise the compiler
// drains vmcnt itself) -- were built, passed the parity tests and measured SLOWER: 22.55 vs 22.14 ms per step (A/B/A/B on one
// box), 74.5 vs 70.7 us on the NewGELU launch.  The stores cost issue / write-path time, not acknowledgement latency.
// Two weight tiles per workgroup barrier (2 x 64 KiB of tile buffers, the first tile's stores in flight during the second tile's MFMA
// phase, half the barriers and DMA waits) were built and measured too: 42.3 / 75.6 / 35.5 us against 42.0 / 72.8 / 36.5 on the three
// shapes of tools/rb16_ablate.py, 22.24 ms per step either way -- the barrier count is not the lever.
#include <cstdlib>
#include "gemm_epi.h"

#define R16_K 256
#define R16_BN 64
#define R16_TILE_HALFS (R16_BN * R16_K)          // 32 KiB per buffer
#define R16_BIAS_MAX 4096                        // bias vectors up to this many columns are staged in LDS (longer: not this kernel)
#define R16_MAXW 16
#ifndef R16_TURNS
#define R16_TURNS 4    // probe builds: -DR16_TURNS=3 is exact for >= 11 waves
#endif
#ifndef R16_PRIO
#define R16_PRIO 0       // probe: 1 = static wave priority 3 - (wave >> 2) (the waves of a SIMD take turns), 2 = MFMA phase raised
#endif
#ifndef R16_ABLATE
#define R16_ABLATE 0     // probe builds (COATI_AMD_CXXFLAGS=-DR16_ABLATE=bits, tools/rb16_ablate.py): timing only, results are wrong
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned r16_v4u __attribute__((ext_vector_type(4)));
#if R16_ABLATE & 16
__device__ unsigned char r16_sink[512 * 4096];
#endif

// cur / nxt reach the tile body as __restrict__ parameters: the compiler otherwise waits for every pending global_load_lds
// before an LDS read it cannot prove disjoint from the DMA's target
template <typename F>
__device__ __forceinline__ void r16_call_restrict(F&& f, int jt, int jn, const bf16_t* __restrict__ cur, bf16_t* __restrict__ nxt) {
  f(jt, jn, cur, nxt);
}

// Output mapping (no LDS transpose).  The MFMA runs as D = Wblk (16 weight rows x 32 k) * slab^T: lane (m = lane & 15, kq = lane >> 4)
// then holds row m of the slab's output and, for column block a, the four weight rows i = 4 kq + r.  WHICH weight row of the tile
// sits at (a, i) is free -- it only decides the LDS address of the operand read:
//   MAP 0:  n = 32 (a >> 1) + 8 kq + 4 (a & 1) + r : the lane owns columns 8 kq .. + 7 and 32 + 8 kq .. + 7; one 16-B store per lane
//           covers a contiguous 64-B half line per row (4 lanes)
//   MAP 1:  n = 16 kq + 4 a + r : the lane owns the 16 consecutive columns 16 kq .. + 15 = one head of 16: the rotary partner of every
//           element is in the same lane (EPI_QKV_ROPE)
// The tile's 16-B chunk c of weight row n sits at chunk position c ^ f(n) with f's low 4 bits = i: the 16 lanes of an operand read
// that share k hit 16 different bank groups.
template <int MAP>
__device__ __forceinline__ int r16_wrow(int a, int i) {
  return MAP == 0 ? 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3) : 16 * (i >> 2) + 4 * a + (i & 3);
}
template <int MAP>
__device__ __forceinline__ int r16_swz(int n) {
  return MAP == 0 ? ((n & 3) | (((n >> 3) & 3) << 2) | (((n >> 2) & 1) << 4)) : ((n & 3) | (((n >> 4) & 3) << 2) | (((n >> 2) & 1) << 4));
}

template <int EPI, bool LN>
__global__ __launch_bounds__(64 * R16_MAXW, 1) void gemm_rb16_kernel(GemmArgs p, int W, int rot) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAP = (EPI == EPI_QKV_ROPE || EPI == EPI_GELU_GRAD) ? 1 : 0;
  bf16_t* const Bs = reinterpret_cast<bf16_t*>(smem);
  float* const BiasS = reinterpret_cast<float*>(smem + 2 * R16_TILE_HALFS * 2);   // bias[N] (+ one tile of slack), read as broadcasts
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = (blockIdx.x * W + wave) * 16;
  const int fr = lane & 15, kq = lane >> 4;       // this lane's row of the slab, its 8-k group inside a 32-k step
  const int ntiles = (p.N + R16_BN - 1) / R16_BN;
  const bool has_bias = p.bias != nullptr;
  if (has_bias) {
    for (int c = tid; c < ntiles * R16_BN; c += blockDim.x) BiasS[c] = c < p.N ? p.bias[c] : 0.f;
  }

  // ---- resident A slab: fragment ks holds k = 32 ks + 8 kq .. + 7 of row fr
  bf16x8 af[8];
  const int row_l = m0 + fr, rc = row_l < p.M ? row_l : p.M - 1;
  if constexpr (!LN) {
    const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.A) + (long long)rc * p.lda + kq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 32);
  } else {
    // LayerNorm inside the slab load: the 4 lanes (fr, kq = 0..3) hold one f32 row of 256; two-pass statistics (as ln_fwd_kernel),
    // gamma / beta as LDS broadcasts out of the still idle second tile buffer; the normalised row goes to the MFMA fragments and
    // to p.A (the copy the weight gradient reads), mean / rstd to the backward
    const float* xp = p.ln_x + (long long)rc * p.ln_ldx + kq * 8;
    float xf[8][8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float4 x0 = *reinterpret_cast<const float4*>(xp + ks * 32), x1 = *reinterpret_cast<const float4*>(xp + ks * 32 + 4);
      xf[ks][0] = x0.x; xf[ks][1] = x0.y; xf[ks][2] = x0.z; xf[ks][3] = x0.w; xf[ks][4] = x1.x; xf[ks][5] = x1.y; xf[ks][6] = x1.z; xf[ks][7] = x1.w;
    }
    float* const GB = reinterpret_cast<float*>(Bs + R16_TILE_HALFS);
    if (tid < 128) {
      const float* src = tid < 64 ? p.ln_gamma + 4 * tid : p.ln_beta + 4 * (tid - 64);
      *reinterpret_cast<float4*>(GB + 4 * tid) = *reinterpret_cast<const float4*>(src);
    }
    float sm = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) sm += xf[ks][i];
    sm += __shfl_xor(sm, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    const float mean = sm * (1.0f / R16_K);
    float vs = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = xf[ks][i] - mean; vs = fmaf(d, d, vs); }
    vs += __shfl_xor(vs, 16, 64);
    vs += __shfl_xor(vs, 32, 64);
    const float rstd = rsqrtf(vs * (1.0f / R16_K) + 1e-5f);
    if (kq == 0 && row_l < p.M) { p.ln_mean[row_l] = mean; p.ln_rstd[row_l] = rstd; }
    __syncthreads();   // gamma / beta are in LDS
    bf16_t* op = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.A)) + (long long)rc * p.lda + kq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float4 g0 = *reinterpret_cast<const float4*>(GB + ks * 32 + kq * 8), g1 = *reinterpret_cast<const float4*>(GB + ks * 32 + kq * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(GB + 256 + ks * 32 + kq * 8), b1 = *reinterpret_cast<const float4*>(GB + 256 + ks * 32 + kq * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (xf[ks][i] - mean) * rstd * g[i] + bt[i];
      const uint4 u = pack8(o);
      af[ks] = __builtin_bit_cast(bf16x8, u);
      // (non-temporal: only the backward's weight gradient reads this copy, a whole forward pass later -- it should not push the rows
      //  the next launch reads out of the caches.  With the NewGELU' codes below: 21.05 vs 21.10 ms per step, A/B x 3 on one box)
      if (row_l < p.M) __builtin_nontemporal_store(r16_v4u{u.x, u.y, u.z, u.w}, reinterpret_cast<r16_v4u*>(op + ks * 32));
    }
    __syncthreads();   // everyone has read gamma / beta: the buffer may receive its weight tile
  }
  // rotary table row of this lane's slab row: [8 cos | 8 sin], the same for every head = every tile: registers for the whole kernel
  float rc_[8], rs_[8];
  if constexpr (EPI == EPI_QKV_ROPE) {
    const int t = p.rope_row_t != nullptr ? p.rope_row_t[rc] : rc % p.rope_T;
    const float4 c0 = *reinterpret_cast<const float4*>(p.rope_cos + t * 16), c1 = *reinterpret_cast<const float4*>(p.rope_cos + t * 16 + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(p.rope_sin + t * 16), s1 = *reinterpret_cast<const float4*>(p.rope_sin + t * 16 + 4);
    rc_[0] = c0.x; rc_[1] = c0.y; rc_[2] = c0.z; rc_[3] = c0.w; rc_[4] = c1.x; rc_[5] = c1.y; rc_[6] = c1.z; rc_[7] = c1.w;
    rs_[0] = s0.x; rs_[1] = s0.y; rs_[2] = s0.z; rs_[3] = s0.w; rs_[4] = s1.x; rs_[5] = s1.y; rs_[6] = s1.z; rs_[7] = s1.w;
  }

  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;
  // weight tile: 32 pieces of 1 KiB (two 512-B rows), piece k by wave k % W; chunk c of row r at position c ^ f(r)
  auto load_tile = [&](int n0, bf16_t* S) {
#pragma unroll
    for (int i = 0; i < R16_TURNS; ++i) {      // W >= 8: 4 turns cover the 32 pieces.  (7 turns, to admit 5-wave workgroups for a threshold probe, cost the QKV launch 29 %: 46.5 -> 59.9 us)
      const int k = wave + W * i;
      if (k < R16_BN / 2) {
        const int r = 2 * k + (lane >> 5), q = lane & 31;
        const int g = n0 + r, gc = g < p.N ? g : p.N - 1;
        const bf16_t* src = p.B + (long long)gc * p.ldb + ((q ^ r16_swz<MAP>(r)) * 8);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(S + k * 512), 16, 0, 0);
      }
    }
  };
  GemmArgs q = p;
  q.bias = nullptr;   // folded into the accumulator initialisation
#if R16_ABLATE & 16       // probe builds: bit 4 = every store goes to a 4-KiB region per workgroup (L2 hits, no HBM write stream)
  q.C = r16_sink + (blockIdx.x & 255) * 4096; q.ldc = 0;
  q.aux_out = r16_sink + (256 + (blockIdx.x & 255)) * 4096; q.ld_aux = 0;
#define p q
#endif

#if R16_PRIO == 1
  switch (__builtin_amdgcn_readfirstlane(wave) >> 2) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(2); break;
    case 2: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
  }
#endif
  const int j0 = rot ? (int)(blockIdx.x % (unsigned)ntiles) : 0;
  load_tile(j0 * R16_BN, Bs);
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();                      // (also: the bias vector is in LDS)

  // this lane's operand rows and its output columns inside a tile
  int wofs[4], wsw[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = r16_wrow<MAP>(a, fr);
    wofs[a] = n * R16_K;
    wsw[a] = r16_swz<MAP>(n);
  }
  const int colA = MAP == 0 ? 8 * kq : 16 * kq, colB = MAP == 0 ? 32 + 8 * kq : 16 * kq + 8;   // acc[0..1] -> colA .. + 7, acc[2..3] -> colB .. + 7
#if R16_ABLATE & 2         // probe builds: bit 1 = no stores in the write-out
  const bool rowok = row_l < 0;
#else
  const bool rowok = row_l < p.M;
#endif

  auto tile = [&](int jt, int jn, const bf16_t* cur, bf16_t* nxt) {
#if !(R16_ABLATE & 1)     // probe builds (tools/rb16_ablate.py): bit 0 = no weight stream behind the first tile
    load_tile(jn * R16_BN, nxt);
#endif
    uint4 xq = make_uint4(0, 0, 0, 0);
    if constexpr (EPI == EPI_MUL_AUX) {
      // the 16 saved NewGELU' codes of this lane's columns (MAP 0: two 8-B pieces), in flight during the MFMA phase
      const unsigned char* X = reinterpret_cast<const unsigned char*>(p.aux_in) + (long long)rc * p.ld_aux + jt * R16_BN;
      const int nb = p.N - jt * R16_BN;   // (N % 16 == 0: an 8-column piece is inside or outside as a whole; outside ones are not stored)
      const uint2 x0 = *reinterpret_cast<const uint2*>(X + (colA + 8 <= nb ? colA : 0)), x1 = *reinterpret_cast<const uint2*>(X + (colB + 8 <= nb ? colB : 0));
      xq = make_uint4(x0.x, x0.y, x1.x, x1.y);
    }
    f32x4_t acc[4];
    if (has_bias) {
      acc[0] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colA);
      acc[1] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colA + 4);
      acc[2] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colB);
      acc[3] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colB + 4);
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    {
#if R16_PRIO == 2
      __builtin_amdgcn_s_setprio(3);
#endif
      bf16x8 wf[2][4];
#pragma unroll
      for (int a = 0; a < 4; ++a) wf[0][a] = *reinterpret_cast<const bf16x8*>(cur + wofs[a] + ((kq ^ wsw[a]) * 8));
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
#if R16_ABLATE & 4        // probe builds: bit 2 = one LDS operand read per tile instead of eight
        if (ks == 0) {
#pragma unroll
          for (int a = 0; a < 4; ++a) wf[1][a] = wf[0][a];
        }
#else
        if (ks + 1 < 8) {
#pragma unroll
          for (int a = 0; a < 4; ++a) wf[(ks + 1) & 1][a] = *reinterpret_cast<const bf16x8*>(cur + wofs[a] + (((4 * (ks + 1) + kq) ^ wsw[a]) * 8));
        }
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][a], af[ks], acc[a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#if R16_PRIO == 2
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);   // the next tile has landed (this wave's pieces): wait BEFORE this tile's stores are issued
    float v0[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
    float v1[8] = {acc[2][0], acc[2][1], acc[2][2], acc[2][3], acc[3][0], acc[3][1], acc[3][2], acc[3][3]};
    const int c0 = jt * R16_BN + colA, c1 = jt * R16_BN + colB;
    if constexpr (EPI == EPI_CE_PARTIAL) {
      // (max, sum exp) over the tile's 64 columns of row fr: 16 in this lane, the rest in lanes fr + 16, + 32, + 48
      // (the activation write-outs add their VALU slots to the tile time -- an MFMA wave and a VALU wave of one SIMD do not overlap,
      //  see the header -- so the column masks are paid only by the one tile that straddles N: V = 10 322 = 161 full tiles + 18 columns)
      float mx = -INFINITY, sm = 0.f;
      if ((jt + 1) * R16_BN <= p.N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fmaxf(v0[e], v1[e]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += __expf(v0[e] - mx) + __expf(v1[e] - mx);   // (subtraction first: see EPI_CE_BWD in gemm_epi.h)
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (c0 + e < p.N) mx = fmaxf(mx, v0[e]);
          if (c1 + e < p.N) mx = fmaxf(mx, v1[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (c0 + e < p.N) sm += __expf(v0[e] - mx);
          if (c1 + e < p.N) sm += __expf(v1[e] - mx);
        }
      }
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      if (rowok && kq == 0) p.partial[(long long)row_l * ntiles + jt] = make_float2(mx, sm);
    } else if constexpr (EPI == EPI_QKV_ROPE) {
      // RotaryEmbedding.rotary_embed (basic_transformer.py:83-100) on one head of 16: v0 = dims 0..7, v1 = dims 8..15 of the same
      // head: y_i = x_i c_i - x_{i+8} s_i ; y_{i+8} = x_{i+8} c_i + x_i s_i.  The v block (columns >= 2C) is stored as it is
      // (wave-uniform: a 64-column tile never straddles 2C)
      if (jt * R16_BN < 2 * p.rope_C) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float lo = v0[e], hi = v1[e];
          v0[e] = fmaf(-hi, rs_[e], lo * rc_[e]);
          v1[e] = fmaf(lo, rs_[e], hi * rc_[e]);
        }
      }
      epilogue8<EPI_BF16>(q, row_l, c0, v0, rowok, jt, ntiles);
      epilogue8<EPI_BF16>(q, row_l, c1, v1, rowok, jt, ntiles);
    } else if constexpr (EPI == EPI_GELU_GRAD) {
      // 16 consecutive columns: NewGELU -> two 16-B stores (32 B of the row), NewGELU' (8-bit fixed point, common.h) -> ONE 16-B store
      float d0[8], d1[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        coati_v2f hh, dd;
        gelu_and_grad_f2(coati_v2f{v0[e], v0[e + 1]}, hh, dd);
        v0[e] = hh.x; v0[e + 1] = hh.y; d0[e] = dd.x; d0[e + 1] = dd.y;
        gelu_and_grad_f2(coati_v2f{v1[e], v1[e + 1]}, hh, dd);
        v1[e] = hh.x; v1[e + 1] = hh.y; d1[e] = dd.x; d1[e + 1] = dd.y;
      }
#if R16_ABLATE & 8       // probe builds: bit 3 (with bit 1) = the activation math stays, its stores do not
      float cks = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) cks += v0[e] + v1[e] + d0[e] + d1[e];
      if ((rowok || cks == 1234.5f) && c0 + 16 <= p.N) {
#else
      if (rowok && c0 + 16 <= p.N) {   // (N % 16 == 0)
#endif
        const uint2 q0 = packq8(d0), q1 = packq8(d1);
        __builtin_nontemporal_store(r16_v4u{q0.x, q0.y, q1.x, q1.y}, reinterpret_cast<r16_v4u*>(reinterpret_cast<unsigned char*>(p.aux_out) + (unsigned)row_l * (unsigned)p.ld_aux + (unsigned)c0));   // (read by the backward only)
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + (unsigned)row_l * (unsigned)p.ldc + (unsigned)c0;
        *reinterpret_cast<uint4*>(C) = pack8(v0);
        *reinterpret_cast<uint4*>(C + 8) = pack8(v1);
      }
    } else if constexpr (EPI == EPI_MUL_AUX) {
      const uint2 x0 = make_uint2(xq.x, xq.y), x1 = make_uint2(xq.z, xq.w);
      epilogue8<EPI>(q, row_l, c0, v0, rowok, jt, ntiles, &x0);
      epilogue8<EPI>(q, row_l, c1, v1, rowok, jt, ntiles, &x1);
    } else {
      epilogue8<EPI>(q, row_l, c0, v0, rowok, jt, ntiles);
      epilogue8<EPI>(q, row_l, c1, v1, rowok, jt, ntiles);
    }
    __syncthreads();
  };
  for (int j = 0, jt = j0; j < ntiles; ++j) {
    const int jn = jt + 1 == ntiles ? 0 : jt + 1;
    r16_call_restrict(tile, jt, jn, Bs + (j & 1) * R16_TILE_HALFS, Bs + ((j + 1) & 1) * R16_TILE_HALFS);
    jt = jn;
  }
#if R16_ABLATE & 16
#undef p
#endif
}

// ---- weight-resident form for N <= 256 (round 3): the E(3)-GNN's edge-level products ------------------------------------
// K = 256, N = 256: the whole weight (128 KiB) stays in LDS for the lifetime of a PERSISTENT workgroup; the waves then walk over
// 16-row slabs independently -- no tile barriers at all.  The row count of the edge products is data dependent (compacted
// neighbour list, read on the device): the 32-row kernel sized its grid for the worst case (all A (A - 1) pairs) and the
// ~125 000 edges of a batch became 1.5 rounds of 10-slab workgroups; here every workgroup derives its equal share of the slabs
// that exist from the device-side count.  Epilogues: bf16 (+ bias) and the GNN's dSiLU(recomputed pre-activation) multiply
// (EPI_EDGE_DPRE; its per-column constants w1c / b1 staged in LDS next to the bias).
#define R16_RES_TILES 4
template <int EPI>
__global__ __launch_bounds__(64 * R16_MAXW, 1) void gemm_rb16_resident_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAP = 0;
  bf16_t* const Bs = reinterpret_cast<bf16_t*>(smem);
  float* const BiasS = reinterpret_cast<float*>(smem + R16_RES_TILES * R16_TILE_HALFS * 2);   // bias[256]
  float* const ColS = BiasS + R16_RES_TILES * R16_BN;                                          // EPI_EDGE_DPRE: per tile [64 w1c | 64 b1]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = blockDim.x >> 6;
  const int fr = lane & 15, kq = lane >> 4;
  const int ntiles = (p.N + R16_BN - 1) / R16_BN;
  const int M = __builtin_amdgcn_readfirstlane(p.m_dev != nullptr ? *p.m_dev : p.M);   // (scalar: everything derived from it stays on the scalar unit)
  const bool has_bias = p.bias != nullptr;
  if (M <= 0) return;
  for (int c = tid; c < ntiles * R16_BN; c += blockDim.x) {
    BiasS[c] = (has_bias && c < p.N) ? p.bias[c] : 0.f;
    if constexpr (EPI == EPI_EDGE_DPRE) {
      const int jt = c / R16_BN, e = c % R16_BN;
      ColS[jt * 128 + e] = c < p.N ? p.w1c[(long long)c * p.w1c_stride] : 0.f;
      ColS[jt * 128 + 64 + e] = c < p.N ? p.b1[c] : 0.f;
    }
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;
  // the whole weight: ntiles x 32 pieces of 1 KiB (two 512-B rows), piece k by wave k % W; chunk c of row r at position c ^ f(r)
  for (int k = wave; k < ntiles * (R16_BN / 2); k += W) {
    const int r = 2 * k + (lane >> 5), q = lane & 31;     // r = weight row (all tiles), 0 .. 64 ntiles - 1
    const int gc = r < p.N ? r : p.N - 1;
    const bf16_t* src = p.B + (long long)gc * p.ldb + ((q ^ r16_swz<MAP>(r & 63)) * 8);
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(Bs + k * 512), 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();

  int wofs[4], wsw[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = r16_wrow<MAP>(a, fr);
    wofs[a] = n * R16_K;
    wsw[a] = r16_swz<MAP>(n);
  }
  const int colA = 8 * kq, colB = 32 + 8 * kq;
  GemmArgs q = p;
  q.bias = nullptr;   // folded into the accumulator initialisation
  q.M = M;

  // this workgroup's equal share of the slabs that exist; its waves take them round-robin
  const int slabs = (M + 15) / 16, per_wg = (slabs + (int)gridDim.x - 1) / (int)gridDim.x;
  const int s_begin = blockIdx.x * per_wg, s_end = s_begin + per_wg < slabs ? s_begin + per_wg : slabs;
  for (int sl = s_begin + wave; sl < s_end; sl += W) {
    const int m0 = sl * 16, row_l = m0 + fr, rc = row_l < M ? row_l : M - 1;
    const bool rowok = row_l < M;
    bf16x8 af[8];
    {
      const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.A) + (long long)rc * p.lda + kq * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 32);
    }
    for (int jt = 0; jt < ntiles; ++jt) {
      const bf16_t* cur = Bs + jt * R16_TILE_HALFS;
      f32x4_t acc[4];
      acc[0] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colA);
      acc[1] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colA + 4);
      acc[2] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colB);
      acc[3] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colB + 4);
      {
        // operand fragments: two sets in flight (bf16 epilogue) or one (EPI_EDGE_DPRE: its gathers need the 16 registers; the
        // other three waves of the SIMD cover the LDS latency)
        constexpr int WFD = (EPI == EPI_EDGE_DPRE) ? 1 : 2;
        bf16x8 wf[WFD][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[0][a] = *reinterpret_cast<const bf16x8*>(cur + wofs[a] + ((kq ^ wsw[a]) * 8));
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (WFD == 2 && ks + 1 < 8) {
#pragma unroll
            for (int a = 0; a < 4; ++a) wf[(ks + 1) % WFD][a] = *reinterpret_cast<const bf16x8*>(cur + wofs[a] + (((4 * (ks + 1) + kq) ^ wsw[a]) * 8));
          }
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks % WFD][a], af[ks], acc[a], 0, 0, 0);
          if (WFD == 1 && ks + 1 < 8) {
#pragma unroll
            for (int a = 0; a < 4; ++a) wf[0][a] = *reinterpret_cast<const bf16x8*>(cur + wofs[a] + (((4 * (ks + 1) + kq) ^ wsw[a]) * 8));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      float v0[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
      float v1[8] = {acc[2][0], acc[2][1], acc[2][2], acc[2][3], acc[3][0], acc[3][1], acc[3][2], acc[3][3]};
      const int c0 = jt * R16_BN + colA, c1 = jt * R16_BN + colB;
      const void* st0 = nullptr;
      const void* st1 = nullptr;
      if constexpr (EPI == EPI_EDGE_DPRE) { st0 = ColS + jt * 128 + colA; st1 = ColS + jt * 128 + colB; }
      epilogue8<EPI>(q, row_l, c0, v0, rowok, jt, ntiles, st0);
      __builtin_amdgcn_sched_barrier(0);   // one 8-column group after the other: interleaved, the two gathers of EPI_EDGE_DPRE spill
      epilogue8<EPI>(q, row_l, c1, v1, rowok, jt, ntiles, st1);
    }
  }
}

bool gemm_rb16_resident_supported(const GemmArgs& a, int a_f32, int epi) {
  if (a_f32 || a.K != R16_K || a.m_dev == nullptr) return false;   // (the transformer's K = 256, N = 256 products: the ring kernel is faster -- proj dgrad 0.63 vs 0.66 ms per step)
  if (epi != EPI_BF16 && epi != EPI_EDGE_DPRE) return false;   // (the GNN's node-level products, 16 384 rows, were tried here with 4-wave workgroups: 0.68 vs 0.64 ms per step on the tiled kernel)
  if (a.N % 16 != 0 || a.N > R16_RES_TILES * R16_BN || a.q8_out != nullptr || a.ln_x != nullptr) return false;
  return a.M >= 16 * 256;
}

template <int EPI>
static int launch_rb16_resident_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_rb16_resident_kernel<EPI>;
  constexpr size_t lds = (size_t)R16_RES_TILES * R16_TILE_HALFS * 2 + (size_t)R16_RES_TILES * R16_BN * 4 + (size_t)R16_RES_TILES * 128 * 4;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      coati_set_error("gemm_rb16(resident): hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * R16_MAXW), lds, s, a);
  COATI_LAUNCH_CHECK("gemm_rb16_resident");
  return COATI_OK;
}

int launch_gemm_rb16_resident(const GemmArgs& a, int epi, hipStream_t s) {
  if (epi == EPI_EDGE_DPRE) return launch_rb16_resident_t<EPI_EDGE_DPRE>(a, s);
  return launch_rb16_resident_t<EPI_BF16>(a, s);
}

// ---- the point encoder's edge path, forward, as ONE launch per layer (round 6) ---------------------------------------------------------
// e_gcl_sparse.edge_model + the per-receiver sum of e_gcl_sparse.forward (e_gcl_sparse.py:169-215, 297-321):
//   e1[e] = SiLU(Pa[bj] + Pb[bk] + d2[e] w1c + b1)        (the first edge Linear, factored: P = [h W1a^T | h W1b^T] per node)
//   s2[e] = e1[e] W3^T + b3                               (the second edge Linear)
//   mi[bj] = sum over the receiver's edges of SiLU(s2[e]) w[e]
// Three launches so far (gnn_edge_pre_c, the weight-resident product above, gnn_edge_reduce_c): e1 and s2 -- 107 MB each per layer at the
// bench batch -- were written, read back by the next launch, and read again by the backward.  Here the same weight-resident workgroup owns
// RECEIVERS: a wave builds the 16-edge slab of one receiver in its MFMA operand registers (the two gathers of P, SiLU), stores it as e1 for
// the backward's weight gradient, runs the 256 x 256 product against the resident W3, stores s2 for the backward, and sums SiLU(s2) w over
// the slab's rows with four DPP steps per value (the 16 rows of a slab are the 16 lanes of a DPP row) -- receivers with more than 16 edges
// take more slabs, summed in the wave's LDS strip.  e1 and s2 are written once and not read back in the forward.  H = 256 only.
template <int CTRL>
__device__ __forceinline__ float r16_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float r16_row_sum(float v) {   // sum over the 16 lanes of a DPP row, every lane ends with the total
  v += r16_dpp<0xB1>(v);    // quad_perm(1, 0, 3, 2)
  v += r16_dpp<0x4E>(v);    // quad_perm(2, 3, 0, 1)
  v += r16_dpp<0x141>(v);   // row_half_mirror
  v += r16_dpp<0x140>(v);   // row_mirror
  return v;
}

struct GnnEdgeFwdArgs {
  const bf16_t* P; long long ldp;       // [BA, 2H] (Pa | Pb)
  const int* seg;                       // [BA + 1] receiver segments of the compacted edge list
  const int* e_bk; const float* e_d2; const float* e_w;
  const float* w1c; long long w1c_stride; const float* b1;
  const bf16_t* W3; long long ldw; const float* b3;   // [H, H] bf16 shadow, bias f32
  bf16_t* e1; bf16_t* s2;               // [E, H] each (saved for the backward)
  bf16_t* mi; long long ldmi;           // [BA, ldmi]
  int BA;
};

__global__ __launch_bounds__(64 * R16_MAXW, 1) void gnn_edge_fwd_fused_kernel(GnnEdgeFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAP = 0, H = R16_K, NT = R16_RES_TILES;
  bf16_t* const Bs = reinterpret_cast<bf16_t*>(smem);
  float* const BiasS = reinterpret_cast<float*>(smem + NT * R16_TILE_HALFS * 2);   // b3[256]
  float* const K1 = BiasS + H;                                                      // [w1c[256] | b1[256]]
  float* const MS = K1 + 2 * H;                                                     // per wave: 256 partial sums
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = blockDim.x >> 6;
  const int fr = lane & 15, kq = lane >> 4;
  for (int c = tid; c < H; c += blockDim.x) {
    BiasS[c] = p.b3[c];
    K1[c] = p.w1c[(long long)c * p.w1c_stride];
    K1[H + c] = p.b1[c];
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;
  for (int k = wave; k < NT * (R16_BN / 2); k += W) {
    const int r = 2 * k + (lane >> 5), q = lane & 31;
    const bf16_t* src = p.W3 + (long long)r * p.ldw + ((q ^ r16_swz<MAP>(r & 63)) * 8);
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(Bs + k * 512), 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();

  int wofs[4], wsw[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = r16_wrow<MAP>(a, fr);
    wofs[a] = n * R16_K;
    wsw[a] = r16_swz<MAP>(n);
  }
  const int colA = 8 * kq, colB = 32 + 8 * kq;
  float* const ms = MS + wave * H;

  const int per_wg = (p.BA + (int)gridDim.x - 1) / (int)gridDim.x;
  const int j_begin = blockIdx.x * per_wg, j_end = j_begin + per_wg < p.BA ? j_begin + per_wg : p.BA;
  for (int bj = j_begin + wave; bj < j_end; bj += W) {
    const int e0 = p.seg[bj], n = p.seg[bj + 1] - e0;
    if (n <= 0) {   // a receiver without edges (a padding atom): mi = 0
      *reinterpret_cast<uint2*>(p.mi + (long long)bj * p.ldmi + 4 * lane) = make_uint2(0u, 0u);
      continue;
    }
    for (int ch = 0; ch * 16 < n; ++ch) {
      const bool ok = ch * 16 + fr < n;
      const int e = ok ? e0 + ch * 16 + fr : e0;
      const int bk = p.e_bk[e];
      const float d2 = p.e_d2[e];
      const float w = ok ? p.e_w[e] : 0.f;
      // ---- the slab: row fr = edge e, fragment ks = features 32 ks + 8 kq .. + 7 of e1[e]
      bf16x8 af[8];
      {
        const bf16_t* pa = p.P + (long long)bj * p.ldp + kq * 8;
        const bf16_t* pb = p.P + (long long)bk * p.ldp + H + kq * 8;
        uint4 ua[8], ub[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { ua[ks] = *reinterpret_cast<const uint4*>(pa + ks * 32); ub[ks] = *reinterpret_cast<const uint4*>(pb + ks * 32); }
        bf16_t* e1row = p.e1 + (long long)e * H + kq * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          float a8[8], b8[8], o[8];
          unpack8(ua[ks], a8);
          unpack8(ub[ks], b8);
          const float4 c0 = *reinterpret_cast<const float4*>(K1 + ks * 32 + kq * 8), c1 = *reinterpret_cast<const float4*>(K1 + ks * 32 + kq * 8 + 4);
          const float4 g0 = *reinterpret_cast<const float4*>(K1 + H + ks * 32 + kq * 8), g1 = *reinterpret_cast<const float4*>(K1 + H + ks * 32 + kq * 8 + 4);
          const float wc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, bb[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = silu_f(a8[i] + b8[i] + d2 * wc[i] + bb[i]);
          const uint4 u = pack8(o);
          af[ks] = __builtin_bit_cast(bf16x8, u);
          if (ok) *reinterpret_cast<uint4*>(e1row + ks * 32) = u;
        }
      }
      // ---- s2 = e1 W3^T + b3 against the resident weight, tile by tile; SiLU(s2) w summed over the slab's rows
      for (int jt = 0; jt < NT; ++jt) {
        const bf16_t* cur = Bs + jt * R16_TILE_HALFS;
        f32x4_t acc[4];
        acc[0] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colA);
        acc[1] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colA + 4);
        acc[2] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colB);
        acc[3] = *reinterpret_cast<const f32x4_t*>(BiasS + jt * R16_BN + colB + 4);
        {
          bf16x8 wf[2][4];
#pragma unroll
          for (int a = 0; a < 4; ++a) wf[0][a] = *reinterpret_cast<const bf16x8*>(cur + wofs[a] + ((kq ^ wsw[a]) * 8));
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            if (ks + 1 < 8) {
#pragma unroll
              for (int a = 0; a < 4; ++a) wf[(ks + 1) & 1][a] = *reinterpret_cast<const bf16x8*>(cur + wofs[a] + (((4 * (ks + 1) + kq) ^ wsw[a]) * 8));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][a], af[ks], acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        float v0[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
        float v1[8] = {acc[2][0], acc[2][1], acc[2][2], acc[2][3], acc[3][0], acc[3][1], acc[3][2], acc[3][3]};
        const uint4 u0 = pack8(v0), u1 = pack8(v1);
        if (ok) {
          bf16_t* srow = p.s2 + (long long)e * H + jt * R16_BN;
          *reinterpret_cast<uint4*>(srow + colA) = u0;
          *reinterpret_cast<uint4*>(srow + colB) = u1;
        }
        // (SiLU of the ROUNDED value: what the backward re-evaluates from the saved bf16 s2)
        unpack8(u0, v0);
        unpack8(u1, v1);
        float t0[8], t1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { t0[i] = r16_row_sum(silu_f(v0[i]) * w); t1[i] = r16_row_sum(silu_f(v1[i]) * w); }
        if (fr == 0) {
          float* d0 = ms + jt * R16_BN + colA;
          float* d1 = ms + jt * R16_BN + colB;
          if (ch > 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { t0[i] += d0[i]; t1[i] += d1[i]; }
          }
          *reinterpret_cast<float4*>(d0) = make_float4(t0[0], t0[1], t0[2], t0[3]);
          *reinterpret_cast<float4*>(d0 + 4) = make_float4(t0[4], t0[5], t0[6], t0[7]);
          *reinterpret_cast<float4*>(d1) = make_float4(t1[0], t1[1], t1[2], t1[3]);
          *reinterpret_cast<float4*>(d1 + 4) = make_float4(t1[4], t1[5], t1[6], t1[7]);
        }
      }
    }
    // the receiver's 256 sums: this wave's own LDS strip (LDS operations of one wave complete in order), 4 channels per lane
    const float4 m4 = *reinterpret_cast<const float4*>(ms + 4 * lane);
    *reinterpret_cast<uint2*>(p.mi + (long long)bj * p.ldmi + 4 * lane) = make_uint2(pack2bf(m4.x, m4.y), pack2bf(m4.z, m4.w));
  }
}

int launch_gnn_edge_fwd_fused(const bf16_t* P, long long ldp, const int* seg, const int* e_bk, const float* e_d2, const float* e_w,
                              const float* w1c, long long w1c_stride, const float* b1, const bf16_t* W3, long long ldw, const float* b3,
                              bf16_t* e1, bf16_t* s2, bf16_t* mi, long long ldmi, int BA, int H, hipStream_t s) {
  COATI_CHECK_ARG(P && seg && e_bk && e_d2 && e_w && w1c && b1 && W3 && b3 && e1 && s2 && mi, "gnn_edge_fwd_fused: null operand");
  COATI_CHECK_SHAPE(H == R16_K && BA > 0 && ldp % 8 == 0 && ldw % 8 == 0 && ldmi % 4 == 0, "gnn_edge_fwd_fused: H = 256 only (H=%d)", H);
  static bool attr_set = false;
  constexpr size_t lds = (size_t)R16_RES_TILES * R16_TILE_HALFS * 2 + (size_t)R16_K * 4 * 3 + (size_t)R16_MAXW * R16_K * 4;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gnn_edge_fwd_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      coati_set_error("gnn_edge_fwd_fused: hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_set = true;
  }
  GnnEdgeFwdArgs a;
  a.P = P; a.ldp = ldp; a.seg = seg; a.e_bk = e_bk; a.e_d2 = e_d2; a.e_w = e_w; a.w1c = w1c; a.w1c_stride = w1c_stride; a.b1 = b1;
  a.W3 = W3; a.ldw = ldw; a.b3 = b3; a.e1 = e1; a.s2 = s2; a.mi = mi; a.ldmi = ldmi; a.BA = BA;
  hipLaunchKernelGGL(gnn_edge_fwd_fused_kernel, dim3(256), dim3(64 * R16_MAXW), lds, s, a);
  COATI_LAUNCH_CHECK("gnn_edge_fwd_fused");
  return COATI_OK;
}

// waves per workgroup for M rows: one round of one workgroup per CU.  (Round 4: TWO workgroups of half the waves per CU -- to fill
// each other's barrier bubbles -- measured 23.04 vs 22.42 ms per step: each workgroup streams the whole weight, twice the L2 -> LDS
// traffic per CU, and the bubbles are not where the time goes, see the header.)
static int rb16_waves(int M) {
  const int slabs = (M + 15) / 16;
  return (slabs + 255) / 256;
}
// fewest waves per workgroup the kernel is chosen for (probe knob COATI_RB16_MINW, >= 8: load_tile covers a tile in 4 turns; default 9 = 36 865 rows)
static int rb16_min_waves() {
  static const int v = []() { const char* e = getenv("COATI_RB16_MINW"); const int w = e ? atoi(e) : 9; return w < 8 ? 8 : w; }();
  return v;
}

bool gemm_rb16_supported(const GemmArgs& a, int a_f32, int epi) {
  if (a_f32 || a.K != R16_K || a.m_dev != nullptr) return false;
  if (epi != EPI_BF16 && epi != EPI_QKV_ROPE && epi != EPI_GELU_GRAD && epi != EPI_MUL_AUX && epi != EPI_CE_PARTIAL && epi != EPI_CE_BWD) return false;
  if (epi == EPI_CE_PARTIAL && a.partial_tile != 64) return false;
  if (epi == EPI_QKV_ROPE && a.rope_hs == 32) return false;
  if (a.N % 16 != 0 && epi != EPI_CE_BWD && epi != EPI_CE_PARTIAL) return false;
  if (a.bias != nullptr && cdiv(a.N, R16_BN) * R16_BN > R16_BIAS_MAX) return false;
  if (a.q8_out != nullptr) return false;                              // (the fused MXFP8 emission assumes epilogue8's lane layout)
  if (epi == EPI_GELU_GRAD && a.n_store > a.N) return false;
  if (epi == EPI_QKV_ROPE && (a.rope_C % 32 != 0 || a.rope_pos != nullptr)) return false;   // 64-column tiles must not straddle 2C
  const int W = rb16_waves(a.M);
  return W >= rb16_min_waves() && W <= R16_MAXW;   // 36 865 .. 65 536 rows: below, the 32-row kernel or the tiled one; above, the 32-row kernel
}

template <int EPI, bool LN>
static int launch_rb16_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_rb16_kernel<EPI, LN>;
  constexpr size_t tile_bytes = (size_t)2 * R16_TILE_HALFS * 2;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(tile_bytes + R16_BIAS_MAX * 4)) != hipSuccess) {
      coati_set_error("gemm_rb16: hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int W = rb16_waves(a.M);
  const int blocks = cdiv(cdiv(a.M, 16), W);
  const int rot = (EPI == EPI_MUL_AUX);
  const size_t bias_bytes = a.bias != nullptr ? (size_t)cdiv(a.N, R16_BN) * R16_BN * 4 : 0;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), tile_bytes + bias_bytes, s, a, W, rot);
  COATI_LAUNCH_CHECK("gemm_rb16");
  return COATI_OK;
}

int launch_gemm_rb16(const GemmArgs& a, int epi, hipStream_t s) {
  const bool ln = a.ln_x != nullptr;
  if (ln && epi != EPI_QKV_ROPE && epi != EPI_GELU_GRAD) {
    coati_set_error("gemm_rb16: epilogue %d has no fused-LayerNorm variant", epi);
    return COATI_EARG;
  }
  switch (epi) {
    case EPI_BF16: return launch_rb16_t<EPI_BF16, false>(a, s);
    case EPI_QKV_ROPE: return ln ? launch_rb16_t<EPI_QKV_ROPE, true>(a, s) : launch_rb16_t<EPI_QKV_ROPE, false>(a, s);
    case EPI_GELU_GRAD: return ln ? launch_rb16_t<EPI_GELU_GRAD, true>(a, s) : launch_rb16_t<EPI_GELU_GRAD, false>(a, s);
    case EPI_MUL_AUX: return launch_rb16_t<EPI_MUL_AUX, false>(a, s);
    case EPI_CE_PARTIAL: return launch_rb16_t<EPI_CE_PARTIAL, false>(a, s);
    case EPI_CE_BWD: return launch_rb16_t<EPI_CE_BWD, false>(a, s);
    default:
      coati_set_error("gemm_rb16: unsupported epilogue %d", epi);
      return COATI_EARG;
  }
}
