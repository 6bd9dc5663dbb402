// Row-block GEMM on 16-row slabs (round 3): the K = 256 products of the transformer at PACKED batch sizes.
//
// Why: the 32-row form (gemm_rb.hip) holds 168 VGPRs per wave, i.e. 2-3 waves per SIMD, and its phases are latency chains of ONE
// wave -- 32 MFMAs, then ~500 dependent-ish epilogue instructions per tile (tools/probes/rb_trace.py: the NewGELU epilogue costs
// the same 5 k cycles per tile with 2 waves per SIMD as with 3).  A wave's own MFMAs and VALU work do not overlap, other waves'
// do (tools/probes/overlap_probe.hip): the kernel lacks waves, not issue slots.  Here a wave owns 16 rows on
// v_mfma_f32_16x16x32_bf16: half the A slab (32 VGPRs), a quarter of the accumulators (16), half the epilogue per tile -- 128
// VGPRs, 13-16 waves per workgroup, 4 waves per SIMD.  A packed batch brings ~200 rows per CU = 13 slabs: one workgroup per CU,
// one round.  Same structure otherwise: A-stationary slab in registers (LayerNorm evaluated inside the slab load for the QKV /
// FC1 products, basic_transformer.py:165-173), 64-column weight tiles L2 -> LDS by global_load_lds, double-buffered, one
// workgroup barrier per tile, per-wave LDS transpose -> epilogue8 on 8 consecutive columns (gemm_epi.h).
#include <cstdlib>
#include "gemm_epi.h"

#define R16_K 256
#define R16_BN 64
#define R16_TILE_HALFS (R16_BN * R16_K)          // 32 KiB per buffer
#define R16_EPITCH (R16_BN + 4)
#define R16_EFLOATS (16 * R16_EPITCH)            // per-wave transpose region: 16 rows x 64 columns (+ pad)
#define R16_ROPE_FLOATS (16 * 16)                // per wave: [16 rows][8 cos | 8 sin]; EPI_MUL_AUX: the 16 x 64 one-byte codes
#define R16_MAXW 16
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// cur / nxt reach the tile body as __restrict__ parameters: the compiler otherwise waits for every pending global_load_lds
// before an LDS read it cannot prove disjoint from the DMA's target
template <typename F>
__device__ __forceinline__ void r16_call_restrict(F&& f, int jt, int jn, const bf16_t* __restrict__ cur, bf16_t* __restrict__ nxt) {
  f(jt, jn, cur, nxt);
}

template <int EPI, bool LN>
__global__ __launch_bounds__(64 * R16_MAXW, 1) void gemm_rb16_kernel(GemmArgs p, int W, int rot) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* const Bs = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* const Es = reinterpret_cast<float*>(smem + 2 * R16_TILE_HALFS * 2) + wave * R16_EFLOATS;
  float* const Rs = reinterpret_cast<float*>(smem + 2 * R16_TILE_HALFS * 2) + W * R16_EFLOATS + wave * R16_ROPE_FLOATS;
  const int m0 = (blockIdx.x * W + wave) * 16;
  const int fr = lane & 15, kq = lane >> 4;       // this lane's row of the slab, its 8-k group inside a 32-k step
  constexpr int CGS = R16_BN / 8;

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
      if (row_l < p.M) *reinterpret_cast<uint4*>(op + ks * 32) = u;
    }
    __syncthreads();   // everyone has read gamma / beta: the buffer may receive its weight tile
  }
  if constexpr (EPI == EPI_QKV_ROPE) {
    // rotary rows of this wave's 16 rows: lane -> (row = lane >> 2, quarter): 4 floats of [8 cos | 8 sin]
    const int r = lane >> 2, qd = lane & 3;
    const int mr = m0 + r < p.M ? m0 + r : p.M - 1;
    const int t = p.rope_row_t != nullptr ? p.rope_row_t[mr] : (m0 + r) % p.rope_T;
    const float* src = (qd < 2 ? p.rope_cos : p.rope_sin) + t * 16 + (qd & 1) * 4;
    *reinterpret_cast<float4*>(Rs + r * 16 + qd * 4) = *reinterpret_cast<const float4*>(src);
  }

  const int ntiles = (p.N + R16_BN - 1) / R16_BN;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;
  // weight tile: 32 pieces of 1 KiB (two 512-B rows), piece k by wave k % W; chunk c of row r at position c ^ (r & 31)
  auto load_tile = [&](int n0, bf16_t* S) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // W >= 8: 4 turns cover the 32 pieces
      const int k = wave + W * i;
      if (k < R16_BN / 2) {
        const int r = 2 * k + (lane >> 5), q = lane & 31;
        const int g = n0 + r, gc = g < p.N ? g : p.N - 1;
        const bf16_t* src = p.B + (long long)gc * p.ldb + ((q ^ (r & 31)) * 8);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(S + k * 512), 16, 0, 0);
      }
    }
  };
  // EPI_MUL_AUX: the 16 x 64 one-byte codes of this wave's output block = ONE 1-KiB DMA (lane -> row lane / 4, 16 columns),
  // issued before the MFMA phase of the tile, behind the same vmcnt(0) as the next weight tile; the epilogue task (row, 8-column
  // group) reads its 8 B back at row * 64 + 8 cg
  unsigned char* const Xs = reinterpret_cast<unsigned char*>(Rs);
  auto load_aux = [&](int n0) {
    const int row = m0 + (lane >> 2), col = n0 + (lane & 3) * 16;
    const int rc2 = row < p.M ? row : p.M - 1, cc = col + 16 <= p.N ? col : 0;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.aux_in) + (long long)rc2 * p.ld_aux + cc;
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)Xs, 16, 0, 0);
  };
  auto bias_at = [&](int col) { return p.bias[col < p.N ? col : p.N - 1]; };
  const bool has_bias = p.bias != nullptr;
  GemmArgs q = p;
  q.bias = nullptr;   // folded into the accumulator initialisation

  const int j0 = rot ? (int)(blockIdx.x % (unsigned)ntiles) : 0;
  load_tile(j0 * R16_BN, Bs);
  float bz[4], bn[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) { bz[a] = has_bias ? bias_at(j0 * R16_BN + 16 * a + fr) : 0.f; bn[a] = 0.f; }
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();

  auto tile = [&](int jt, int jn, const bf16_t* cur, bf16_t* nxt) {
    load_tile(jn * R16_BN, nxt);
    if constexpr (EPI == EPI_MUL_AUX) load_aux(jt * R16_BN);
    if (has_bias) {
#pragma unroll
      for (int a = 0; a < 4; ++a) bn[a] = bias_at(jn * R16_BN + 16 * a + fr);
    }
    f32x4_t acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][r] = bz[a];
    {
      // B fragment of column block a (16 weight rows), k step ks: lane (n = fr, kq) reads chunk 4 ks + kq of row 16 a + fr
      const bf16_t* wp = cur + fr * R16_K;
      const int sw = fr;   // (16 a is a multiple of 16: row & 31 = fr + 16 (a & 1))
      bf16x8 wf[2][4];
#pragma unroll
      for (int a = 0; a < 4; ++a) wf[0][a] = *reinterpret_cast<const bf16x8*>(wp + a * 16 * R16_K + (((kq) ^ (sw + 16 * (a & 1))) * 8));
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
#pragma unroll
          for (int a = 0; a < 4; ++a)
            wf[(ks + 1) & 1][a] = *reinterpret_cast<const bf16x8*>(wp + a * 16 * R16_K + (((4 * (ks + 1) + kq) ^ (sw + 16 * (a & 1))) * 8));
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], wf[ks & 1][a], acc[a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);   // the next tile has landed (this wave's pieces): wait BEFORE this tile's stores are issued
    // wave-private transpose: accumulator (lane = column fr of block a, register r = row 4 kq + r) -> rows of 64 contiguous columns
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) Es[(4 * kq + r) * R16_EPITCH + 16 * a + fr] = acc[a][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = lane + 64 * i, rl = t / CGS, cg = t % CGS;
      float v[8];
      const float4 c0 = *reinterpret_cast<const float4*>(Es + rl * R16_EPITCH + cg * 8);
      const float4 c1 = *reinterpret_cast<const float4*>(Es + rl * R16_EPITCH + cg * 8 + 4);
      v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
      const void* staged = nullptr;
      if constexpr (EPI == EPI_QKV_ROPE) staged = Rs + rl * 16;
      if constexpr (EPI == EPI_MUL_AUX) staged = Xs + rl * 64 + cg * 8;
      epilogue8<EPI, 1, R16_BN / 8, 16>(q, m0 + rl, jt * R16_BN + cg * 8, v, (m0 + rl) < p.M, jt, ntiles, staged);
    }
    __builtin_amdgcn_wave_barrier();      // the next writes to Es stay behind these reads
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) bz[a] = bn[a];
  };
  for (int j = 0, jt = j0; j < ntiles; ++j) {
    const int jn = jt + 1 == ntiles ? 0 : jt + 1;
    r16_call_restrict(tile, jt, jn, Bs + (j & 1) * R16_TILE_HALFS, Bs + ((j + 1) & 1) * R16_TILE_HALFS);
    jt = jn;
  }
}

// waves per workgroup for M rows: one round of one workgroup per CU
static int rb16_waves(int M) {
  const int slabs = (M + 15) / 16;
  return (slabs + 255) / 256;
}

bool gemm_rb16_supported(const GemmArgs& a, int a_f32, int epi) {
  static const bool off = getenv("COATI_NO_RB16") != nullptr;   // A/B switch: the 32-row kernel everywhere
  if (off || a_f32 || a.K != R16_K || a.m_dev != nullptr) return false;
  if (epi != EPI_BF16 && epi != EPI_QKV_ROPE && epi != EPI_GELU_GRAD && epi != EPI_MUL_AUX && epi != EPI_CE_PARTIAL && epi != EPI_CE_BWD) return false;
  if (epi == EPI_CE_PARTIAL && a.partial_tile != 64) return false;
  if (epi == EPI_QKV_ROPE && a.rope_hs == 32) return false;
  if (a.N % 16 != 0 && epi != EPI_CE_BWD && epi != EPI_CE_PARTIAL) return false;
  const int W = rb16_waves(a.M);
  return W >= 9 && W <= R16_MAXW;   // 36 865 .. 65 536 rows: below, the 32-row kernel or the tiled one; above, the 32-row kernel
}

template <int EPI, bool LN>
static int launch_rb16_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_rb16_kernel<EPI, LN>;
  constexpr size_t tile_bytes = (size_t)2 * R16_TILE_HALFS * 2;
  constexpr size_t per_wave = (size_t)R16_EFLOATS * 4 + (EPI == EPI_QKV_ROPE || EPI == EPI_MUL_AUX ? R16_ROPE_FLOATS * 4 : 0);
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(tile_bytes + R16_MAXW * per_wave)) != hipSuccess) {
      coati_set_error("gemm_rb16: hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int W = rb16_waves(a.M);
  const int blocks = cdiv(cdiv(a.M, 16), W);
  const int rot = (EPI == EPI_MUL_AUX);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), tile_bytes + W * per_wave, s, a, W, rot);
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
