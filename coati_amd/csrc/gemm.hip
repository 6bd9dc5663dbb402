// GEMM family for gfx950: bf16 MFMA (v_mfma_f32_32x32x16_bf16) with fp32 accumulation.
//
//   gemm_nt   C[M,N] = A[M,K] * B[N,K]^T  with fused epilogues        (all Linear forwards, dgrads)
//   wgrad_tn  dW[N,K] += A[M,N]^T * B[M,K] (+ column sums of A)       (all weight/bias gradients)
//   sgemm     exact-f32 MFMA (v_mfma_f32_32x32x2_f32), generic strides (contrastive head, [B,256] maths)
//
//   (every GEMM with K = 256 runs on the row-block kernel of gemm_rb.hip instead: launch_gemm_nt dispatches)
//
// Tiling: 128x128 output tile per workgroup (8 waves, each 32x64 = 1x2 MFMA tiles; 4 waves x 64x64 for f32 A),
// BK = 64, two LDS buffers + two register stage sets (one barrier per k-tile).  LDS rows are padded to
// 72 halfs (144 B): 16 consecutive rows then hit 16 distinct 16-B slots of the 256-B bank row, so the
// ds_read_b128 fragment reads are conflict-free without an XOR swizzle.  The accumulator tile is
// transposed through LDS so the epilogue works on 8 consecutive columns per lane (16-B stores).
#include <stdlib.h>
#include <algorithm>
#include <cstdlib>
#include "kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define PITCH 72       // halfs per LDS row
#define CPITCH 132     // floats per LDS row of the staged accumulator tile
#define TILE_HALFS (128 * PITCH)
#define GEMM_LDS_BYTES (4 * TILE_HALFS * 2)   // 73,728 B  (>= 128*132*4 = 67,584 B for the C stage)

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  // blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous run of tiles so
  // neighbouring tiles (same A row panel) share an L2.  Bijective for any nwg.
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- staging: global -> registers -> LDS --------------------------------------------------------
struct StageB16 { uint4 v[4]; };
struct StageF32 { float4 v[8]; };

template <int NT>
__device__ __forceinline__ void stage_load(StageB16& s, const bf16_t* base, long long ld, int row0, int rows,
                                           int k0, int tid) {
#pragma unroll
  for (int i = 0; i < 1024 / NT; ++i) {
    const int c = tid + NT * i, row = c >> 3, kc = (c & 7) * 8;
    const int g = row0 + row;
    const int gc = g < rows ? g : rows - 1;   // always a valid address; out-of-range rows are zeroed in stage_store
    s.v[i] = *reinterpret_cast<const uint4*>(base + (long long)gc * ld + k0 + kc);
  }
}
// The loaded registers are not touched between the load and this store (a select or mask right after the load makes the
// compiler wait for the load on the spot, or predicate it behind an exec branch): rows >= `rows` are zeroed here.
template <int NT>
__device__ __forceinline__ void stage_store(const StageB16& s, bf16_t* S, int row0, int rows, int tid) {
#pragma unroll
  for (int i = 0; i < 1024 / NT; ++i) {
    const int c = tid + NT * i, row = c >> 3, kc = (c & 7) * 8;
    const unsigned keep = (row0 + row) < rows ? 0xffffffffu : 0u;
    *reinterpret_cast<uint4*>(S + row * PITCH + kc) = make_uint4(s.v[i].x & keep, s.v[i].y & keep, s.v[i].z & keep, s.v[i].w & keep);
  }
}
template <int NT>
__device__ __forceinline__ void stage_load(StageF32& s, const float* base, long long ld, int row0, int rows,
                                           int k0, int tid) {
#pragma unroll
  for (int i = 0; i < 2048 / NT; ++i) {
    const int c = tid + NT * i, row = c >> 4, kc = (c & 15) * 4;
    const int g = row0 + row;
    const int gc = g < rows ? g : rows - 1;
    s.v[i] = *reinterpret_cast<const float4*>(base + (long long)gc * ld + k0 + kc);
  }
}
template <int NT>
__device__ __forceinline__ void stage_store(const StageF32& s, bf16_t* S, int row0, int rows, int tid) {
#pragma unroll
  for (int i = 0; i < 2048 / NT; ++i) {
    const int c = tid + NT * i, row = c >> 4, kc = (c & 15) * 4;
    const unsigned keep = (row0 + row) < rows ? 0xffffffffu : 0u;
    uint2 u;
    u.x = pack2bf(s.v[i].x, s.v[i].y) & keep;
    u.y = pack2bf(s.v[i].z, s.v[i].w) & keep;
    *reinterpret_cast<uint2*>(S + row * PITCH + kc) = u;
  }
}

// ---- MFMA over one staged k-tile: each wave owns a 64x64 sub-tile ----------------------------------
template <int MI>
__device__ __forceinline__ void mma_ktile(const bf16_t* As, const bf16_t* Bs, int wm, int wn, int lane,
                                          f32x16 (&acc)[MI][2]) {
  const int r = lane & 31, kg = (lane >> 5) * 8;
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    bf16x8 a[MI], b[2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
      a[i] = *reinterpret_cast<const bf16x8*>(As + (wm * 32 * MI + i * 32 + r) * PITCH + ks * 16 + kg);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      b[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + j * 32 + r) * PITCH + ks * 16 + kg);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

#include "gemm_epi.h"

// ---- NT GEMM kernel -------------------------------------------------------------------------------------
// WAVES = 4: each wave owns 64x64 of the 128x128 tile (2x2 MFMA tiles); WAVES = 8: 32x64 per wave (1x2).  The 8-wave
// form doubles the resident waves per CU (LDS still allows 2 workgroups) and halves the per-thread epilogue work.
template <typename AT, typename STAGE_A, int EPI, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void gemm_nt_kernel(GemmArgs p) {
  if (p.m_dev) p.M = *p.m_dev;   // data-dependent row count (<= the M the grid was sized for)
  constexpr int NT = 64 * WAVES, MI = 8 / WAVES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* const lds = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_m = wg / tiles_n, tile_n = wg - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (m0 >= p.M) return;   // (only with m_dev: the grid covers every row otherwise)
  const AT* A = reinterpret_cast<const AT*>(p.A);

  // the bias is folded into the accumulator initialisation (lane = output column): no bias loads in the epilogue
  f32x16 acc[MI][2];
  GemmArgs q = p;
  q.bias = nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    const float b = (p.bias != nullptr) ? p.bias[col < p.N ? col : p.N - 1] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
  }

  // Two LDS buffers + two register stage sets: the loads of k-tile kt+2 are issued before the MFMAs of k-tile kt and
  // are written to LDS one iteration later, so two k-tiles (64 KB per workgroup) are in flight.  All loads are
  // unconditional (k offsets past the end are clamped to the last k-tile and never stored): no branch, no vmcnt(0).
  const int nk = p.K / BK;
  auto kofs = [&](int kt) { return (kt < nk ? kt : nk - 1) * BK; };
  if constexpr (sizeof(AT) == 2) {
    STAGE_A sa0, sa1;
    StageB16 sb0, sb1;
    stage_load<NT>(sa0, A, p.lda, m0, p.M, 0, tid);
    stage_load<NT>(sb0, p.B, p.ldb, n0, p.N, 0, tid);
    stage_load<NT>(sa1, A, p.lda, m0, p.M, kofs(1), tid);
    stage_load<NT>(sb1, p.B, p.ldb, n0, p.N, kofs(1), tid);
    stage_store<NT>(sa0, lds, m0, p.M, tid);
    stage_store<NT>(sb0, lds + 2 * TILE_HALFS, n0, p.N, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      // even k-tile: LDS buffer 0; stage set 0 is free -> prefetch kt + 2; stage set 1 (kt + 1) goes to buffer 1
      stage_load<NT>(sa0, A, p.lda, m0, p.M, kofs(kt + 2), tid);
      stage_load<NT>(sb0, p.B, p.ldb, n0, p.N, kofs(kt + 2), tid);
      mma_ktile<MI>(lds, lds + 2 * TILE_HALFS, wm, wn, lane, acc);
      if (kt + 1 < nk) {
        stage_store<NT>(sa1, lds + TILE_HALFS, m0, p.M, tid);
        stage_store<NT>(sb1, lds + 3 * TILE_HALFS, n0, p.N, tid);
      }
      __syncthreads();
      if (kt + 1 >= nk) break;
      // odd k-tile: LDS buffer 1; prefetch kt + 3 into stage set 1; stage set 0 (kt + 2) goes to buffer 0
      stage_load<NT>(sa1, A, p.lda, m0, p.M, kofs(kt + 3), tid);
      stage_load<NT>(sb1, p.B, p.ldb, n0, p.N, kofs(kt + 3), tid);
      mma_ktile<MI>(lds + TILE_HALFS, lds + 3 * TILE_HALFS, wm, wn, lane, acc);
      if (kt + 2 < nk) {
        stage_store<NT>(sa0, lds, m0, p.M, tid);
        stage_store<NT>(sb0, lds + 2 * TILE_HALFS, n0, p.N, tid);
      }
      __syncthreads();
    }
  } else {
    // f32 A (twice the staging registers): one stage set, prefetch distance 1
    STAGE_A sa;
    StageB16 sb;
    stage_load<NT>(sa, A, p.lda, m0, p.M, 0, tid);
    stage_load<NT>(sb, p.B, p.ldb, n0, p.N, 0, tid);
    stage_store<NT>(sa, lds, m0, p.M, tid);
    stage_store<NT>(sb, lds + 2 * TILE_HALFS, n0, p.N, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      stage_load<NT>(sa, A, p.lda, m0, p.M, kofs(kt + 1), tid);
      stage_load<NT>(sb, p.B, p.ldb, n0, p.N, kofs(kt + 1), tid);
      mma_ktile<MI>(lds + cur * TILE_HALFS, lds + (2 + cur) * TILE_HALFS, wm, wn, lane, acc);
      if (kt + 1 < nk) {
        stage_store<NT>(sa, lds + (cur ^ 1) * TILE_HALFS, m0, p.M, tid);
        stage_store<NT>(sb, lds + (2 + (cur ^ 1)) * TILE_HALFS, n0, p.N, tid);
      }
      __syncthreads();
    }
  }

  // accumulators -> LDS (fp32) -> row-contiguous epilogue
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(wm * 32 * MI + i * 32 + frag_row(r, lane)) * CPITCH + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
  __syncthreads();
  // extra row-major operands of the epilogue (f32 residual / accumulate target, saved pre-activations): all tasks' loads
  // are issued first, so their latency is paid once per workgroup instead of once per task
  constexpr int TASKS = 2048 / NT;
  constexpr bool PRE_F32 = (EPI == EPI_RES_F32 || EPI == EPI_ACC_F32);
  constexpr bool PRE_B16 = (EPI == EPI_DGELU || EPI == EPI_DSILU || EPI == EPI_MUL_AUX);
  EpiPre pre[TASKS];
#pragma unroll
  for (int i = 0; i < TASKS; ++i) {
    const int task = tid + NT * i, r = task >> 4, cg = task & 15;
    const int row = m0 + r, col0 = n0 + cg * 8;
    pre[i].have = (PRE_F32 || PRE_B16) && row < p.M && col0 + 8 <= p.N;
    if (pre[i].have) {
      if constexpr (PRE_F32) {
        const float* src = (EPI == EPI_RES_F32) ? reinterpret_cast<const float*>(p.aux_in) + (long long)row * p.ld_aux + col0
                                                : reinterpret_cast<const float*>(p.C) + (long long)row * p.ldc + col0;
        pre[i].f0 = *reinterpret_cast<const float4*>(src);
        pre[i].f1 = *reinterpret_cast<const float4*>(src + 4);
      }
      if constexpr (EPI == EPI_MUL_AUX) {   // 8 one-byte codes
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(p.aux_in) + (long long)row * p.ld_aux + col0);
        pre[i].h = make_uint4(u.x, u.y, 0, 0);
      } else if constexpr (PRE_B16) {
        pre[i].h = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.aux_in) + (long long)row * p.ld_aux + col0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TASKS; ++i) {
    const int task = tid + NT * i, r = task >> 4, cg = task & 15;
    float v[8];
    const float4 c0 = *reinterpret_cast<const float4*>(Cs + r * CPITCH + cg * 8);
    const float4 c1 = *reinterpret_cast<const float4*>(Cs + r * CPITCH + cg * 8 + 4);
    v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
    epilogue8<EPI>(q, m0 + r, n0 + cg * 8, v, (m0 + r) < p.M, tile_n, tiles_n, nullptr, &pre[i]);
  }
}

template <typename AT, typename STAGE_A, int EPI, int WAVES>
static int launch_nt_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_nt_kernel<AT, STAGE_A, EPI, WAVES>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       GEMM_LDS_BYTES);
    if (e != hipSuccess) {
      coati_set_error("gemm_nt: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WAVES), GEMM_LDS_BYTES, s, a);
  COATI_LAUNCH_CHECK("gemm_nt");
  return COATI_OK;
}

int launch_gemm_nt(const GemmArgs& a, int a_f32, int epi, hipStream_t s) {
  COATI_CHECK_ARG(a.A && a.B, "gemm_nt: null operand");
  COATI_CHECK_ARG(a.C || epi == EPI_CE_PARTIAL, "gemm_nt: null output");
  COATI_CHECK_SHAPE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % BK == 0, "gemm_nt: K=%d must be a positive multiple of %d", a.K, BK);
  COATI_CHECK_SHAPE(a.lda % (a_f32 ? 4 : 8) == 0 && a.ldb % 8 == 0, "gemm_nt: lda/ldb alignment (lda=%lld ldb=%lld)", a.lda, a.ldb);
  const bool out_f32 = (epi == EPI_F32 || epi == EPI_RES_F32 || epi == EPI_ACC_F32 || epi == EPI_LNBWD);
  if (epi == EPI_LNBWD) {   // the ring GEMM's fused LayerNorm backward: no other kernel has it (callers ask gemm_ring_lnbwd_supported first)
    COATI_CHECK_ARG(!a_f32, "gemm_nt: EPI_LNBWD takes a bf16 A operand");
    return launch_gemm_ring256(a, epi, s);
  }
  if (epi != EPI_CE_PARTIAL)
    COATI_CHECK_SHAPE(a.ldc % (out_f32 ? 4 : 8) == 0, "gemm_nt: ldc=%lld alignment", a.ldc);
  COATI_CHECK_SHAPE(((long long)a.M + 128) * a.ldc < (1LL << 32) && ((long long)a.M + 128) * a.ld_aux < (1LL << 32),
                    "gemm_nt: output / aux rows do not fit 32-bit element offsets (M=%d ldc=%lld ld_aux=%lld)", a.M, a.ldc, a.ld_aux);
  if (epi == EPI_RES_F32) COATI_CHECK_ARG(a.aux_in && a.ld_aux % 4 == 0, "gemm_nt: residual missing/misaligned");
  if (epi == EPI_GELU || epi == EPI_SILU || epi == EPI_GELU_GRAD) COATI_CHECK_ARG(a.aux_out && a.ld_aux % 8 == 0, "gemm_nt: aux_out missing");
  if (epi == EPI_DGELU || epi == EPI_DSILU || epi == EPI_MUL_AUX) COATI_CHECK_ARG(a.aux_in && a.ld_aux % 8 == 0, "gemm_nt: aux_in missing");
  if (epi == EPI_CE_PARTIAL) COATI_CHECK_ARG(a.partial, "gemm_nt: partial buffer missing");
  if (epi == EPI_CE_BWD) COATI_CHECK_ARG(a.lse && a.target && a.scal, "gemm_nt: CE operands missing");
  if (epi == EPI_QKV_ROPE) COATI_CHECK_ARG(a.rope_cos && a.rope_sin && a.rope_T > 0 && a.rope_C > 0 && a.N % 16 == 0 && a.rope_C % 16 == 0 && (a.rope_hs == 0 || a.rope_hs == 16 || a.rope_hs == 32) && (a.rope_hs != 32 || a.rope_C % 32 == 0), "gemm_nt: rope operands missing / unsupported head size");
  if (epi == EPI_EDGE_DPRE) COATI_CHECK_ARG(a.P && a.d2 && a.w1c && a.b1 && a.natom > 0 && a.ldp % 8 == 0, "gemm_nt: edge operands missing");
  {
    // N = 256 with a bf16 / residual epilogue: the ring kernel, also at K = 256 (proj forward 57 -> 44 us against the
    // row-block kernel); every other K = 256 product: the row-block kernel
    if (gemm_rb16_resident_supported(a, a_f32, epi)) return launch_gemm_rb16_resident(a, epi, s);
    if (gemm_ring256_supported(a, a_f32, epi)) return launch_gemm_ring256(a, epi, s);
#ifdef COATI_EXPERIMENTAL
    if (gemm_t32_supported(a, a_f32, epi)) return launch_gemm_t32(a, epi, s);
#endif
    if (gemm_rb16_supported(a, a_f32, epi)) return launch_gemm_rb16(a, epi, s);
    if (gemm_rb256_supported(a, a_f32, epi)) return launch_gemm_rb256(a, epi, s);
  }
#define NT_CASE(E)                                                                  \
  case E:                                                                           \
    return a_f32 ? launch_nt_t<float, StageF32, E, 4>(a, s) : launch_nt_t<bf16_t, StageB16, E, 8>(a, s);
#define NT_CASE_B16(E)                                                              \
  case E:                                                                           \
    COATI_CHECK_ARG(!a_f32, "gemm_nt: epilogue %d has no f32-A variant", (int)E);   \
    return launch_nt_t<bf16_t, StageB16, E, 8>(a, s);
  switch (epi) {
    NT_CASE(EPI_BF16)
    NT_CASE(EPI_F32)
    NT_CASE_B16(EPI_RES_F32)
    NT_CASE_B16(EPI_GELU)
    NT_CASE(EPI_DGELU)
    NT_CASE_B16(EPI_SILU)
    NT_CASE(EPI_DSILU)
    NT_CASE_B16(EPI_ACC_F32)
    NT_CASE_B16(EPI_CE_PARTIAL)
    NT_CASE_B16(EPI_CE_BWD)
    NT_CASE_B16(EPI_EDGE_DPRE)
    NT_CASE_B16(EPI_QKV_ROPE)
    NT_CASE_B16(EPI_GELU_GRAD)
    NT_CASE_B16(EPI_MUL_AUX)
    default:
      coati_set_error("gemm_nt: unknown epilogue %d", epi);
      return COATI_EARG;
  }
#undef NT_CASE
#undef NT_CASE_B16
}

// =================================================================================================
// wgrad: dW[N,K] += A[M,N]^T * B[M,K]
// The reduction runs over m, the strided dimension of both operands.  Tiles are staged ROW-MAJOR ([64 m][128 cols],
// coalesced 16-B loads and ds_write_b128) and the MFMA fragments are fetched with the gfx950 LDS transpose read
// ds_read_b64_tr_b16: each 16-lane group passes the addresses of a 4(m) x 16(col) block and every lane receives
// the 4 m-values of its own column (probed on hardware: tools/probes/tr_probe.hip).  Two such reads give the
// 8 m-consecutive bf16 a 32x32x16 fragment needs -- no register transposes, no scalar LDS traffic.
// LDS rows are exactly 256 B; the 64-B chunk index is XOR-swizzled with (row & 3) so the four rows of a
// transpose read fall on different bank windows.  Two register stage sets keep two m-chunks of loads in flight.
// =================================================================================================
typedef short v4s16 __attribute__((ext_vector_type(4)));
#define WT_ROW_BYTES 256
#define WT_TILE_BYTES (64 * WT_ROW_BYTES)          // one [64 m][128 col] bf16 tile
#define WGRAD_LDS_BYTES (4 * WT_TILE_BYTES)        // 2 operands x 2 stages = 64 KiB

struct StageW { uint4 v[4]; };   // 4 x (one m-row, 8 columns)

__device__ __forceinline__ unsigned wt_offset(int row, int bytecol) {
  return (unsigned)(row * WT_ROW_BYTES + ((((bytecol >> 6) ^ (row & 3)) << 6) | (bytecol & 63)));
}

template <typename AT, int NT>
__device__ __forceinline__ void wstage_load(StageW& s, const AT* base, long long ld, int m0, int m_end, int c0,
                                            int ncols, int tid) {
  const int col = c0 + (tid & 15) * 8;
  const bool colok = col < ncols;
  const int colc = colok ? col : 0;
#pragma unroll
  for (int i = 0; i < 1024 / NT; ++i) {
    const int m = m0 + (tid >> 4) + (NT / 16) * i;
    const int mc = m < m_end ? m : m_end - 1;
    uint4 t;
    if constexpr (sizeof(AT) == 2) {
      t = *reinterpret_cast<const uint4*>(base + (long long)mc * ld + colc);
    } else {
      const float4 f0 = *reinterpret_cast<const float4*>(base + (long long)mc * ld + colc);
      const float4 f1 = *reinterpret_cast<const float4*>(base + (long long)mc * ld + colc + 4);
      t = make_uint4(pack2bf(f0.x, f0.y), pack2bf(f0.z, f0.w), pack2bf(f1.x, f1.y), pack2bf(f1.z, f1.w));
    }
    // zero-fill by masking, not by select: a select lets the compiler predicate the load itself (one exec branch per load)
    const unsigned keep = (m < m_end && colok) ? 0xffffffffu : 0u;
    s.v[i] = make_uint4(t.x & keep, t.y & keep, t.z & keep, t.w & keep);
  }
}
template <int NT>
__device__ __forceinline__ void wstage_store(const StageW& s, unsigned char* T, int tid) {
#pragma unroll
  for (int i = 0; i < 1024 / NT; ++i) {
    const int row = (tid >> 4) + (NT / 16) * i;
    *reinterpret_cast<uint4*>(T + wt_offset(row, (tid & 15) * 16)) = s.v[i];
  }
}
template <int NT>
__device__ __forceinline__ void wstage_colsum(const StageW& s, float (&csum)[8]) {
#pragma unroll
  for (int i = 0; i < 1024 / NT; ++i) {
    float f[8];
    unpack8(s.v[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] += f[e];
  }
}

// fragment for output rows/cols nbase..nbase+31 and reduction rows mbase..mbase+15 of a row-major tile
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* T, int mbase, int nbase, int lane) {
  const int g = lane >> 4, kg = g >> 1;
  const int row = mbase + 8 * kg + ((lane & 15) >> 2);
  const int bytecol = (nbase + 16 * (g & 1) + 4 * (lane & 3)) * 2;
  typedef __attribute__((address_space(3))) v4s16 lds_v4;
  const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(T + wt_offset(row, bytecol)));
  const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(T + wt_offset(row + 4, bytecol)));
  typedef short v8s16 __attribute__((ext_vector_type(8)));
  const v8s16 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <int MI>
__device__ __forceinline__ void wmma_chunk(const unsigned char* At, const unsigned char* Bt, int wm, int wn, int lane,
                                           f32x16 (&acc)[MI][2]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bf16x8 a[MI], b[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) a[i] = tr_frag(At, ks * 16, wm * 32 * MI + i * 32, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = tr_frag(Bt, ks * 16, wn * 64 + j * 32, lane);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

template <typename AT, bool BIAS, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void wgrad_kernel(WgradArgs p, int tiles_k, int chunks_per_split, int n_splits) {
  constexpr int NT = 64 * WAVES, MI = 8 / WAVES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // 1-D grid, XCD-aware: all output tiles of one M-split run back to back on the same XCD, so the split's A / B slabs
  // are fetched from HBM once and re-read by the other tiles out of that XCD's L2.
  const int tiles = gridDim.x / n_splits;
  const int wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int split = wg / tiles, tile = wg - split * tiles;
  const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
  const int n0 = tile_n * BM, k0 = tile_k * BN;
  if (p.m_dev) {   // data-dependent row count: the splits share the rows that exist
    p.M = *p.m_dev;
    chunks_per_split = ((p.M + 63) / 64 + n_splits - 1) / n_splits;
  }
  const int nchunks = (p.M + 63) / 64;
  const int c_begin = split * chunks_per_split;
  int c_end = c_begin + chunks_per_split;
  if (c_end > nchunks) c_end = nchunks;
  if (c_begin >= c_end) return;
  const AT* A = reinterpret_cast<const AT*>(p.A);
  const int nout = p.n_out > 0 ? p.n_out : p.N;
  unsigned char* const At0 = smem;
  unsigned char* const At1 = smem + WT_TILE_BYTES;
  unsigned char* const Bt0 = smem + 2 * WT_TILE_BYTES;
  unsigned char* const Bt1 = smem + 3 * WT_TILE_BYTES;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  const bool do_bias = BIAS && (tile_k == 0);

  StageW sa0, sb0, sa1, sb1;
  wstage_load<AT, NT>(sa0, A, p.lda, c_begin * 64, p.M, n0, p.N, tid);
  wstage_load<bf16_t, NT>(sb0, p.B, p.ldb, c_begin * 64, p.M, k0, p.K, tid);
  // every prefetch below is issued UNCONDITIONALLY (rows past the end are clamped and zero-filled by wstage_load): a
  // branch around the loads would make the compiler wait for vmcnt(0) at the join and kill the two-deep prefetch
  wstage_load<AT, NT>(sa1, A, p.lda, (c_begin + 1) * 64, p.M, n0, p.N, tid);
  wstage_load<bf16_t, NT>(sb1, p.B, p.ldb, (c_begin + 1) * 64, p.M, k0, p.K, tid);
  if (do_bias) wstage_colsum<NT>(sa0, csum);
  wstage_store<NT>(sa0, At0, tid);
  wstage_store<NT>(sb0, Bt0, tid);
  __syncthreads();
  for (int c = c_begin; c < c_end; c += 2) {
    wstage_load<AT, NT>(sa0, A, p.lda, (c + 2) * 64, p.M, n0, p.N, tid);
    wstage_load<bf16_t, NT>(sb0, p.B, p.ldb, (c + 2) * 64, p.M, k0, p.K, tid);
    wmma_chunk<MI>(At0, Bt0, wm, wn, lane, acc);
    if (c + 1 < c_end) {
      if (do_bias) wstage_colsum<NT>(sa1, csum);
      wstage_store<NT>(sa1, At1, tid);
      wstage_store<NT>(sb1, Bt1, tid);
    }
    __syncthreads();
    if (c + 1 >= c_end) break;
    wstage_load<AT, NT>(sa1, A, p.lda, (c + 3) * 64, p.M, n0, p.N, tid);
    wstage_load<bf16_t, NT>(sb1, p.B, p.ldb, (c + 3) * 64, p.M, k0, p.K, tid);
    wmma_chunk<MI>(At1, Bt1, wm, wn, lane, acc);
    if (c + 2 < c_end) {
      if (do_bias) wstage_colsum<NT>(sa0, csum);
      wstage_store<NT>(sa0, At0, tid);
      wstage_store<NT>(sb0, Bt0, tid);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 32 * MI + i * 32 + frag_row(r, lane);
        const int k = k0 + wn * 64 + j * 32 + (lane & 31);
        if (n < nout && k < p.K) atomicAdd(p.dW + (long long)n * p.ldw + k, acc[i][j][r]);
      }

  if (do_bias) {
    // reduce the 16 row-groups that share a column octet, then one atomic per column
    float* red = reinterpret_cast<float*>(smem);  // [NT/16][128]
    const int noct = tid & 15, mq = tid >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[mq * 128 + noct * 8 + e] = csum[e];
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < NT / 16; ++q) t += red[q * 128 + tid];
      if (n0 + tid < nout) atomicAdd(p.dbias + n0 + tid, t);
    }
  }
}

// =================================================================================================
// wgrad, LDS-DMA variant (bf16 A).  One 512-thread workgroup per CU owns one 128 x 128 output tile and one slice of M:
//   * operand rows go HBM/L2 -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write) into a ring of NS
//     stages of 64 m-rows ([2 sub-chunks][A: 32 m x 128 n | B: 32 m x 128 k], 32 KiB).  One wave instruction fills
//     1 KiB = 4 tile rows in lane order; the 64-B-chunk swizzle of the transpose reads is applied on the GLOBAL side
//     (the lane that writes slot q of a row fetches chunk (q>>2 ^ row&3, q&3)).  NS - 1 stages are always in flight:
//     per stage one s_waitcnt vmcnt(4 (NS - 2)) + one s_barrier (no fence: a fence would drain every DMA);
//   * the 8 waves form two groups of 2 x 2 waves; group g multiplies sub-chunk g of every stage into its own copy of
//     the tile, and at the end the groups swap halves through the (then idle) ring so that each wave adds up and
//     commits half of its 64 x 64 block.  Two groups per tile instead of two workgroups per CU halve the fp32 atomics,
//     the largest single cost of the register-staged kernel (ablation: streaming 30 us, MFMA phase +12, atomics +17 of
//     62 us at 768 x 256);
//   * rows past the end of the slice and columns past N / K are fetched from a 256-B page of zeros: no masking anywhere;
//   * bias column sums are read back from the A sub-chunk in LDS (8 ds_read_b32 per thread) and spread evenly over the
//     tiles_k workgroups that see the same A rows: workgroup tile_k takes the stages with stage % tiles_k == tile_k.
// The DMA is issued from inline assembly so that the only vmcnt waits in the ring are the hand-placed ones (with the
// builtin the compiler puts a vmcnt(0) in front of every ds_read_b64_tr_b16: its memory operand carries no alias
// scope that would separate it from the DMA targets).
// =================================================================================================
#define WD_CH 64                                 // m rows per stage (two sub-chunks of 32)
#define WD_HALF_BYTES (32 * WT_ROW_BYTES)        // one operand of one sub-chunk: 8 KiB
#define WD_STAGE_BYTES (4 * WD_HALF_BYTES)       // 32 KiB
struct WdFrags { bf16x8 a[2][2], b[2][2]; };
__device__ __attribute__((aligned(256))) const uint4 wd_zero_page[16] = {};

// One global_load_lds_dwordx4: lane i -> LDS byte address lds + 16 i (lds wave-uniform).  Per-lane 64-bit address ...
__device__ __forceinline__ void wd_dma16(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
// ... or scalar base + 32-bit per-lane byte offset
__device__ __forceinline__ void wd_dma16s(const void* base, unsigned off, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
template <int S> struct WdSlot { static constexpr int value = S; };

// EXCL: this workgroup is the only one that writes its output tile (it streams ALL of M): the tile is committed with
// plain read-add-stores instead of fp32 atomics (the grouped launch below); the bias partials stay atomic (tiles_k
// workgroups share a bias slice).
template <bool BIAS, int NS, bool EXCL>
__device__ __forceinline__ void wgrad_dma_body(const WgradArgs& p, int tiles_k, int tile, int c_begin, int c_end, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
  const int n0 = tile_n * BM, k0 = tile_k * BN;
  if (c_begin >= c_end) return;
  const int m_begin = c_begin * WD_CH;
  const int m_end = c_end * WD_CH < p.M ? c_end * WD_CH : p.M;
  const int nout = p.n_out > 0 ? p.n_out : p.N;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);

  // DMA sources of this lane.  Piece (1 KiB = 4 rows) wave + 8 i of a stage, i = 0..3: sub-chunk i >> 1, operand i & 1,
  // rows 4 wave .. +3; LDS slot (row & 3 = lane >> 4, 16-B chunk q = lane & 15) <- global chunk qg of that row.
  const int rsub = lane >> 4, q = lane & 15;
  const int qg = (((q >> 2) ^ rsub) << 2) | (q & 3);
  const bool aok = n0 + qg * 8 < p.N, bok = k0 + qg * 8 < p.K;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(wd_zero_page) + q * 8;
  const bool full_tile = n0 + BM <= p.N && k0 + BN <= p.K && 130LL * p.lda < (1LL << 30) && 130LL * p.ldb < (1LL << 30);
  const unsigned offa = (unsigned)(((4 * wave + rsub) * (int)p.lda + qg * 8) * 2);   // byte offsets of this lane's row 0 chunk
  const unsigned offb = (unsigned)(((4 * wave + rsub) * (int)p.ldb + qg * 8) * 2);
  int cnext = c_begin;   // next stage to fetch

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }
  }
  // bias: thread -> columns 2 cp, 2 cp + 1 and rows 8 (wave & 3) .. +7 of its group's sub-chunk
  float cs0 = 0.f, cs1 = 0.f;
  const int cp = lane, rq = wave & 3;
  int bias_ctr = (tile_k - c_begin % tiles_k + tiles_k) % tiles_k;   // stages until this workgroup's turn

  // DMA of the next stage -> ring slot `slot` (this wave's 4 pieces).  Interior stages of full tiles take the fast path:
  // scalar bases (advanced on the scalar unit) + constant per-lane byte offsets, no vector work at all; stages that touch
  // the end of the slice and tiles that stick out of N / K select per lane between the row and the zero page.
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u8*)smem) + wave * 1024;
  const bf16_t* ab = A + n0 + (long long)c_begin * WD_CH * p.lda;     // row 0 of the next stage (scalar)
  const bf16_t* bb = p.B + k0 + (long long)c_begin * WD_CH * p.ldb;
  const long long ab_step = (long long)WD_CH * p.lda, bb_step = (long long)WD_CH * p.ldb;
  const unsigned offa1 = offa + 64u * (unsigned)p.lda, offb1 = offb + 64u * (unsigned)p.ldb;
  auto issue = [&](int slot) __attribute__((always_inline)) {
    const unsigned S = ldsw + slot * WD_STAGE_BYTES;
    if (full_tile && (cnext + 1) * WD_CH <= m_end) {
      wd_dma16s(ab, offa, S);
      wd_dma16s(bb, offb, S + WD_HALF_BYTES);
      wd_dma16s(ab, offa1, S + 2 * WD_HALF_BYTES);
      wd_dma16s(bb, offb1, S + 3 * WD_HALF_BYTES);
    } else {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int m = cnext * WD_CH + 32 * sub + 4 * wave + rsub;
        const bool ok = m < m_end;
        wd_dma16((ok && aok) ? A + n0 + qg * 8 + (long long)m * p.lda : zero, S + (2 * sub) * WD_HALF_BYTES);
        wd_dma16((ok && bok) ? p.B + k0 + qg * 8 + (long long)m * p.ldb : zero, S + (2 * sub + 1) * WD_HALF_BYTES);
      }
    }
    ++cnext;
    ab += ab_step;
    bb += bb_step;
  };
  // fragments of the stage in ring slot `slot` (compile-time in the main loop): every transpose read is one of four
  // per-lane LDS addresses (2 A blocks, 2 B blocks of this wave, incl. the group's sub-chunk) + an immediate
  typedef __attribute__((address_space(3))) v4s16 lds_v4;
  unsigned la[2], lb[2];
  {
    const int g = lane >> 4, kg = g >> 1, r0 = 8 * kg + ((lane & 15) >> 2), cb = (16 * (g & 1) + 4 * (lane & 3)) * 2;
    const unsigned sub0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u8*)smem) + grp * 2 * WD_HALF_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      la[i] = sub0 + wt_offset(r0, (wm * 64 + i * 32) * 2 + cb);
      lb[i] = sub0 + WD_HALF_BYTES + wt_offset(r0, (wn * 64 + i * 32) * 2 + cb);
    }
  }
  auto frag_at = [&](unsigned addr) __attribute__((always_inline)) {
    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(size_t)addr);
    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(size_t)(addr + 4 * WT_ROW_BYTES));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    const v8s16 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  };
  // a second set 64 KiB up (opaque to the compiler) keeps the immediates of ring slots >= 2 inside the 16-bit offset field
  unsigned lah[2], lbh[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    lah[i] = la[i] + 0x10000u;
    lbh[i] = lb[i] + 0x10000u;
    asm volatile("" : "+v"(lah[i]), "+v"(lbh[i]));
  }
  auto fetch = [&](int slot, WdFrags& f) __attribute__((always_inline)) {
    const bool high = slot * WD_STAGE_BYTES >= 0x10000;
    const unsigned so = slot * WD_STAGE_BYTES - (high ? 0x10000u : 0u);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) f.a[ks][i] = frag_at((high ? lah[i] : la[i]) + so + ks * 16 * WT_ROW_BYTES);
#pragma unroll
      for (int j = 0; j < 2; ++j) f.b[ks][j] = frag_at((high ? lbh[j] : lb[j]) + so + ks * 16 * WT_ROW_BYTES);
    }
  };
  auto mma = [&](const WdFrags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][i], f.b[ks][j], acc[i][j], 0, 0, 0);
  };
  // bias operands of this thread in the stage in ring slot `slot`: rows 8 rq .. +7, columns 2 cp, 2 cp + 1
  typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
  // (1, 0) and (0, 1) selectors of v_dot2c_f32_bf16, kept in registers: as immediates the compiler turns both into the
  // inline constant 1.0, which does not mean the same thing for a packed operand
  unsigned sel_lo_u, sel_hi_u;
  asm volatile("v_mov_b32 %0, 0x3f80\n\tv_mov_b32 %1, 0x3f800000" : "=v"(sel_lo_u), "=v"(sel_hi_u));
  const v2bf sel_lo = __builtin_bit_cast(v2bf, sel_lo_u), sel_hi = __builtin_bit_cast(v2bf, sel_hi_u);
  unsigned bu[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) bu[rr] = 0;
  auto bias_read = [&](int slot) __attribute__((always_inline)) {
    const unsigned char* At = smem + slot * WD_STAGE_BYTES + grp * 2 * WD_HALF_BYTES;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) bu[rr] = *reinterpret_cast<const unsigned*>(At + wt_offset(rq * 8 + rr, cp * 4));
  };
  auto bias_add = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const v2bf u = __builtin_bit_cast(v2bf, bu[rr]);
      cs0 = __builtin_amdgcn_fdot2_f32_bf16(u, sel_lo, cs0, false);
      cs1 = __builtin_amdgcn_fdot2_f32_bf16(u, sel_hi, cs1, false);
    }
  };
  constexpr int VM_KEEP = 4 * (NS - 2);   // DMAs that may stay in flight when the oldest stage is needed
  constexpr int VM_WAIT = 0x0f70 | (VM_KEEP & 15) | ((VM_KEEP >> 4) << 14);

  // Stage c is multiplied out of registers while the transpose reads of stage c + 1 are in flight: one step =
  // [stage c + 1 has landed | barrier] -> DMA of stage c + NS into the slot of stage c (every wave finished reading it
  // before the barrier) -> LDS reads of stage c + 1 interleaved with the MFMAs of stage c.  NS - 1 stages are in flight
  // throughout.  The ring slot is a compile-time constant of each step (the loop is unrolled over slots x fragment
  // sets), so a step is ~55 instructions per wave: with two waves per SIMD the instruction count per stage is on the
  // critical path (every instruction shaved off the step showed up in the kernel time).
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0);
  __builtin_amdgcn_s_waitcnt(VM_WAIT);
  __builtin_amdgcn_s_barrier();
  issue(NS - 1);
  WdFrags fa, fb;
  fetch(0, fa);
  if (BIAS) {   // bias operands of the first stage
    if (bias_ctr == 0) {
      bias_read(0);
      bias_add();
      bias_ctr = tiles_k;
    }
    --bias_ctr;
  }
  auto step = [&](auto slot_c, const WdFrags& cur, WdFrags& nxt) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value, SN = (SL + 1) % NS;   // ring slots of stage c and stage c + 1
    __builtin_amdgcn_s_waitcnt(VM_WAIT & 0xf0ff);   // + lgkmcnt(0): this wave's reads of stage c are complete
    __builtin_amdgcn_s_barrier();
    issue(SL);
    // issue order: one MFMA (32 cycles in the matrix core, 4 to issue), then two of the transpose reads in its shadow --
    // the matrix core never waits for the read phase of the next stage.  On this workgroup's bias turns the 8 x 2 bias
    // operands of the thread are read in front of that block and added up behind it, when they have long arrived (the
    // interleave is per basic block, so the block itself stays free of branches).
    const bool turn = BIAS && bias_ctr == 0;
    if (BIAS) bias_ctr = (bias_ctr == 0 ? tiles_k : bias_ctr) - 1;
    if (turn) bias_read(SN);
    fetch(SN, nxt);
    mma(cur);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
    }
    __builtin_amdgcn_sched_barrier(0);
    if (turn) bias_add();
  };
  static_assert(NS == 3 || NS == 4, "the main loop is unrolled for ring depths 3 and 4");
  for (int c = c_begin;;) {
    if constexpr (NS == 3) {
      step(WdSlot<0>(), fa, fb); if (++c >= c_end) break;
      step(WdSlot<1>(), fb, fa); if (++c >= c_end) break;
      step(WdSlot<2>(), fa, fb); if (++c >= c_end) break;
      step(WdSlot<0>(), fb, fa); if (++c >= c_end) break;
      step(WdSlot<1>(), fa, fb); if (++c >= c_end) break;
      step(WdSlot<2>(), fb, fa); if (++c >= c_end) break;
    } else {
      step(WdSlot<0>(), fa, fb); if (++c >= c_end) break;
      step(WdSlot<1>(), fb, fa); if (++c >= c_end) break;
      step(WdSlot<2>(), fa, fb); if (++c >= c_end) break;
      step(WdSlot<3>(), fb, fa); if (++c >= c_end) break;
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);   // the over-issued (zero-page) DMAs have landed: the ring is free
  __builtin_amdgcn_s_barrier();

  // the groups swap halves: wave w keeps block row (i == grp) of its 64 x 64 block and hands the other one to wave w ^ 4
  float* const xch = reinterpret_cast<float*>(smem);
  float* const bsum = xch + 8 * 2048;   // [8 waves][128 columns] bias partials (2 sub-chunks x 4 row quarters), behind the exchange buffer
  {
    float* mine = xch + wave * 2048 + lane;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[(j * 16 + r) * 64] = grp == 0 ? acc[1][j][r] : acc[0][j][r];
    if (BIAS) {
      bsum[wave * 128 + 2 * cp] = cs0;
      bsum[wave * 128 + 2 * cp + 1] = cs1;
    }
  }
  __syncthreads();
  // the bias atomics go first: all workgroups of a tile row hit the same 128 addresses at the same time, and that
  // serialised chain then runs in the L2 underneath the (much larger) tile commit below
  if (BIAS && tid < 128 && p.dbias != nullptr) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += bsum[w * 128 + tid];
    if (n0 + tid < nout) atomicAdd(p.dbias + n0 + tid, t);
  }
  {
    const float* theirs = xch + (wave ^ 4) * 2048 + lane;
    const int i = grp;
    float v[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[j][r] = (grp == 0 ? acc[0][j][r] : acc[1][j][r]) + theirs[(j * 16 + r) * 64];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 64 + i * 32 + frag_row(r, lane);
        const int k = k0 + wn * 64 + j * 32 + (lane & 31);
        if (n < nout && k < p.K) {
          if constexpr (EXCL) p.dW[(long long)n * p.ldw + k] += v[j][r];
          else atomicAdd(p.dW + (long long)n * p.ldw + k, v[j][r]);
        }
      }
  }
}

template <bool BIAS, int NS>
__global__ __launch_bounds__(512, 1) void wgrad_dma_kernel(WgradArgs p, int tiles_k, int chunks_per_split, int n_splits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (p.m_dev) {   // data-dependent row count: the splits share the rows that exist
    p.M = *p.m_dev;
    chunks_per_split = ((p.M + WD_CH - 1) / WD_CH + n_splits - 1) / n_splits;
  }
  const int tiles = gridDim.x / n_splits;
  const int wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int split = wg / tiles, tile = wg - split * tiles;
  const int nchunks = (p.M + WD_CH - 1) / WD_CH;
  const int c_begin = split * chunks_per_split;
  int c_end = c_begin + chunks_per_split;
  if (c_end > nchunks) c_end = nchunks;
  wgrad_dma_body<BIAS, NS, false>(p, tiles_k, tile, c_begin, c_end, smem);
}

// Grouped launch: one workgroup per OUTPUT TILE of a whole list of weight-gradient problems (all four Linear layers of
// every transformer layer of a pass: 16 x 48 = 768 tiles = three rounds of 256 CUs).  Each workgroup streams all of M for
// its tile, so nothing is split over M: no fp32 atomics on the tiles, no per-problem launch tail.  The table (one entry
// per workgroup, device memory) is ordered problem by problem; xcd_swizzle hands each XCD a contiguous run of entries, so
// the tiles that share an operand panel run at the same time on the same L2.
template <bool BIAS, int NS>
__global__ __launch_bounds__(512, 1) void wgrad_dma_table_kernel(const WgradTile* __restrict__ table, int M_rt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WgradTile& d = table[xcd_swizzle(blockIdx.x, gridDim.x)];
  WgradArgs p = d.p;
  if (M_rt > 0) p.M = M_rt;   // the rows of THIS launch (the cached table holds the largest count: packed rows change every batch)
  wgrad_dma_body<BIAS, NS, true>(p, d.tiles_k, d.tile, 0, (p.M + WD_CH - 1) / WD_CH, smem);
}

// Grouped launch for small row counts (round 3: the E(3)-GNN's 22 node-level weight gradients, 16 384 rows each, were 22
// launches of ~22 us that each filled a fraction of the machine): one entry per (problem, tile, slice of M); the slices of a tile
// meet in fp32 atomics like the ungrouped kernel's splits.
template <int NS>
__global__ __launch_bounds__(512, 1) void wgrad_dma_split_table_kernel(const WgradTile* __restrict__ table) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WgradTile& d = table[blockIdx.x];
  wgrad_dma_body<true, NS, false>(d.p, d.tiles_k, d.tile, d.c_begin, d.c_end, smem);
}

// =================================================================================================
// Grouped weight gradients, 256 x 256 output tiles (round 2).  The 128 x 128 table kernel above is bound by the L2 -> LDS
// fill rate of a CU (profiles/r02_wgrad_traffic.txt: 0.73 us per 32-KiB stage against 0.49 us for the DMAs alone) and by
// what its sibling tiles re-read: a 128-wide tile fetches every operand panel of its problem 2..8 times.  A 256 x 256 tile
// does twice the flops per byte brought into LDS (131 flop/B instead of 64) and has at most four siblings per panel.
//   * one 512-thread workgroup = 2 x 4 waves, each wave a 128 (n) x 64 (k) block: acc[4][2] = 128 accumulator VGPRs;
//   * a stage is 32 m-rows: [A0 | A1 | B0 | B1], four [32 m][128 col] blocks in the layout of the kernel above (256-B rows,
//     64-B chunks XOR-swizzled by row & 3 on the GLOBAL side of the DMA, ds_read_b64_tr_b16 fragments) = 32 KiB; wave w
//     issues the four 1-KiB pieces "rows 4 w .. 4 w + 3" of the four blocks; ring of NS stages, NS - 1 in flight;
//   * the fragments are double-buffered per k-step of 16 rows (6 fragments = 24 VGPRs a set), not per stage:
//       step c:  read (c, ks 1) | 8 MFMAs of (c, ks 0) | vmcnt + barrier | DMA of stage c + NS | read (c + 1, ks 0) | 8 MFMAs of (c, ks 1)
//     16 MFMAs (512 matrix-core cycles) per wave per barrier, twice the ratio of the 128-wide kernel;
//   * the tile is exclusive (one workgroup streams all of M): plain read-add-store commit, no atomics, no exchange;
//   * N and K must be multiples of 256 (the transformer's are); only the last stage of M needs the zero page.
// 16 layers x 12 tiles = 192 workgroups: one round on 192 of the 256 CUs, 24 per XCD = two whole layers per L2.
// =================================================================================================
#define W2_CH 32
#define W2_BLK_BYTES (32 * WT_ROW_BYTES)
#define W2_STAGE_BYTES (4 * W2_BLK_BYTES)
struct W2Frags { bf16x8 a[4], b[2]; };

// (A split form that also used the 64 idle CUs -- three quarters of M on a tile's main workgroup, the last quarter on a helper,
// ordered commits through ticket counters -- was built in round 2 and measured no faster (2.14 vs 2.04 ms): the launch is bound
// by what the memory system delivers, not by the number of CUs.  Removed in round 3; see DESIGN.md.)
template <int NS>
__global__ __launch_bounds__(512, 1) void wgrad256_table_kernel(const WgradTile* __restrict__ table, int M_rt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WgradTile& d = table[xcd_swizzle(blockIdx.x, gridDim.x)];
  WgradArgs p = d.p;
  if (M_rt > 0) p.M = M_rt;   // the rows of THIS launch (see wgrad_dma_table_kernel)
  const int tiles_k = d.tiles_k, tile = d.tile;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
  const int n0 = tile_n * 256, k0 = tile_k * 256;
  const int c_begin = 0;
  const int c_end = (p.M + W2_CH - 1) / W2_CH;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);

  // DMA sources of this lane: piece "rows 4 wave .. + 3" of block b = 0..3 (A0, A1, B0, B1); LDS slot (row & 3 = lane >> 4,
  // 16-B chunk q = lane & 15) <- global chunk qg of that row
  const int rsub = lane >> 4, q = lane & 15;
  const int qg = (((q >> 2) ^ rsub) << 2) | (q & 3);
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(wd_zero_page) + q * 8;
  const unsigned offa = (unsigned)(((4 * wave + rsub) * (int)p.lda + qg * 8) * 2);
  const unsigned offb = (unsigned)(((4 * wave + rsub) * (int)p.ldb + qg * 8) * 2);
  int cnext = c_begin;
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u8*)smem);
  const unsigned ldsw = lds0 + wave * 1024;
  const bf16_t* ab = A + n0 + (long long)c_begin * W2_CH * p.lda;       // row 0 of the next stage (scalar)
  const bf16_t* bb = p.B + k0 + (long long)c_begin * W2_CH * p.ldb;
  const long long ab_step = (long long)W2_CH * p.lda, bb_step = (long long)W2_CH * p.ldb;
  auto issue = [&](int slot) __attribute__((always_inline)) {
    const unsigned S = ldsw + slot * W2_STAGE_BYTES;
    if ((cnext + 1) * W2_CH <= p.M) {
      wd_dma16s(ab, offa, S);
      wd_dma16s(ab, offa + 256u, S + W2_BLK_BYTES);
      wd_dma16s(bb, offb, S + 2 * W2_BLK_BYTES);
      wd_dma16s(bb, offb + 256u, S + 3 * W2_BLK_BYTES);
    } else {
      const int m = cnext * W2_CH + 4 * wave + rsub;
      const bool ok = m < p.M;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        wd_dma16(ok ? A + n0 + 128 * h + qg * 8 + (long long)m * p.lda : zero, S + h * W2_BLK_BYTES);
        wd_dma16(ok ? p.B + k0 + 128 * h + qg * 8 + (long long)m * p.ldb : zero, S + (2 + h) * W2_BLK_BYTES);
      }
    }
    ++cnext;
    ab += ab_step;
    bb += bb_step;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }

  // fragment addresses: A block wm, 64-B chunk i (n columns 32 i ..); B block wn >> 1, chunk 2 (wn & 1) + j.  A second set
  // 64 KiB up (opaque to the compiler) keeps the immediates of ring slots 2, 3 inside the 16-bit offset field.
  typedef __attribute__((address_space(3))) v4s16 lds_v4;
  unsigned la[4], lb[2], lah[4], lbh[2], lax[4], lbx[2];   // (lax / lbx: a third set 128 KiB up, ring slot 4 of the 5-deep ring)
  {
    const int g = lane >> 4, kg = g >> 1, r0 = 8 * kg + ((lane & 15) >> 2), cb = (16 * (g & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) la[i] = lds0 + wm * W2_BLK_BYTES + wt_offset(r0, i * 64 + cb);
#pragma unroll
    for (int j = 0; j < 2; ++j) lb[j] = lds0 + (2 + (wn >> 1)) * W2_BLK_BYTES + wt_offset(r0, (2 * (wn & 1) + j) * 64 + cb);
#pragma unroll
    for (int i = 0; i < 4; ++i) { lah[i] = la[i] + 0x10000u; asm volatile("" : "+v"(lah[i])); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { lbh[j] = lb[j] + 0x10000u; asm volatile("" : "+v"(lbh[j])); }
    if constexpr (NS > 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { lax[i] = la[i] + 0x20000u; asm volatile("" : "+v"(lax[i])); }
#pragma unroll
      for (int j = 0; j < 2; ++j) { lbx[j] = lb[j] + 0x20000u; asm volatile("" : "+v"(lbx[j])); }
    }
  }
  auto frag_at = [&](unsigned addr) __attribute__((always_inline)) {
    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(size_t)addr);
    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(size_t)(addr + 4 * WT_ROW_BYTES));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    const v8s16 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  };
  auto fetch = [&](int slot, int ks, W2Frags& f) __attribute__((always_inline)) {
    const int bank = slot * W2_STAGE_BYTES / 0x10000;   // which 64-KiB window the slot lies in
    const unsigned so = slot * W2_STAGE_BYTES - bank * 0x10000u + ks * 16 * WT_ROW_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = frag_at((bank == 2 ? lax[i] : bank == 1 ? lah[i] : la[i]) + so);
#pragma unroll
    for (int j = 0; j < 2; ++j) f.b[j] = frag_at((bank == 2 ? lbx[j] : bank == 1 ? lbh[j] : lb[j]) + so);
  };
  auto mma = [&](const W2Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
  };
  // bias (column sums of A): thread -> columns 2 cp, 2 cp + 1 of the 256, rows 8 rq .. + 7 of the stage; the tiles_k
  // workgroups that stream the same A rows take turns (stage % tiles_k == tile_k)
  typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
  unsigned sel_lo_u, sel_hi_u;
  asm volatile("v_mov_b32 %0, 0x3f80\n\tv_mov_b32 %1, 0x3f800000" : "=v"(sel_lo_u), "=v"(sel_hi_u));
  const v2bf sel_lo = __builtin_bit_cast(v2bf, sel_lo_u), sel_hi = __builtin_bit_cast(v2bf, sel_hi_u);
  float cs0 = 0.f, cs1 = 0.f;
  const int cp = tid & 127, rq = tid >> 7;
  int bias_ctr = (tile_k - c_begin % tiles_k + tiles_k) % tiles_k;   // stages until this workgroup's turn
  unsigned bu[8];
  auto bias_read = [&](int slot) __attribute__((always_inline)) {
    const unsigned char* At = smem + slot * W2_STAGE_BYTES + (cp >> 6) * W2_BLK_BYTES;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) bu[rr] = *reinterpret_cast<const unsigned*>(At + wt_offset(rq * 8 + rr, (cp & 63) * 4));
  };
  auto bias_add = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const v2bf u = __builtin_bit_cast(v2bf, bu[rr]);
      cs0 = __builtin_amdgcn_fdot2_f32_bf16(u, sel_lo, cs0, false);
      cs1 = __builtin_amdgcn_fdot2_f32_bf16(u, sel_hi, cs1, false);
    }
  };
  constexpr int VM_KEEP = 4 * (NS - 2);
  constexpr int VM_WAIT = 0x0f70 | (VM_KEEP & 15) | ((VM_KEEP >> 4) << 14);

#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0);
  __builtin_amdgcn_s_waitcnt(VM_WAIT);
  __builtin_amdgcn_s_barrier();
  issue(NS - 1);
  W2Frags f0, f1;
  fetch(0, 0, f0);
  auto step = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value, SN = (SL + 1) % NS;
    const bool turn = bias_ctr == 0;
    bias_ctr = (bias_ctr == 0 ? tiles_k : bias_ctr) - 1;
    if (turn) bias_read(SL);
    fetch(SL, 1, f1);
    mma(f0);
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
    }
    __builtin_amdgcn_sched_barrier(0);
    if (turn) bias_add();
    __builtin_amdgcn_s_waitcnt(VM_WAIT & 0xf0ff);   // stage c + 1 has landed; + lgkmcnt(0): this wave's reads of stage c are complete
    __builtin_amdgcn_s_barrier();
    issue(SL);
    fetch(SN, 0, f0);
    mma(f1);
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  static_assert(NS == 4 || NS == 5, "the main loop is unrolled for rings of 4 and 5");
  for (int c = c_begin;;) {
    step(WdSlot<0>()); if (++c >= c_end) break;
    step(WdSlot<1>()); if (++c >= c_end) break;
    step(WdSlot<2>()); if (++c >= c_end) break;
    step(WdSlot<3>()); if (++c >= c_end) break;
    if constexpr (NS == 5) { step(WdSlot<4>()); if (++c >= c_end) break; }
  }
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) + lgkmcnt(0): the over-issued (zero-page) DMAs have landed, the ring is free
  __builtin_amdgcn_s_barrier();

  // bias partials: [4 row quarters][256 columns] through the (idle) ring; all workgroups of a tile row add into one slice
  float* const bsum = reinterpret_cast<float*>(smem);
  bsum[rq * 256 + 2 * cp] = cs0;
  bsum[rq * 256 + 2 * cp + 1] = cs1;
  __syncthreads();
  if (tid < 256) {
    const float t = bsum[tid] + bsum[256 + tid] + bsum[512 + tid] + bsum[768 + tid];
    if (tiles_k == 1) p.dbias[n0 + tid] += t;
    else atomicAdd(p.dbias + n0 + tid, t);
  }
  // this workgroup is the only one that touches the tile now: plain read-add-store
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 128 + i * 32 + frag_row(r, lane);
        const int k = k0 + wn * 64 + j * 32 + (lane & 31);
        p.dW[(long long)n * p.ldw + k] += acc[i][j][r];
      }
}

bool wgrad_table_tile256_ok(const WgradArgs& a) {
  return a.N % 256 == 0 && a.K % 256 == 0 && a.n_out == 0 && a.m_dev == nullptr && 40LL * a.lda < (1LL << 30) && 40LL * a.ldb < (1LL << 30);
}

int wgrad_table_append(std::vector<WgradTile>& tab, const WgradArgs& a, int tile_size) {
  COATI_CHECK_ARG(a.A && a.B && a.dW && a.dbias, "wgrad_table: null operand (the grouped kernel is the bias variant)");
  COATI_CHECK_SHAPE(a.M > 0 && a.N % 8 == 0 && a.K % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0, "wgrad_table: shape / alignment");
  COATI_CHECK_ARG(tile_size == 128 || tile_size == 256, "wgrad_table: tile size %d", tile_size);
  if (tile_size == 256) {
    COATI_CHECK_SHAPE(wgrad_table_tile256_ok(a), "wgrad_table: 256-wide tiles need N and K to be multiples of 256 (N=%d K=%d)", a.N, a.K);
    const int tk = a.K / 256, nt = (a.N / 256) * tk;
    for (int t = 0; t < nt; ++t) tab.push_back(WgradTile{a, tk, t});
    return COATI_OK;
  }
  const int tiles_n = cdiv(a.N, BM), tiles_k = cdiv(a.K, BN);
  const int n = tiles_n * tiles_k;
  for (int t = 0; t < n; ++t) tab.push_back(WgradTile{a, tiles_k, t});
  return COATI_OK;
}

int wgrad_table_append_split(std::vector<WgradTile>& tab, const WgradArgs& a, int n_splits) {
  COATI_CHECK_ARG(a.A && a.B && a.dW, "wgrad_table(split): null operand");
  COATI_CHECK_SHAPE(a.M > 0 && a.N % 8 == 0 && a.K % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.m_dev == nullptr, "wgrad_table(split): shape / alignment");
  const int tiles_n = cdiv(a.N, BM), tiles_k = cdiv(a.K, BN), nchunks = cdiv(a.M, WD_CH);
  if (n_splits > cdiv(nchunks, 4)) n_splits = cdiv(nchunks, 4);
  if (n_splits < 1) n_splits = 1;
  const int cps = cdiv(nchunks, n_splits);
  for (int c0 = 0; c0 < nchunks; c0 += cps)
    for (int t = 0; t < tiles_n * tiles_k; ++t) {
      WgradTile w{a, tiles_k, t};
      w.c_begin = c0;
      w.c_end = c0 + cps < nchunks ? c0 + cps : nchunks;
      tab.push_back(w);
    }
  return COATI_OK;
}

int launch_wgrad_split_table(const WgradTile* dev_table, int n_entries, hipStream_t s) {
  COATI_CHECK_ARG(dev_table && n_entries > 0, "wgrad_table(split): empty table");
  static bool attr_set = false;
  auto kern = wgrad_dma_split_table_kernel<3>;
  constexpr int lds = 3 * WD_STAGE_BYTES;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      coati_set_error("wgrad(split table): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(n_entries), dim3(512), lds, s, dev_table);
  COATI_LAUNCH_CHECK("wgrad_split_table");
  return COATI_OK;
}

template <int NS>
static int launch_wgrad_table_t(const WgradTile* dev_table, int n_tiles, hipStream_t s, int M_rt) {
  static bool attr_set = false;
  auto kern = wgrad_dma_table_kernel<true, NS>;
  constexpr int lds = NS * WD_STAGE_BYTES;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      coati_set_error("wgrad(table): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(512), lds, s, dev_table, M_rt);
  COATI_LAUNCH_CHECK("wgrad_table");
  return COATI_OK;
}

template <int NS>
static int launch_wgrad256_t(const WgradTile* dev_table, int n_tiles, hipStream_t s, int M_rt) {
  static bool attr_set = false;
  auto kern = wgrad256_table_kernel<NS>;
  constexpr int lds = NS * W2_STAGE_BYTES;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      coati_set_error("wgrad(table 256): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(512), lds, s, dev_table, M_rt);
  COATI_LAUNCH_CHECK("wgrad_table256");
  return COATI_OK;
}

int launch_wgrad_table(const WgradTile* dev_table, int n_tiles, hipStream_t s, int tile_size, int M_rt) {
  COATI_CHECK_ARG(dev_table && n_tiles > 0, "wgrad_table: empty table");
  if (tile_size == 256) {
    // ring of 4 stages of 32 KiB (96 KiB in flight per CU); a 5-deep ring (all 160 KiB of LDS) measured the same (round 2)
    return launch_wgrad256_t<4>(dev_table, n_tiles, s, M_rt);
  }
  return launch_wgrad_table_t<4>(dev_table, n_tiles, s, M_rt);   // ring of 4 stages of 32 KiB, 3 in flight
}

template <bool BIAS, int NS>
static int launch_wgrad_dma_t(const WgradArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = wgrad_dma_kernel<BIAS, NS>;
  constexpr int lds = NS * WD_STAGE_BYTES;
  static_assert(lds >= 8 * 2048 * 4 + 8 * 128 * 4, "the ring doubles as the 64-KiB exchange buffer + the bias partials");
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      coati_set_error("wgrad(dma): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int tiles_n = cdiv(a.N, BM), tiles_k = cdiv(a.K, BN);
  const int tiles = tiles_n * tiles_k;
  const int nchunks = cdiv(a.M, WD_CH);
  const int slots = tiles > 128 ? 512 : 256;   // one workgroup per CU (two rounds when the tiles alone almost fill one)
  int splits = slots / tiles;
  if (splits > cdiv(nchunks, 4)) splits = cdiv(nchunks, 4);
  if (splits < 1) splits = 1;
  const int cps = cdiv(nchunks, splits);
  splits = cdiv(nchunks, cps);
  hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(512), lds, s, a, tiles_k, cps, splits);
  COATI_LAUNCH_CHECK("wgrad_dma");
  return COATI_OK;
}

template <typename AT, bool BIAS>
static int launch_wgrad_t(const WgradArgs& a, hipStream_t s) {
  static bool attr_set = false;
  constexpr int WAVES = 4;   // measured: 8 waves is no faster without bias and slower with the bias column sums
  auto kern = wgrad_kernel<AT, BIAS, WAVES>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       WGRAD_LDS_BYTES);
    if (e != hipSuccess) {
      coati_set_error("wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int tiles_n = cdiv(a.N, BM), tiles_k = cdiv(a.K, BN);
  const int tiles = tiles_n * tiles_k;
  const int nchunks = cdiv(a.M, 64);
  // ~2 workgroups per CU over the whole launch; every split keeps >= 8 chunks (512 rows) of work
  // 2 workgroups per CU x 256 CUs = 512 resident slots: never launch a partial second round.  Every split costs one
  // fp32 atomic per output element, so tiny outputs (<= 4 tiles) use half the slots (measured 43 -> 35 us at 256x256).
  const int slots = tiles <= 4 ? 256 : 512;
  int splits = slots / tiles;
  if (splits > cdiv(nchunks, 8)) splits = cdiv(nchunks, 8);
  if (splits < 1) splits = 1;
  const int cps = cdiv(nchunks, splits);
  splits = cdiv(nchunks, cps);
  hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(64 * WAVES), WGRAD_LDS_BYTES, s, a, tiles_k, cps, splits);
  COATI_LAUNCH_CHECK("wgrad");
  return COATI_OK;
}

int launch_wgrad(const WgradArgs& a, int a_f32, hipStream_t s) {
  COATI_CHECK_ARG(a.A && a.B && a.dW, "wgrad: null operand");
  COATI_CHECK_SHAPE(a.M > 0 && a.N > 0 && a.K > 0, "wgrad: empty problem");
  COATI_CHECK_SHAPE(a.N % 8 == 0 && a.K % 8 == 0, "wgrad: N=%d and K=%d must be multiples of 8", a.N, a.K);
  COATI_CHECK_SHAPE(a.lda % 8 == 0 && a.ldb % 8 == 0, "wgrad: lda/ldb alignment");
  if (a_f32) return a.dbias ? launch_wgrad_t<float, true>(a, s) : launch_wgrad_t<float, false>(a, s);
  // bf16 A: the LDS-DMA kernel (one 512-thread workgroup per CU) when the launch gives every CU a tile and a slice of at
  // least 16 stages; otherwise (few rows: GNN node level; many tiles: lm_head) the register-staged kernel.
  constexpr int dma = 3;   // ring depth (4 was measured equal, round 2)
  const int tiles = cdiv(a.N, BM) * cdiv(a.K, BN);
  // tiles <= 64: one round of 256 workgroups; 129..256 tiles (lm_head: 162): two rounds of up to 512; in between the
  // register-staged kernel's 512 half-CU slots fill the machine better
  const bool fits = tiles <= 64 || (tiles > 128 && tiles <= 256);
  if (dma >= 3 && fits && (long long)a.M >= 16LL * WD_CH * (256 / tiles > 0 ? 256 / tiles : 1)) {
    if (dma == 3) return a.dbias ? launch_wgrad_dma_t<true, 3>(a, s) : launch_wgrad_dma_t<false, 3>(a, s);
    return a.dbias ? launch_wgrad_dma_t<true, 4>(a, s) : launch_wgrad_dma_t<false, 4>(a, s);
  }
  return a.dbias ? launch_wgrad_t<bf16_t, true>(a, s) : launch_wgrad_t<bf16_t, false>(a, s);
}

// =================================================================================================
// exact-f32 MFMA GEMM with generic strides (small problems: heads, InfoNCE logits and their grads)
// =================================================================================================
#define SPITCH 65
#define SBK 32
// 64x64 tile per 256-thread workgroup (one 32x32 MFMA tile per wave), BK = 32, register-prefetched double-buffered
// LDS: the next k-tile's (generic-stride) loads are in flight while the current one feeds 16 MFMAs per wave.
__device__ __forceinline__ void sgemm_body(const float* __restrict__ A, long long ars, long long acs,
                                           const float* __restrict__ B, long long brs, long long bcs,
                                           float* __restrict__ C, long long ldc, int M, int N, int K,
                                           const float* __restrict__ bias, float alpha, int accumulate, int ksplit,
                                           int bx, int by, int bz, bool atomic) {
  __shared__ float As[2][SBK * SPITCH];
  __shared__ float Bs[2][SBK * SPITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * 64, n0 = bx * 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool a_kfast = (acs == 1), b_kfast = (brs == 1);
  float ra[8], rb[8];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + 256 * i;
      int m, k;
      if (a_kfast) { k = id & 31; m = id >> 5; } else { m = id & 63; k = id >> 6; }
      const int gm = m0 + m, gk = k0 + k;
      ra[i] = (gm < M && gk < K) ? A[gm * ars + gk * acs] : 0.f;
      int n, kb;
      if (b_kfast) { kb = id & 31; n = id >> 5; } else { n = id & 63; kb = id >> 6; }
      const int gn = n0 + n, gkb = k0 + kb;
      rb[i] = (gn < N && gkb < K) ? B[gkb * brs + gn * bcs] : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + 256 * i;
      int m, k;
      if (a_kfast) { k = id & 31; m = id >> 5; } else { m = id & 63; k = id >> 6; }
      As[buf][k * SPITCH + m] = ra[i];
      int n, kb;
      if (b_kfast) { kb = id & 31; n = id >> 5; } else { n = id & 63; kb = id >> 6; }
      Bs[buf][kb * SPITCH + n] = rb[i];
    }
  };
  // split-K (bz over the splits, accumulate only): this workgroup reduces k in [kbeg, kend) and adds with atomics
  const int kbeg = bz * ksplit;
  const int kend_ = kbeg + ksplit < K ? kbeg + ksplit : K;
  K = kend_;
  load_tile(kbeg);
  store_tile(0);
  __syncthreads();
  const int nk = (kend_ - kbeg + SBK - 1) / SBK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kbeg + (kt + 1) * SBK);
#pragma unroll
    for (int kk = 0; kk < SBK / 2; ++kk) {
      const int k = 2 * kk + (lane >> 5);
      const float a = As[cur][k * SPITCH + wm * 32 + (lane & 31)];
      const float b = Bs[cur][k * SPITCH + wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + frag_row(r, lane), n = n0 + wn * 32 + (lane & 31);
    if (m < M && n < N) {
      float v = alpha * acc[r];
      if (bias && bz == 0) v += bias[n];
      float* c = C + (long long)m * ldc + n;
      if (atomic) atomicAdd(c, v);
      else *c = accumulate ? (*c + v) : v;
    }
  }
}

__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, long long ars, long long acs,
                                                    const float* __restrict__ B, long long brs, long long bcs,
                                                    float* __restrict__ C, long long ldc, int M, int N, int K,
                                                    const float* __restrict__ bias, float alpha, int accumulate, int ksplit) {
  sgemm_body(A, ars, acs, B, brs, bcs, C, ldc, M, N, K, bias, alpha, accumulate, ksplit, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z > 1);
}
// several independent small problems in ONE launch (round 4: the heads / InfoNCE glue was 22 launches of ~ 18 us, each a grid of
// 16 .. 64 workgroups walking its reduction serially): workgroup w of the grid belongs to the problem whose [wg0, wg0 + count)
// range holds it: 14 launches less per step, 22.34 vs 22.47 ms (A/B/A/B on one box).  Accumulating problems always add with atomics here (two problems of a batch may target one buffer: the shared
// special-token head's weight gradient).
__global__ __launch_bounds__(256) void sgemm_batch_kernel(SgemmBatch b) {
  int w = blockIdx.x, i = 0;
#pragma unroll
  for (int q = 1; q < SGEMM_BATCH_MAX; ++q)
    if (q < b.n && w >= b.p[q].wg0) i = q;
  const SgemmProb& p = b.p[i];
  w -= p.wg0;
  const int bx = w % p.tx, by = (w / p.tx) % p.ty, bz = w / (p.tx * p.ty);
  sgemm_body(p.A, p.ars, p.acs, p.B, p.brs, p.bcs, p.C, p.ldc, p.M, p.N, p.K, p.bias, p.alpha, p.accumulate, p.ksplit, bx, by, bz, p.accumulate != 0);
}

static void sgemm_split(int M, int N, int K, int accumulate, int& splits, int& ksplit) {
  // few output tiles + a long reduction (the head weight / bias gradients: K = batch): split K over gridDim.z
  const int tiles = cdiv(N, 64) * cdiv(M, 64);
  splits = 1;
  if (accumulate && K >= 256 && tiles < 128) {
    splits = 256 / tiles;
    if (splits > K / 64) splits = K / 64;
    if (splits < 1) splits = 1;
  }
  ksplit = cdiv(cdiv(K, splits), SBK) * SBK;
  splits = cdiv(K, ksplit);
}

int launch_sgemm(const float* A, long long ars, long long acs, const float* B, long long brs, long long bcs,
                 float* C, long long ldc, int M, int N, int K, const float* bias, float alpha, int accumulate,
                 hipStream_t s) {
  COATI_CHECK_ARG(A && B && C, "sgemm: null operand");
  COATI_CHECK_SHAPE(M > 0 && N > 0 && K > 0, "sgemm: empty problem");
  int splits, ksplit;
  sgemm_split(M, N, K, accumulate, splits, ksplit);
  hipLaunchKernelGGL(sgemm_kernel, dim3(cdiv(N, 64), cdiv(M, 64), splits), dim3(256), 0, s, A, ars, acs, B, brs, bcs, C, ldc,
                     M, N, K, bias, alpha, accumulate, ksplit);
  COATI_LAUNCH_CHECK("sgemm");
  return COATI_OK;
}

int sgemm_batch_add(SgemmBatch& b, const float* A, long long ars, long long acs, const float* B, long long brs, long long bcs,
                    float* C, long long ldc, int M, int N, int K, const float* bias, float alpha, int accumulate) {
  COATI_CHECK_ARG(A && B && C, "sgemm_batch: null operand");
  COATI_CHECK_SHAPE(M > 0 && N > 0 && K > 0, "sgemm_batch: empty problem");
  COATI_CHECK_SHAPE(b.n < SGEMM_BATCH_MAX, "sgemm_batch: more than %d problems", SGEMM_BATCH_MAX);
  SgemmProb& p = b.p[b.n];
  p.A = A; p.ars = ars; p.acs = acs; p.B = B; p.brs = brs; p.bcs = bcs; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.alpha = alpha; p.accumulate = accumulate;
  int splits;
  sgemm_split(M, N, K, accumulate, splits, p.ksplit);
  p.tx = cdiv(N, 64); p.ty = cdiv(M, 64); p.tz = splits;
  p.wg0 = b.n == 0 ? 0 : b.p[b.n - 1].wg0 + b.p[b.n - 1].tx * b.p[b.n - 1].ty * b.p[b.n - 1].tz;
  ++b.n;
  return COATI_OK;
}

int launch_sgemm_batch(const SgemmBatch& b, hipStream_t s) {
  if (b.n == 0) return COATI_OK;
  const SgemmProb& l = b.p[b.n - 1];
  hipLaunchKernelGGL(sgemm_batch_kernel, dim3(l.wg0 + l.tx * l.ty * l.tz), dim3(256), 0, s, b);
  COATI_LAUNCH_CHECK("sgemm_batch");
  return COATI_OK;
}
