// Ring GEMM for N = 256, second shape: 320-row blocks (round 2).  C[M,256] = A[M,K] * W[256,K]^T (+ bias, + f32 residual),
// bf16 operands, K % 32 == 0, K >= 256 -- the same products as gemm_ring.hip (reference basic_transformer.py:103-123 MLP
// down-projection forward, the input gradients of c_fc / c_attn / c_proj).
//
// Why a second shape.  The shader-clock trace of the 160-row ring kernel (tools/probes/rb_trace.py) puts a 64-k stage at
// ~2.6 k cycles against 1.5 k of MFMA issue: its k loop is bound by the L2 -> LDS fill (~20 B/clk per CU, the ceiling the
// weight-gradient study measured is ~23), and 32 of the 52 KiB of a stage are the weight rows, which EVERY 160-row block
// streams again.  With 320 rows per block (M = 81,920: 256 blocks = exactly one per CU) the weight is streamed once per CU
// instead of twice: fill bytes per (row x k) drop from 5.1 to 3.5.
//   * 10 waves, wave w owns rows 32 w .. + 31 and ALL 256 columns: 8 accumulator blocks = 128 VGPRs;
//   * a stage is 64 k: [320 activation rows | 256 weight rows] x 128 B = 72 KiB, DOUBLE-buffered (144 KiB): one stage in
//     flight while the other is multiplied -- enough when a stage's fill (~3.6 k cycles at the ~20 B/clk a CU takes in) is
//     longer than its 32 MFMAs per wave (~3.1 k cycles on a 3-wave SIMD); rows are unpadded, the 16-B chunk c of row r sits at
//     chunk position c ^ ((r >> 1) & 7) (applied on the global side of the DMA);
//   * per stage and wave: 4 activation fragments, 32 weight fragments streamed three deep, 32 MFMAs (32 x 256 x 64);
//   * results leave straight from the accumulator layout (lane = column: one 128-B line per half-wave and instruction),
//     the residual in four passes of 32 values per lane.
// (A first version used 32-k stages in a 4-slot ring: 64-B rows, DMA pieces of 16 half lines instead of 8 whole ones --
// slower than the 160-row kernel on every site.)
// The DMA is issued from inline assembly (see gemm.hip): the vmcnt bookkeeping of the ring is hand-placed.
#include <cstdlib>
#include "kernels.h"

#define R2_BR 320
#define R2_BK 64
#define R2_WAVES 10
#define R2_NS 2
#define R2_A_BYTES (R2_BR * R2_BK * 2)      // 40 KiB
#define R2_W_BYTES (256 * R2_BK * 2)        // 32 KiB
#define R2_STAGE_BYTES (R2_A_BYTES + R2_W_BYTES)
#define R2_LDS_BYTES (R2_NS * R2_STAGE_BYTES)   // 147,456 B

__device__ __forceinline__ void r2_dma16(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ int r2_frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int EPI>
__global__ __launch_bounds__(64 * R2_WAVES, 1) void gemm_ring320_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x;
  const int nk = p.K / R2_BK;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u8*)smem);
  const int row_blk = blk * R2_BR;

  // ---- DMA side.  Piece q (1 KiB = 8 rows x 128 B, whole lines) of an operand: lane -> row 8 q + (lane >> 3), LDS chunk slot
  // lane & 7, global chunk (lane & 7) ^ ((row >> 1) & 7).  Activation pieces 0..39: wave w takes w, w + 10, w + 20, w + 30;
  // weight pieces 0..31: wave w takes w, w + 10, w + 20 (, w + 30 for waves 0 and 1).  Pieces of one wave have the same
  // row parity pattern, so one chunk permutation per lane: (8 q) >> 1 = 4 q, (4 q) & 7 = 4 (q & 1): q of one parity class
  // only if all pieces of a wave share q & 1 -- w, w + 10, w + 20, w + 30 do.
  const int lrow = lane >> 3;
  const int cg = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const int nwp = wave < 2 ? 4 : 3;
  // (64-k stages, double-buffered: the 32-k form of this kernel had to use 64-B rows to fit four stages, i.e. DMA pieces of
  // 16 half lines -- the cost of a DMA instruction follows the lines it touches -- and lost what the halved weight stream gains)
  const bool ROT = (EPI != EPI_RES_F32) && nk >= 8;
  int kk = ROT ? blk % nk : 0;           // k chunk of the next stage to fetch
  int slot = 0;
  const bf16_t* arow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = row_blk + 8 * (wave + R2_WAVES * i) + lrow;
    r = r < p.M ? r : p.M - 1;             // rows past M re-read row M - 1 (their outputs are never stored)
    arow[i] = A + (long long)r * p.lda + cg * 8;
  }
  const bf16_t* const wrow0 = p.B + (long long)(8 * wave + lrow) * p.ldb + cg * 8;
  const long long w_p = 8LL * R2_WAVES * p.ldb;
  auto issue_a = [&]() __attribute__((always_inline)) {
    const unsigned S = lds0 + slot * R2_STAGE_BYTES + wave * 1024;
    const int ko = kk * R2_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) r2_dma16(arow[i] + ko, S + i * R2_WAVES * 1024);
  };
  auto issue_w = [&]() __attribute__((always_inline)) {
    const unsigned S = lds0 + slot * R2_STAGE_BYTES + R2_A_BYTES + wave * 1024;
    const bf16_t* wb = wrow0 + kk * R2_BK;
    r2_dma16(wb, S);
    r2_dma16(wb + w_p, S + R2_WAVES * 1024);
    r2_dma16(wb + 2 * w_p, S + 2 * R2_WAVES * 1024);
    if (nwp == 4) r2_dma16(wb + 3 * w_p, S + 3 * R2_WAVES * 1024);
    kk = kk + 1 == nk ? 0 : kk + 1;
    slot ^= 1;
  };

  // ---- MFMA side: lane (fr = lane & 31, kg = lane >> 5) reads chunk (2 ks + kg) ^ ((fr >> 1) & 7) of its rows
  const int fr = lane & 31, kg = lane >> 5, swz = (fr >> 1) & 7;
  unsigned xo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xo[ks] = (unsigned)(fr * 128 + (((2 * ks + kg) ^ swz) << 4));
  const unsigned a_row = (unsigned)(wave * 32 * 128);

  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float b = p.bias != nullptr ? p.bias[j * 32 + fr] : 0.f;   // lane = output column
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = b;
  }

  issue_a();
  issue_w();
  for (int s = 0; s < nk; ++s) {
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): stage s has landed (this wave's pieces; the barrier covers the others)
    __builtin_amdgcn_s_barrier();
    // every wave is done with stage s - 1: its buffer takes stage s + 1, the activation pieces now, the weight pieces after
    // the first k-step (a steadier request stream than one burst)
    const bool more = s + 1 < nk;
    if (more) issue_a();
    const unsigned char* S = smem + (s & 1) * R2_STAGE_BYTES;
    const unsigned char* Sa = S + a_row;
    const unsigned char* Sw = S + R2_A_BYTES;
    // 32 weight fragments (k-step ks, column block j), streamed three deep ahead of their MFMAs
    bf16x8 fa[2];
    fa[0] = *reinterpret_cast<const bf16x8*>(Sa + xo[0]);
    bf16x8 fw[3];
    fw[0] = *reinterpret_cast<const bf16x8*>(Sw + xo[0]);
    fw[1] = *reinterpret_cast<const bf16x8*>(Sw + 4096 + xo[0]);
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const int ks = t >> 3, j = t & 7;
      if (t + 2 < 32) {
        const int t2 = t + 2, ks2 = t2 >> 3, j2 = t2 & 7;
        fw[t2 % 3] = *reinterpret_cast<const bf16x8*>(Sw + j2 * 4096 + xo[ks2]);
      }
      if (j == 4 && ks < 3) fa[(ks + 1) & 1] = *reinterpret_cast<const bf16x8*>(Sa + xo[ks + 1]);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1], fw[t % 3], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t == 7 && more) issue_w();
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's reads of stage s are complete before the next barrier frees the buffer
  }

  // ---- write-out, straight from the accumulator layout: register r of block j = row frag_row(r), column 32 j + fr
  const int row0 = row_blk + wave * 32;
  if (EPI == EPI_RES_F32) {
    const float* res = reinterpret_cast<const float*>(p.aux_in) + fr;
    float* out = reinterpret_cast<float*>(p.C) + fr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {           // four passes of 4 rows x 8 column blocks: 32 residual values in flight per lane
      float x[4][8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + r2_frag_row(4 * q + i, lane), rc = row < p.M ? row : p.M - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[i][j] = res[(long long)rc * p.ld_aux + j * 32];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + r2_frag_row(4 * q + i, lane);
        if (row < p.M) {
#pragma unroll
          for (int j = 0; j < 8; ++j) out[(long long)row * p.ldc + j * 32] = acc[j][4 * q + i] + x[i][j];
        }
      }
    }
  } else {
    bf16_t* out = reinterpret_cast<bf16_t*>(p.C) + fr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + r2_frag_row(r, lane);
      if (row < p.M) {
#pragma unroll
        for (int j = 0; j < 8; ++j) out[(long long)row * p.ldc + j * 32] = f2bf(acc[j][r]);
      }
    }
  }
}

bool gemm_ring320_supported(const GemmArgs& a, int a_f32, int epi) {
  if (a.m_dev) return false;
  // EXPERIMENT, off by default: measured per step at M = 81,920 against the 160-row kernel (bench --all-sites, same box), this
  // (second) version: FC2 + residual 3.06 -> 3.16 ms, FC1 dgrad 1.75 -> 1.83, QKV dgrad 1.53 -> 1.71, proj 1.59 -> 1.67; the
  // first version (32-k stages, 4-slot ring, 64-B rows) 3.14..3.40 / 1.73..1.82 / 1.58..1.72 / 1.68..1.75.  Halving the weight
  // stream does not pay for one stage in flight instead of two, the exposed prologue / write-out of a single block per
  // workgroup and a kernel at the 168-register limit.  COATI_RING320=1 selects it.
  static const bool on = getenv("COATI_RING320") != nullptr && atoi(getenv("COATI_RING320")) == 1;
  if (!on || a_f32) return false;
  if (epi != EPI_BF16 && epi != EPI_RES_F32) return false;
  if (a.N != 256 || a.K % R2_BK != 0 || a.K < 256) return false;
  if (a.M < 200 * R2_BR) return false;                           // (fewer than ~200 CUs busy: the 160-row kernel spreads better)
  if (40LL * a.lda >= (1LL << 30) || 40LL * a.ldb >= (1LL << 30)) return false;
  return true;
}

template <int EPI>
static int launch_ring320_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_ring320_kernel<EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R2_LDS_BYTES);
    if (e != hipSuccess) {
      coati_set_error("gemm_ring320: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(a.M, R2_BR)), dim3(64 * R2_WAVES), R2_LDS_BYTES, s, a);
  COATI_LAUNCH_CHECK("gemm_ring320");
  return COATI_OK;
}

int launch_gemm_ring320(const GemmArgs& a, int epi, hipStream_t s) {
  return epi == EPI_RES_F32 ? launch_ring320_t<EPI_RES_F32>(a, s) : launch_ring320_t<EPI_BF16>(a, s);
}
