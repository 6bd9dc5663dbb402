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
//   * a stage is 32 k: [320 activation rows | 256 weight rows] x 64 B = 36 KiB, 4-slot ring = 144 KiB, three stages in
//     flight (108 KiB, as many bytes as the 160-row kernel keeps in flight); rows are unpadded, the 16-B chunk c of row r
//     sits at chunk position c ^ ((r >> 2) & 3) (applied on the global side of the DMA): the 16 lanes of one ds_read_b128
//     cycle (16 different rows, same k chunk) hit 16 different bank groups;
//   * per stage and wave: 2 activation fragments, 16 weight fragments streamed three deep, 16 MFMAs (32 x 256 x 32);
//   * results leave straight from the accumulator layout (lane = column: one 128-B line per half-wave and instruction),
//     the residual in four passes of 32 values per lane.
// The DMA is issued from inline assembly (see gemm.hip): the vmcnt bookkeeping of the ring is hand-placed.
#include <cstdlib>
#include "kernels.h"

#define R2_BR 320
#define R2_BK 32
#define R2_WAVES 10
#define R2_NS 4
#define R2_A_BYTES (R2_BR * R2_BK * 2)      // 20 KiB
#define R2_W_BYTES (256 * R2_BK * 2)        // 16 KiB
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

  // ---- DMA side.  Piece q (1 KiB = 16 rows x 64 B) of an operand: lane -> row 16 q + (lane >> 2), LDS chunk slot lane & 3,
  // global chunk (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3).  A wave takes activation pieces
  // {wave, wave + 10} and weight pieces {wave (, wave + 10 for waves 0..5)}.
  const int lrow = lane >> 2;
  const int cg = (lane & 3) ^ ((lane >> 4) & 3);
  const int nwp = wave < 6 ? 2 : 1;
  const int my_dmas = 2 + nwp;
  // every block walks k from its own starting chunk (only for the bf16-output products: see gemm_ring.hip)
  const bool ROT = (EPI != EPI_RES_F32) && nk >= 16;
  int kk = ROT ? (blk * 2) % nk : 0;     // k chunk of the next stage to fetch
  int slot = 0;
  const bf16_t* arow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = row_blk + 16 * (wave + R2_WAVES * i) + lrow;
    r = r < p.M ? r : p.M - 1;             // rows past M re-read row M - 1 (their outputs are never stored)
    arow[i] = A + (long long)r * p.lda + cg * 8;
  }
  const bf16_t* wrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) wrow[i] = p.B + (long long)(16 * (wave + R2_WAVES * i) + lrow) * p.ldb + cg * 8;
  // the DMAs of a stage are issued in two halves (activation pieces | weight pieces), the second one in the middle of the
  // stage being multiplied: a steadier request stream than a burst behind every barrier
  auto issue_a = [&]() __attribute__((always_inline)) {
    const unsigned S = lds0 + slot * R2_STAGE_BYTES + wave * 1024;
    const int ko = kk * R2_BK;
    r2_dma16(arow[0] + ko, S);
    r2_dma16(arow[1] + ko, S + R2_WAVES * 1024);
  };
  auto issue_w = [&]() __attribute__((always_inline)) {
    const unsigned S = lds0 + slot * R2_STAGE_BYTES + wave * 1024;
    const int ko = kk * R2_BK;
    r2_dma16(wrow[0] + ko, S + R2_A_BYTES);
    if (nwp == 2) r2_dma16(wrow[1] + ko, S + R2_A_BYTES + R2_WAVES * 1024);
    kk = kk + 1 == nk ? 0 : kk + 1;
    slot = slot + 1 == R2_NS ? 0 : slot + 1;
  };
  auto issue = [&]() __attribute__((always_inline)) {
    issue_a();
    issue_w();
  };

  // ---- MFMA side: lane (fr = lane & 31, kg = lane >> 5) reads chunk (2 ks + kg) ^ ((fr >> 2) & 3) of row fr (+ 32 j)
  const int fr = lane & 31, kg = lane >> 5, sw = (fr >> 2) & 3;
  const unsigned xo0 = (unsigned)(fr * 64 + (((0 + kg) ^ sw) << 4)), xo1 = (unsigned)(fr * 64 + (((2 + kg) ^ sw) << 4));
  const unsigned a_row = (unsigned)(wave * 32 * 64);

  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float b = p.bias != nullptr ? p.bias[j * 32 + fr] : 0.f;   // lane = output column
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = b;
  }

  // prologue: NS - 1 stages in flight
#pragma unroll
  for (int i = 0; i < R2_NS - 1; ++i) issue();
  int sc = 0;
  for (int s = 0; s < nk; ++s) {
    // stage s has landed (this wave's pieces; the barrier covers the others): NS - 2 younger stages may stay in flight
    if (s + R2_NS - 1 <= nk) {
      if (my_dmas == 4) __builtin_amdgcn_s_waitcnt(0x0f78);   // vmcnt(8)
      else __builtin_amdgcn_s_waitcnt(0x0f76);                // vmcnt(6)
    } else {
      __builtin_amdgcn_s_waitcnt(0x0f70);                     // the stream ends: wait for everything
    }
    __builtin_amdgcn_s_barrier();
    // every wave is done with stage s - 1: its slot takes stage s + NS - 1
    const bool more = s + R2_NS - 1 < nk;
    if (more) issue_a();
    const unsigned char* S = smem + sc * R2_STAGE_BYTES;
    const unsigned char* Sa = S + a_row;
    const unsigned char* Sw = S + R2_A_BYTES;
    const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(Sa + xo0), fa1 = *reinterpret_cast<const bf16x8*>(Sa + xo1);
    // 16 weight fragments (k-step ks, column block j), streamed three deep ahead of their MFMAs
    bf16x8 fw[3];
    fw[0] = *reinterpret_cast<const bf16x8*>(Sw + xo0);
    fw[1] = *reinterpret_cast<const bf16x8*>(Sw + 2048 + xo0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int ks = t >> 3, j = t & 7;
      if (t + 2 < 16) {
        const int t2 = t + 2, ks2 = t2 >> 3, j2 = t2 & 7;
        fw[t2 % 3] = *reinterpret_cast<const bf16x8*>(Sw + j2 * 2048 + (ks2 ? xo1 : xo0));
      }
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks ? fa1 : fa0, fw[t % 3], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t == 7 && more) issue_w();
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's reads of stage s are complete before the next barrier frees the slot
    sc = sc + 1 == R2_NS ? 0 : sc + 1;
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
  // EXPERIMENT, off by default: measured per step at M = 81,920 against the 160-row kernel (bench --all-sites, same box):
  // FC2 + residual 3.02 -> 3.14..3.40 ms, FC1 dgrad 1.73 -> 1.73..1.82, QKV dgrad 1.53 -> 1.58..1.72, proj 1.57 -> 1.68..1.75:
  // halving the weight stream did not pay for the exposed prologue / write-out of a single block per workgroup and twice
  // the barriers per k.  COATI_RING320=1 selects it.
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
