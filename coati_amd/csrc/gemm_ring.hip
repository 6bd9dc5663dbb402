// Ring GEMM for N = 256: C[M,256] = A[M,K] * W[256,K]^T (+ bias, + f32 residual), bf16 operands, K % 64 == 0, K >= 256.
// These are the long-K products of the transformer (reference basic_transformer.py:103-123 MLP down-projection forward,
// and the input gradients of c_fc / c_attn): M = 81 920 rows stream through once, the 256 x K weight sits in L2.
//
// One persistent 640-thread workgroup per CU walks over row blocks of 160 rows (grande: 512 blocks = exactly 2 per CU;
// the 128-row tiles of the tiled kernel make 2.5 rounds there and lose a fifth of the machine in the last one).
//   * the block's A rows and the weight rows go HBM/L2 -> LDS with global_load_lds_dwordx4 into a ring of 3 stages of
//     64 k ([160 rows | 256 weight rows] x 128 B = 52 KiB); rows are unpadded, the 16-B chunk c of row r sits at chunk
//     position c ^ ((r >> 1) & 7) (applied on the global side of the DMA) so that the 16 lanes of one LDS cycle of a
//     fragment read (16 consecutive rows, same k chunk) hit 16 different 16-B columns;
//   * the stream of stages is continuous ACROSS row blocks: the DMA pointer runs two stages ahead of the MFMA pointer, so
//     while the waves write a block's results, the first stages of the next block are already landing;
//   * 10 waves = 5 row groups x 2 column halves, 32 x 128 outputs per wave (4 accumulator blocks, 5 fragment reads per
//     4 MFMAs);
//   * results leave straight from the accumulator layout: for one register the 32 lanes of a half-wave hold 32
//     consecutive columns of one row, i.e. one whole 128-B line of an f32 row (residual read + store) -- no LDS staging,
//     so the ring can use all of the LDS.
// The DMA is issued from inline assembly (see gemm.hip: the compiler then leaves the vmcnt bookkeeping of the ring to
// the hand-placed waits).
#include <cstdlib>
#include "kernels.h"

#ifndef RG_ABLATE
#define RG_ABLATE 0     // probe builds (COATI_AMD_CXXFLAGS=-DRG_ABLATE=n, tools/ring_ablate.py): 1 = no weight stream, 2 = no A stream, 3 = neither, 4 = one fragment read per stage, 7 = 3 + 4 (gemm_ring1_kernel's k loop)
#endif
#ifndef RG_PRIO
#define RG_PRIO 0   // (probe: MFMA section at raised priority -- within the noise on every ring site, unlike the row-block kernel)
#endif
// Row groups per block RGM = 5 (160 rows, 10 waves: M = 81 920 is exactly 2 blocks per CU) or 4 (128 rows, 8 waves): the launch
// picks the block height whose last round wastes less -- packed rows bring ~50 000 rows per pass, i.e. 1.2 rounds of 160-row
// blocks (every CU waits for the 53 that run a second block) but 1.5 rounds of 128-row blocks, 20 % fewer rows on the busiest CU.
#define RG_BK 64                            // k per stage
#define RG_NS 3
#define RG_MAXW 10
#define RG_W_BYTES (256 * RG_BK * 2)        // 32 KiB
template <int RGM> struct RgShape {
  static constexpr int BR = 32 * RGM;                        // rows per block
  static constexpr int WAVES = 2 * RGM;
  static constexpr int A_BYTES = BR * RG_BK * 2;             // 20 / 16 KiB
  static constexpr int STAGE_BYTES = A_BYTES + RG_W_BYTES;
  static constexpr int LDS_BYTES = RG_NS * STAGE_BYTES;      // 159,744 / 147,456 B
};

__device__ __forceinline__ void rg_dma16(const void* g, unsigned lds) {   // per-lane 64-bit address
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void rg_dma16s(const void* base, unsigned off, unsigned lds) {   // scalar base + lane offset
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
// uniform base + 32-bit BYTE offset per lane + compile-time immediate: the global_load / global_store "saddr" form (one VGPR of
// address per row instead of a 64-bit pair per access)
template <typename T> __device__ __forceinline__ T rg_ld(const T* base, unsigned byte_off, int imm) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off + imm);
}
// (volatile: a second read of the same address must BE a second read -- the compiler otherwise keeps the first one's value alive in
// registers it does not have)
template <typename T> __device__ __forceinline__ T rg_ld_again(const T* base, unsigned byte_off, int imm) {
  return *reinterpret_cast<const volatile T*>(reinterpret_cast<const char*>(base) + byte_off + imm);
}
template <typename T> __device__ __forceinline__ void rg_st(T* base, unsigned byte_off, int imm, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off + imm) = v;
}
__device__ __forceinline__ int rg_frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// s_waitcnt vmcnt(n) for the values the ring uses (the count is an immediate); anything else waits for less
__device__ __forceinline__ void rg_wait_vm(int n) {
  if (n >= 63) __builtin_amdgcn_s_waitcnt(0xcf7f);        // vmcnt(63)
  else if (n >= 38) __builtin_amdgcn_s_waitcnt(0x8f76);   // vmcnt(38)
  else if (n >= 37) __builtin_amdgcn_s_waitcnt(0x8f75);   // vmcnt(37)
  else if (n >= 6) __builtin_amdgcn_s_waitcnt(0x0f76);    // vmcnt(6)
  else if (n >= 5) __builtin_amdgcn_s_waitcnt(0x0f75);    // vmcnt(5)
  else __builtin_amdgcn_s_waitcnt(0x0f70);
}

// Probe build (-DCOATI_RB_TRACE, tools/probes/rb_trace.py): shader-clock totals per phase for the waves of the first 16
// workgroups of the last launch: [wg][wave][before the loop, wait (vmcnt + barrier), MFMA + DMA issue, write-out].
#ifdef COATI_RB_TRACE
__device__ unsigned long long rg_trace_buf[16 * 16 * 8];   // [wg][wave <= 16][8 phases]; the ring256 kernels fill [wave < 10][4], the one-round kernel [wave < 14][6]
extern "C" int coati_rg_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_trace_buf), sizeof(rg_trace_buf)) == hipSuccess ? 0 : -3;
}
#define RG_T0() unsigned long long rg_t_last = __builtin_amdgcn_s_memtime(), rg_t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define RG_T(i) do { const unsigned long long rg_t_now = __builtin_amdgcn_s_memtime(); rg_t_acc[i] += rg_t_now - rg_t_last; rg_t_last = rg_t_now; } while (0)
#define RG_TDUMP() do { if (blockIdx.x < 16 && lane == 0) { for (int i = 0; i < 8; ++i) rg_trace_buf[(blockIdx.x * 16 + wave) * 8 + i] = rg_t_acc[i]; } } while (0)
#else
#define RG_T0() do { } while (0)
#define RG_T(i) do { } while (0)
#define RG_TDUMP() do { } while (0)
#endif

template <int EPI, int RGM>
__global__ __launch_bounds__(64 * 2 * RGM, 1) void gemm_ring256_kernel(GemmArgs p, int nblocks) {
  constexpr int RG_BR = RgShape<RGM>::BR, RG_WAVES = RgShape<RGM>::WAVES, RG_A_BYTES = RgShape<RGM>::A_BYTES,
                RG_STAGE_BYTES = RgShape<RGM>::STAGE_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x, wg = blockIdx.x;
  const int nk = p.K / RG_BK;
  const int mine = (nblocks - wg + G - 1) / G;   // row blocks wg, wg + G, ...
  if (mine <= 0) return;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u8*)smem);

  // ---- DMA side.  Piece q (1 KiB = 8 rows x 128 B) of an operand: lane -> row 8 q + (lane >> 3), LDS chunk slot
  // lane & 7, global chunk (lane & 7) ^ ((row >> 1) & 7).  A wave takes A pieces {wave, wave + 10} and weight pieces
  // {wave, wave + 10, wave + 20 (, wave + 30 for waves 0 and 1)}: all of one parity, so one chunk permutation per lane.
  const int lrow = lane >> 3;
  const int cg = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const unsigned offa = (unsigned)((lrow * (int)p.lda + cg * 8) * 2);
  const unsigned offw = (unsigned)((lrow * (int)p.ldb + cg * 8) * 2);
  const int nwp = (32 - wave + RG_WAVES - 1) / RG_WAVES;   // weight pieces of this wave: 4 for waves 0, 1 of 10, else 3; 4 each of 8
  const int my_dmas = 2 + nwp;
  // DMA pointer: all scalar, carried from stage to stage (the instruction count of a stage is on the critical path: the
  // first version recomputed bases and the k rotation every stage and spent 7 scalar instructions per MFMA).
  //   every block walks k from its own starting chunk (blk % nk): with a power-of-two row pitch all blocks would
  //   otherwise read the same 128-B column of their rows at the same time, i.e. the same few HBM channels
  int bi = 0, ki = 0;                            // block index (of mine), stages issued of that block
  // (only for the bf16-output products, i.e. the input gradients: the residual product is the forward fc2, whose rows
  //  must not depend on where in the batch they sit -- the k order is part of the fp32 sum)
  //  K < 512 never rotates: no camping at a 512-B pitch, and the GNN's edge product -- a forward one -- has K = 256)
  const bool ROT = (EPI != EPI_RES_F32) && nk >= 8;
  int kk = ROT ? wg % nk : 0;                    // k chunk of the next stage
  int row0 = wg * RG_BR;
  const bf16_t* pa = A + ((long long)row0 + 8 * wave) * p.lda;          // this wave's first A piece, k = 0
  const long long a_p1 = 8LL * RG_WAVES * p.lda;                          // second A piece
  const bf16_t* const pw = p.B + (long long)8 * wave * p.ldb;            // first weight piece, k = 0
  const long long w_p = 8LL * RG_WAVES * p.ldb;                           // stride between this wave's weight pieces
  unsigned sl = lds0 + wave * 1024;                                       // LDS address of piece `wave` of the next slot
  // The DMAs of a stage are issued in three parts (A pieces | weight pieces 0, 1 | weight pieces 2, 3), spread over the
  // k steps of the stage being multiplied, so that the requests reach the memory system as a steady stream instead of a
  // burst behind every barrier.
  auto issue_part = [&](int part) __attribute__((always_inline)) {
    if (bi < mine) {
      if (part == 0) {
        if (row0 + RG_BR <= p.M) {
          const bf16_t* ab = pa + kk * RG_BK;
          rg_dma16s(ab, offa, sl);
          rg_dma16s(ab + a_p1, offa, sl + RG_WAVES * 1024);
        } else {   // last block: rows past M re-read row M - 1 (their outputs are never stored)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            int r = row0 + (wave + RG_WAVES * i) * 8 + lrow;
            r = r < p.M ? r : p.M - 1;
            rg_dma16(A + (long long)r * p.lda + kk * RG_BK + cg * 8, sl + i * RG_WAVES * 1024);
          }
        }
      } else {
        const bf16_t* wb = pw + kk * RG_BK;
        if (part == 1) {
          rg_dma16s(wb, offw, sl + RG_A_BYTES);
          rg_dma16s(wb + w_p, offw, sl + RG_A_BYTES + RG_WAVES * 1024);
        } else {
          rg_dma16s(wb + 2 * w_p, offw, sl + RG_A_BYTES + 2 * RG_WAVES * 1024);
          if (nwp == 4) rg_dma16s(wb + 3 * w_p, offw, sl + RG_A_BYTES + 3 * RG_WAVES * 1024);
        }
      }
    }
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    kk = kk + 1 == nk ? 0 : kk + 1;
    sl = sl + RG_STAGE_BYTES == lds0 + wave * 1024 + RG_NS * RG_STAGE_BYTES ? lds0 + wave * 1024 : sl + RG_STAGE_BYTES;
    if (++ki == nk) {   // next block of this workgroup
      ki = 0;
      ++bi;
      row0 += G * RG_BR;
      pa += (long long)G * RG_BR * p.lda;
      kk = ROT ? (wg + bi * G) % nk : 0;
    }
  };
  auto issue = [&]() __attribute__((always_inline)) {
    issue_part(0);
    issue_part(1);
    issue_part(2);
    issue_advance();
  };

  // ---- MFMA side: lane (r = lane & 31, kg = lane >> 5) reads chunk (2 ks + kg) ^ ((r >> 1) & 7) of its rows
  const int fr = lane & 31, kg = lane >> 5, swz = (fr >> 1) & 7;
  unsigned xo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xo[ks] = (unsigned)(fr * 128 + (((2 * ks + kg) ^ swz) << 4));
  const unsigned a_row = (unsigned)(wm * 32 * 128), w_row = (unsigned)(RG_A_BYTES + wn * 128 * 128);

  f32x16 acc[4];
  auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = p.bias != nullptr ? p.bias[wn * 128 + j * 32 + fr] : 0.f;   // lane = output column
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = b;
    }
  };

  const int total = mine * nk;
  const int vm_keep = my_dmas * (RG_NS - 2);
  RG_T0();
  issue();
  issue();
  acc_init();
  RG_T(0);
  int sc = 0, kc = 0, bc = 0;   // MFMA pointer: ring slot, k chunk, block index
  int st_cnt = 0, st_age = 2;   // stores issued by the last write-out, stages since then
  for (int s = 0; s < total; ++s) {
    // stage s has landed (this wave's pieces; the barrier covers the others) and every wave is done with stage s - 1,
    // whose slot the next DMA overwrites.  Near the end of the stream nothing is issued any more: wait for everything.
    // After a block's write-out the stores are the youngest entries of the (in-order) vmcnt queue: for the next two
    // stages the wait lets them -- `st_cnt` instructions -- stay in flight too instead of draining them.
    if (s + RG_NS - 1 < total) {
      rg_wait_vm(vm_keep + (st_age < 2 ? st_cnt : 0));
    } else {
      __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    ++st_age;
    __builtin_amdgcn_s_barrier();
    RG_T(1);
#if RG_PRIO
    __builtin_amdgcn_s_setprio(3);   // MFMA + DMA-issue section at raised wave priority, write-out at the default one (see gemm_rb.hip)
#endif
    const unsigned char* S = smem + sc * RG_STAGE_BYTES;
    // two fragment sets: the 5 reads of k step ks + 1 are issued in the shadow of the 4 MFMAs of step ks
    {
      bf16x8 fa[2], fw[2][4];
      fa[0] = *reinterpret_cast<const bf16x8*>(S + a_row + xo[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[0][j] = *reinterpret_cast<const bf16x8*>(S + w_row + j * 4096 + xo[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {
          issue_part(ks);
          fa[nxt] = *reinterpret_cast<const bf16x8*>(S + a_row + xo[ks + 1]);
#pragma unroll
          for (int j = 0; j < 4; ++j) fw[nxt][j] = *reinterpret_cast<const bf16x8*>(S + w_row + j * 4096 + xo[ks + 1]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur], fw[cur][j], acc[j], 0, 0, 0);
        if (ks < 3) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
          for (int j = 1; j < 4; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
      }
      issue_advance();
    }
    sc = sc + 1 == RG_NS ? 0 : sc + 1;
    RG_T(2);
#if RG_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (++kc == nk) {
      // the block is complete: write it out straight from the accumulator layout (the next block's first stages are in
      // flight meanwhile)
      const int row0 = (wg + bc * G) * RG_BR + wm * 32;
      const int brow0 = (wg + bc * G) * RG_BR;
      const bool full = brow0 + RG_BR <= p.M;   // uniform: no per-row guards (and no branches between the stores) inside
      if (EPI == EPI_RES_F32) {
        // all 64 residual loads of the wave are issued before the first add (rows past M re-read row M - 1)
        const float* res = reinterpret_cast<const float*>(p.aux_in) + wn * 128 + fr;
        float* out = reinterpret_cast<float*>(p.C) + wn * 128 + fr;
        float x[16][4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + rg_frag_row(r, lane), rc = (full || row < p.M) ? row : p.M - 1;
#pragma unroll
          for (int j = 0; j < 4; ++j) x[r][j] = res[(long long)rc * p.ld_aux + j * 32];
        }
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + rg_frag_row(r, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(long long)row * p.ldc + j * 32] = acc[j][r] + x[r][j];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + rg_frag_row(r, lane);
            if (row < p.M) {
#pragma unroll
              for (int j = 0; j < 4; ++j) out[(long long)row * p.ldc + j * 32] = acc[j][r] + x[r][j];
            }
          }
        }
      } else {
        // bf16: one 2-B store per element (64-B row segments).  Regrouping the columns into 128-B lines with ds_bpermute
        // was measured slower (4 permutes per output dword: 73 vs 64 us at K = 1024).
        bf16_t* out = reinterpret_cast<bf16_t*>(p.C) + wn * 128 + fr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + rg_frag_row(r, lane);
          if (full || row < p.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(long long)row * p.ldc + j * 32] = f2bf(acc[j][r]);
          }
        }
      }
      st_cnt = full ? 64 : 0;   // a partial block skips stores: no credit
      st_age = 0;
      kc = 0;
      ++bc;
      acc_init();
      RG_T(3);
    }
  }
  RG_TDUMP();
}

bool gemm_ring256_supported(const GemmArgs& a, int a_f32, int epi) {
  if (a.m_dev) return false;   // data-dependent row counts: the row-block / tiled kernels read them on the device
  if (a_f32) return false;
  if (epi != EPI_BF16 && epi != EPI_RES_F32) return false;
  if (a.N != 256 || a.K % RG_BK != 0 || a.K < 256) return false;
  if (a.M < 256 * 160 / 2) return false;                        // fewer than half the CUs busy: the tiled kernel spreads better
  if (130LL * a.lda >= (1LL << 30) || 260LL * a.ldb >= (1LL << 30)) return false;
  return true;
}

template <int EPI, int RGM>
static int launch_ring_t(const GemmArgs& a, hipStream_t s) {
  constexpr int RG_BR = RgShape<RGM>::BR, RG_WAVES = RgShape<RGM>::WAVES, RG_LDS_BYTES = RgShape<RGM>::LDS_BYTES;
  static bool attr_set = false;
  auto kern = gemm_ring256_kernel<EPI, RGM>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_BYTES);
    if (e != hipSuccess) {
      coati_set_error("gemm_ring: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int nblocks = cdiv(a.M, RG_BR);
  const int grid = nblocks < 256 ? nblocks : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * RG_WAVES), RG_LDS_BYTES, s, a, nblocks);
  COATI_LAUNCH_CHECK("gemm_ring");
  return COATI_OK;
}


// ---- one round: every workgroup owns ONE span of R <= 224 rows (round 3) -------------------------------------------------
// A packed batch brings ~50 000 rows: 391 blocks of 128 rows are 1.53 rounds over the 256 persistent workgroups -- every launch
// takes the time of two rounds (tools/ring_rounds.py: K = 1024, bf16 out: 22.9 us for 32 768 rows, 37-40 us for anything between
// 36 000 and 65 536).  Here the M rows are cut into 256 equal spans of R = ceil(M / 256) rows (rounded up to 8 = one DMA piece),
// one per CU, 14 waves = 7 row groups x 2 column halves.  A 224-row stage of 64 k is 28 KiB of A + 32 KiB of weight: three of
// them do not fit the LDS, so the two operands get rings of their own depth -- 3 slots for A (HBM latency: issued two stages
// ahead), 2 slots for the weight (L2 latency: one stage ahead): 84 + 64 = 148 KiB.  Issue order inside a stage: W(s + 1) first,
// then A(s + 2), so that the in-order vmcnt wait in front of the next barrier can leave exactly the A(s + 2) pieces in flight.
// Pieces whose rows lie behind the span are not loaded at all (their products are never stored).
#define R1_GROUPS 7
#define R1_BR (32 * R1_GROUPS)
#define R1_WAVES (2 * R1_GROUPS)
#define R1_A_BYTES (R1_BR * RG_BK * 2)                                  // 28 KiB
#define R1_LDS_BYTES (3 * R1_A_BYTES + 2 * RG_W_BYTES)                  // 151,552 B

__device__ __forceinline__ void r1_wait_vm(int n) {   // s_waitcnt vmcnt(n), n = 0 .. 2 (the count is an immediate)
  if (n >= 2) __builtin_amdgcn_s_waitcnt(0x0f72);
  else if (n == 1) __builtin_amdgcn_s_waitcnt(0x0f71);
  else __builtin_amdgcn_s_waitcnt(0x0f70);
}

template <int EPI>
__global__ __launch_bounds__(64 * R1_WAVES, 1) void gemm_ring1_kernel(GemmArgs p, int R) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = p.K / RG_BK;
  const int row0 = blockIdx.x * R;
  if (row0 >= p.M) return;
  RG_T0();   // probe build: [before the loop, wait (vmcnt + barrier), MFMA + DMA issue, LayerNorm write-out (or the plain one), dgamma / dbeta, chained product]
  const int row_end = row0 + R < p.M ? row0 + R : p.M, nvalid = row_end - row0;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u8*)smem);
  const unsigned ldsW = lds0 + 3 * R1_A_BYTES;

  // ---- DMA side (pieces of 8 rows x 128 B, chunk permutation as in gemm_ring256_kernel): A pieces {wave, wave + 14}, weight
  // pieces {wave, wave + 14 (, wave + 28 for waves 0 .. 3)}
  const int lrow = lane >> 3;
  const int cg = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const unsigned offa = (unsigned)((lrow * (int)p.lda + cg * 8) * 2);
  const unsigned offw = (unsigned)((lrow * (int)p.ldb + cg * 8) * 2);
  const int nwp = wave < 4 ? 3 : 2;
  // an A piece is loaded whole (all 8 rows inside the span), clamped (the span ends inside it: the last workgroup only, R is a
  // multiple of 8) or not at all
  const int q0 = wave, q1 = wave + R1_WAVES;
  const int a0 = 8 * q0 + 8 <= nvalid ? 2 : (8 * q0 < nvalid ? 1 : 0), a1 = 8 * q1 + 8 <= nvalid ? 2 : (8 * q1 < nvalid ? 1 : 0);
  const int nA = (a0 != 0) + (a1 != 0);
  const bool ROT = (EPI != EPI_RES_F32) && nk >= 8;   // (see gemm_ring256_kernel: the forward product keeps its k order)
  const int kk0 = ROT ? (int)(blockIdx.x % (unsigned)nk) : 0;
  const bf16_t* const pa = A + (long long)row0 * p.lda;
  auto issue_a_piece = [&](int q, int mode, int kk, unsigned slot) __attribute__((always_inline)) {
    if (mode == 2) {
      rg_dma16s(pa + (long long)8 * q * p.lda + kk * RG_BK, offa, slot + q * 1024);
    } else if (mode == 1) {
      int r = row0 + 8 * q + lrow;
      r = r < row_end ? r : row_end - 1;
      rg_dma16(A + (long long)r * p.lda + kk * RG_BK + cg * 8, slot + q * 1024);
    }
  };
  auto issue_w = [&](int kk, unsigned slot) __attribute__((always_inline)) {
    const bf16_t* wb = p.B + (long long)8 * wave * p.ldb + kk * RG_BK;
    const long long w_p = 8LL * R1_WAVES * p.ldb;
    rg_dma16s(wb, offw, slot + wave * 1024);
    rg_dma16s(wb + w_p, offw, slot + (wave + R1_WAVES) * 1024);
    if (nwp == 3) rg_dma16s(wb + 2 * w_p, offw, slot + (wave + 2 * R1_WAVES) * 1024);
  };
  auto kwrap = [&](int k) { return k >= nk ? k - nk : k; };

  // ---- MFMA side: lane (r = lane & 31, kg = lane >> 5) reads chunk (2 ks + kg) ^ ((r >> 1) & 7) of its rows
  const int fr = lane & 31, kg = lane >> 5, swz = (fr >> 1) & 7;
  unsigned xo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xo[ks] = (unsigned)(fr * 128 + (((2 * ks + kg) ^ swz) << 4));
  const unsigned a_row = (unsigned)(wm * 32 * 128), w_row = (unsigned)(wn * 128 * 128);

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float b = p.bias != nullptr ? p.bias[wn * 128 + j * 32 + fr] : 0.f;   // lane = output column
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = b;
  }

  // prologue: A(0), W(0), A(1)
  issue_a_piece(q0, a0, kk0, lds0);
  issue_a_piece(q1, a1, kk0, lds0);
  issue_w(kk0, ldsW);
  if (nk > 1) {
    issue_a_piece(q0, a0, kwrap(kk0 + 1), lds0 + R1_A_BYTES);
    issue_a_piece(q1, a1, kwrap(kk0 + 1), lds0 + R1_A_BYTES);
  }
  int sa = 0, sw = 0;                      // ring slots of the stage being multiplied
  int kk1 = kwrap(kk0 + 1), kk2 = kwrap(kk1 + 1);   // k chunks of stages s + 1, s + 2
  // (Round 4, tools/ring_ablate.py: with NO operand stream behind the prologue -- -DRG_ABLATE=3 -- the K = 1024 launch takes 31.3 us
  //  instead of 32.8: the k loop is bound by its own barrier / fragment-read / MFMA cadence, 1.6 us per stage against 0.85 us of MFMA
  //  issue on the 4-wave SIMDs, not by HBM or L2.  Running the MFMA stream one k step behind the reads ACROSS the stage barrier, so
  //  that the matrix core has 4 MFMAs per wave to chew on while the new stage's first fragments come out of LDS, was built and
  //  measured on one box: 34.0-34.2 vs 34.4-34.9 us isolated, 22.42 vs 22.44 ms per step -- nothing; not kept.)
  RG_T(0);
  for (int s = 0; s < nk; ++s) {
    // stage s has landed (this wave's pieces; the barrier covers the others) and every wave is done with stage s - 1, whose
    // slots the DMAs of this stage overwrite; in flight behind the wait: the A pieces of stage s + 1
    r1_wait_vm(s + 1 < nk ? nA : 0);
    __builtin_amdgcn_s_barrier();
    RG_T(1);
    const unsigned char* SA = smem + sa * R1_A_BYTES;
    const unsigned char* SW = smem + 3 * R1_A_BYTES + sw * RG_W_BYTES;
    const int sa2 = sa == 0 ? 2 : sa - 1;                        // (sa + 2) % 3
    {
      bf16x8 fa[2], fw[2][4];
      fa[0] = *reinterpret_cast<const bf16x8*>(SA + a_row + xo[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[0][j] = *reinterpret_cast<const bf16x8*>(SW + w_row + j * 4096 + xo[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {
#if RG_ABLATE == 1      // probe build: no weight stream behind the prologue (stale stage in LDS: timing only)
          if (ks == 0) { }
          else if (s + 2 < nk) issue_a_piece(ks == 1 ? q0 : q1, ks == 1 ? a0 : a1, kk2, lds0 + sa2 * R1_A_BYTES);
#elif RG_ABLATE == 2    // probe build: no A stream behind the prologue
          if (ks == 0) { if (s + 1 < nk) issue_w(kk1, ldsW + (sw ^ 1) * RG_W_BYTES); }
#elif RG_ABLATE == 3 || RG_ABLATE == 7   // probe build: neither (7: and one fragment read per stage instead of four)
#else
          if (ks == 0) { if (s + 1 < nk) issue_w(kk1, ldsW + (sw ^ 1) * RG_W_BYTES); }
          else if (s + 2 < nk) issue_a_piece(ks == 1 ? q0 : q1, ks == 1 ? a0 : a1, kk2, lds0 + sa2 * R1_A_BYTES);
#endif
#if RG_ABLATE >= 4       // probe builds 4 / 7: the fragments of k step 0 serve the whole stage (one LDS read round instead of four)
          fa[nxt] = fa[cur];
#pragma unroll
          for (int j = 0; j < 4; ++j) fw[nxt][j] = fw[cur][j];
#else
          fa[nxt] = *reinterpret_cast<const bf16x8*>(SA + a_row + xo[ks + 1]);
#pragma unroll
          for (int j = 0; j < 4; ++j) fw[nxt][j] = *reinterpret_cast<const bf16x8*>(SW + w_row + j * 4096 + xo[ks + 1]);
#endif
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur], fw[cur][j], acc[j], 0, 0, 0);
        if (RG_ABLATE < 4 && ks < 3) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
          for (int j = 1; j < 4; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
      }
    }
    sa = sa == 2 ? 0 : sa + 1;
    sw ^= 1;
    kk1 = kk2;
    kk2 = kwrap(kk2 + 1);
    RG_T(2);
  }

  // ---- write-out straight from the accumulator layout (one register: 32 consecutive columns of one row per half-wave)
  const int wrow0 = row0 + wm * 32;
  const bool full = (wm + 1) * 32 <= nvalid;   // wave-uniform
  if (EPI == EPI_LNBWD) {
    // The product IS dy of a LayerNorm over these 256 columns (fc1 / c_attn input gradients -> ln_2 / ln_1): its backward runs here
    // on the accumulators instead of in a kernel of its own (one bf16 round trip of dy through HBM and 64 launches per pass less):
    //   g = dy gamma ; c1 = mean_n g ; c2 = mean_n (g xhat) ; dx = rstd (g - c1 - xhat c2) + dres ; dgamma += dy xhat ; dbeta += dy
    // A row's 256 columns live in the two waves (wn = 0 / 1) of its row group, 4 per lane: the two row sums are reduced over the 32
    // lanes of a half-wave by a halving butterfly (8 values -> 9 shuffles) and exchanged with the partner wave through LDS (the ring
    // is idle by now), 4 row slots ("a quarter") per exchange.  Registers are what bounds the write-out (128 per wave, 64 of them
    // accumulators): the second half of the accumulators waits in LDS while the first half is written out, and the loads run one
    // quarter ahead of their use -- x(q + 1) is issued slot by slot while quarter q is stored, dres(q) while its sums are reduced --
    // so that no memory round trip sits between two barriers.
    asm volatile("s_barrier" ::: "memory");     // every wave is done with the ring before its LDS becomes scratch
    float* stash = reinterpret_cast<float*>(smem) + wave * 2048;         // [8 registers][64 lanes][4 column blocks] per wave: 8 KiB
    float* red = reinterpret_cast<float*>(smem) + R1_WAVES * 2048;       // [4 quarters][14 waves][2 half-waves][8]
    float* red2 = reinterpret_cast<float*>(smem);                        // [7 row groups][512]: dgamma | dbeta partials -- over the stash, which is dead by then
    const int hw = lane >> 5;
    // uniform base pointers + one 32-bit BYTE offset per row (x, dres, dx and dx16 share one row pitch: launch check)
    const float* const xin = p.lnb_x;
    const float* const res = reinterpret_cast<const float*>(p.aux_in);
    const bool has_res = p.aux_in != nullptr;                     // (ln_f: nothing joins the stream behind it)
    float* const out = reinterpret_cast<float*>(p.C);
    bf16_t* const out16 = reinterpret_cast<bf16_t*>(p.aux_out);   // (same row pitch in elements: byte offsets halve)
    const bool has16 = p.aux_out != nullptr;
    const unsigned col0 = (unsigned)(wn * 128 + fr);
    float gm[4], dgam[4], dbet[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { gm[j] = p.lnb_gamma[col0 + j * 32]; dgam[j] = 0.f; dbet[j] = 0.f; }
    const float invC = 1.0f / 256.0f;
    if (!full) {   // rows behind the span hold whatever the unloaded LDS held: zero them in place, once (wave-uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool rok = wrow0 + rg_frag_row(r, lane) < row_end;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][r] = rok ? acc[j][r] : 0.f;
      }
    }
    // accumulator registers 8 .. 15 (quarters 2, 3) wait in LDS
#pragma unroll
    for (int r = 8; r < 16; ++r)
      *reinterpret_cast<float4*>(stash + ((r - 8) * 64 + lane) * 4) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    float xq[2][4][4], mu[2][4], rs[2][4];
    unsigned ro[2][4];
    auto issue_x = [&](int q, int i) __attribute__((always_inline)) {
      const int b = q & 1;
      const int row = wrow0 + i + 8 * q + 4 * hw;      // = wrow0 + rg_frag_row(4 q + i, lane)
      const int rc = (full || row < row_end) ? row : row_end - 1;
      ro[b][i] = ((unsigned)rc * (unsigned)p.ldc + col0) * 4u;
      mu[b][i] = p.lnb_mean[rc]; rs[b][i] = p.lnb_rstd[rc];
#pragma unroll
      for (int j = 0; j < 4; ++j) xq[b][i][j] = rg_ld(xin, ro[b][i], j * 128);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_x(0, i);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = q & 1;
      if (q == 2 && p.chain_W != nullptr) {
        // chained product (below): its first weight stage goes on its way now -- weight slot 1 lies behind the stash and the sums
        const bf16_t* wb = p.chain_W + (long long)8 * wave * p.chain_ldw;
        const long long w_p = 8LL * R1_WAVES * p.chain_ldw;
        const unsigned slot = ldsW + RG_W_BYTES, offw2 = (unsigned)(((lane >> 3) * (int)p.chain_ldw + ((lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7)) * 8) * 2);
        rg_dma16s(wb, offw2, slot + wave * 1024);
        rg_dma16s(wb + w_p, offw2, slot + (wave + R1_WAVES) * 1024);
        if (nwp == 3) rg_dma16s(wb + 2 * w_p, offw2, slot + (wave + 2 * R1_WAVES) * 1024);
      }
      if (q == 2) {   // quarters 0, 1 are out: their accumulators are dead, the other half comes back from LDS
#pragma unroll
        for (int r = 8; r < 16; ++r) {
          const float4 t4 = *reinterpret_cast<const float4*>(stash + ((r - 8) * 64 + lane) * 4);
          acc[0][r] = t4.x; acc[1][r] = t4.y; acc[2][r] = t4.z; acc[3][r] = t4.w;
        }
      }
      // dres of this quarter: in flight while the sums are reduced
      float dr[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dr[i][j] = has_res ? rg_ld(res, ro[b][i], j * 128) : 0.f;
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = acc[j][4 * q + i];
          const float xn = (xq[b][i][j] - mu[b][i]) * rs[b][i];
          const float g = a * gm[j];
          s1 += g; s2 += g * xn;
          dgam[j] += a * xn; dbet[j] += a;
          xq[b][i][j] = xn;
        }
        v[2 * i] = s1; v[2 * i + 1] = s2;
      }
      // pin the column sums HERE: left alone the compiler sinks all 128 updates to the end of the kernel (that is where they are
      // used), which keeps every accumulator and every normalised value alive until then -- 200 spilled registers
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(dgam[j]), "+v"(dbet[j]));
      // halving butterfly over the 32 lanes of the half-wave: lane l ends with the total of value (l >> 2) & 7
      {
        const bool b4 = (lane & 16) != 0, b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
        float w[4], u[2], t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float send = b4 ? v[i] : v[i + 4], mine = b4 ? v[i + 4] : v[i];
          w[i] = mine + __shfl_xor(send, 16, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float send = b3 ? w[i] : w[i + 2], mine = b3 ? w[i + 2] : w[i];
          u[i] = mine + __shfl_xor(send, 8, 64);
        }
        {
          const float send = b2 ? u[0] : u[1], mine = b2 ? u[1] : u[0];
          t = mine + __shfl_xor(send, 4, 64);
        }
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 1, 64);
        if ((lane & 3) == 0) red[((q * R1_WAVES + wave) * 2 + hw) * 8 + ((lane >> 2) & 7)] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the sums are in LDS before any wave reads its partner's
      const float4* mine4 = reinterpret_cast<const float4*>(red + ((q * R1_WAVES + wave) * 2 + hw) * 8);
      const float4* part4 = reinterpret_cast<const float4*>(red + ((q * R1_WAVES + (wave ^ 1)) * 2 + hw) * 8);
      const float4 m0 = mine4[0], m1 = mine4[1], p0 = part4[0], p1 = part4[1];
      const float c1[4] = {(m0.x + p0.x) * invC, (m0.z + p0.z) * invC, (m1.x + p1.x) * invC, (m1.z + p1.z) * invC};
      const float c2[4] = {(m0.y + p0.y) * invC, (m0.w + p0.w) * invC, (m1.y + p1.y) * invC, (m1.w + p1.w) * invC};
      // dx of the quarter, slot by slot; behind every slot the same slot of the NEXT quarter's x goes on its way
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wrow0 + i + 8 * q + 4 * hw;
        if (full || row < row_end) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float o = rs[b][i] * (acc[j][4 * q + i] * gm[j] - c1[i] - xq[b][i][j] * c2[i]) + dr[i][j];
            rg_st(out, ro[b][i], j * 128, o);
            if (has16) rg_st(out16, ro[b][i] >> 1, j * 64, f2bf(o));
          }
        }
        if (q < 3) issue_x(q + 1, i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    RG_T(3);
    // dgamma | dbeta: the two half-waves hold different rows of the same columns; then the 7 row groups through LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) { dgam[j] += __shfl_xor(dgam[j], 32, 64); dbet[j] += __shfl_xor(dbet[j], 32, 64); }
    if (hw == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red2[wm * 512 + col0 + j * 32] = dgam[j];
        red2[wm * 512 + 256 + col0 + j * 32] = dbet[j];
      }
    }
    __syncthreads();
    if (tid < 512) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < R1_GROUPS; ++g) t += red2[g * 512 + tid];
      p.lnb_partial[(long long)blockIdx.x * 512 + tid] = t;
    }
    RG_T(4);
    if (p.chain_W != nullptr) {
      // ---- chained product: chain_C = dx16 chain_W^T (N = K = 256: the c_proj input gradient that follows ln_2's backward) on the
      // rows this workgroup has just written -- they come back from L2 by DMA as four 64-k slabs (the three A slots + weight slot 0),
      // the weight streams through weight slot 1 in four 64-k stages; one launch and one trip of the rows through HBM less
      __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): this wave's dx16 stores have left
      __syncthreads();                            // ... every wave's; and red2 has been read
      const bf16_t* const A2 = reinterpret_cast<const bf16_t*>(p.aux_out);
      // the lane constants of the k loop again, from a laundered lane id: kept alive across the LayerNorm write-out they cost it 14
      // spilled registers
      int ln2 = lane;
      asm volatile("" : "+v"(ln2));
      const int lrow = ln2 >> 3, cg = (ln2 & 7) ^ ((((wave & 1) << 2) + (ln2 >> 4)) & 7);
      const int fr = ln2 & 31, kg = ln2 >> 5, swz = (fr >> 1) & 7;
      unsigned xo[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xo[ks] = (unsigned)(fr * 128 + (((2 * ks + kg) ^ swz) << 4));
      const unsigned offa2 = (unsigned)((lrow * (int)p.ldc + cg * 8) * 2), offw2 = (unsigned)((lrow * (int)p.chain_ldw + cg * 8) * 2);
      const bf16_t* const pa2 = A2 + (long long)row0 * p.ldc;
      // LDS: slabs 0, 1, 2 in the three A slots, slab 3 follows slab 0 into A slot 0 once stage 0 has been multiplied; the weight
      // stages alternate between the two weight slots (stage 0 -> slot 1, prefetched during the write-out above), each issued one
      // stage ahead: the counted vmcnt waits leave exactly the younger transfers in flight
      auto slab = [&](int kk) -> unsigned { return lds0 + (kk % 3) * R1_A_BYTES; };
      auto wslot = [&](int kk) -> unsigned { return ldsW + ((kk & 1) ^ 1) * RG_W_BYTES; };
      auto issue_a2 = [&](int q, int mode, int kk) __attribute__((always_inline)) {
        if (mode == 2) {
          rg_dma16s(pa2 + (long long)8 * q * p.ldc + kk * RG_BK, offa2, slab(kk) + q * 1024);
        } else if (mode == 1) {
          int r = row0 + 8 * q + lrow;
          r = r < row_end ? r : row_end - 1;
          rg_dma16(A2 + (long long)r * p.ldc + kk * RG_BK + cg * 8, slab(kk) + q * 1024);
        }
      };
      auto issue_w2 = [&](int kk) __attribute__((always_inline)) {
        const bf16_t* wb = p.chain_W + (long long)8 * wave * p.chain_ldw + kk * RG_BK;
        const long long w_p = 8LL * R1_WAVES * p.chain_ldw;
        const unsigned slot = wslot(kk);
        rg_dma16s(wb, offw2, slot + wave * 1024);
        rg_dma16s(wb + w_p, offw2, slot + (wave + R1_WAVES) * 1024);
        if (nwp == 3) rg_dma16s(wb + 2 * w_p, offw2, slot + (wave + 2 * R1_WAVES) * 1024);
      };
      auto wait_vm = [&](int n) __attribute__((always_inline)) {   // s_waitcnt vmcnt(n), n = 0 .. 5 (the count is an immediate)
        if (n >= 5) __builtin_amdgcn_s_waitcnt(0x0f75);
        else if (n == 4) __builtin_amdgcn_s_waitcnt(0x0f74);
        else if (n == 3) __builtin_amdgcn_s_waitcnt(0x0f73);
        else if (n == 2) __builtin_amdgcn_s_waitcnt(0x0f72);
        else if (n == 1) __builtin_amdgcn_s_waitcnt(0x0f71);
        else __builtin_amdgcn_s_waitcnt(0x0f70);
      };
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) { issue_a2(q0, a0, kk); issue_a2(q1, a1, kk); }
      issue_w2(1);   // (weight stage 0 has been on its way since the third quarter of the write-out)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // in flight behind the wait: what was issued after this stage's operands -- stage 1: W(2) + slab 3; stage 2: W(3)
        wait_vm(kk == 1 ? nwp + nA : (kk == 2 ? nwp : 0));
        __builtin_amdgcn_s_barrier();
        const unsigned char* SA = smem + (kk % 3) * R1_A_BYTES;
        const unsigned char* SW = smem + 3 * R1_A_BYTES + ((kk & 1) ^ 1) * RG_W_BYTES;
        bf16x8 fa[2], fw[2][4];
        fa[0] = *reinterpret_cast<const bf16x8*>(SA + a_row + xo[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[0][j] = *reinterpret_cast<const bf16x8*>(SW + w_row + j * 4096 + xo[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int cur = ks & 1, nxt = cur ^ 1;
          if (ks < 3) {
            fa[nxt] = *reinterpret_cast<const bf16x8*>(SA + a_row + xo[ks + 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) fw[nxt][j] = *reinterpret_cast<const bf16x8*>(SW + w_row + j * 4096 + xo[ks + 1]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur], fw[cur][j], acc[j], 0, 0, 0);
        }
        if (kk < 2) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave has read this stage's weight slot (and slab 0): they may be refilled
          issue_w2(kk + 2);
          if (kk == 0) { issue_a2(q0, a0, 3); issue_a2(q1, a1, 3); }
        }
      }
      bf16_t* out2 = p.chain_C + wn * 128 + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wrow0 + rg_frag_row(r, ln2);
        if (full || row < row_end) {
#pragma unroll
          for (int j = 0; j < 4; ++j) out2[(long long)row * p.chain_ldc + j * 32] = f2bf(acc[j][r]);
        }
      }
      RG_T(5);
    }
    RG_TDUMP();
    return;
  } else if (EPI == EPI_RES_F32) {
    const float* res = reinterpret_cast<const float*>(p.aux_in) + wn * 128 + fr;
    float* out = reinterpret_cast<float*>(p.C) + wn * 128 + fr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // 8 rows at a time: 32 residual values in flight per lane (128-VGPR budget: 4 waves per SIMD)
      float x[8][4];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = wrow0 + rg_frag_row(8 * h + r, lane), rc = (full || row < row_end) ? row : row_end - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[r][j] = res[(long long)rc * p.ld_aux + j * 32];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = wrow0 + rg_frag_row(8 * h + r, lane);
        if (full || row < row_end) {
#pragma unroll
          for (int j = 0; j < 4; ++j) out[(long long)row * p.ldc + j * 32] = acc[j][8 * h + r] + x[r][j];
        }
      }
    }
  } else {
    bf16_t* out = reinterpret_cast<bf16_t*>(p.C) + wn * 128 + fr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wrow0 + rg_frag_row(r, lane);
      if (full || row < row_end) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(long long)row * p.ldc + j * 32] = f2bf(acc[j][r]);
      }
    }
  }
  RG_T(3);
  RG_TDUMP();
}

template <int EPI>
static int launch_ring1_t(const GemmArgs& a, int R, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_ring1_kernel<EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R1_LDS_BYTES);
    if (e != hipSuccess) {
      coati_set_error("gemm_ring1: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(a.M, R)), dim3(64 * R1_WAVES), R1_LDS_BYTES, s, a, R);
  COATI_LAUNCH_CHECK("gemm_ring1");
  return COATI_OK;
}


// ---- one round, 64 x 128 wave tiles (round 3): 57 345 .. 65 536 rows ------------------------------------------------------
// The 32 x 128 wave tile of the kernels above reads 5 KiB of LDS fragments per 4 MFMAs -- 160 B/clk per CU at full matrix-core
// rate against the 128 the LDS delivers.  A 64 x 128 tile reads 6 KiB per 8 MFMAs (96 B/clk): 8 waves = 4 row groups of 64 x 2 column
// halves, 128 accumulator registers per wave (2 waves per SIMD), spans of <= 256 rows, A ring of 3 x 32 KiB + weight ring of 2 x 32 KiB
// = exactly the 160 KiB of LDS.  Same DMA / wait choreography as gemm_ring1_kernel (4 A pieces + 4 weight pieces per wave and stage).
#define R2W_GROUPS 4
#define R2W_BR (64 * R2W_GROUPS)
#define R2W_WAVES (2 * R2W_GROUPS)
#define R2W_A_BYTES (R2W_BR * RG_BK * 2)                                // 32 KiB
#define R2W_LDS_BYTES (3 * R2W_A_BYTES + 2 * RG_W_BYTES)                // 163,840 B

__device__ __forceinline__ void r2w_wait_vm(int n) {   // s_waitcnt vmcnt(n), n = 0 .. 4
  if (n >= 4) __builtin_amdgcn_s_waitcnt(0x0f74);
  else if (n == 3) __builtin_amdgcn_s_waitcnt(0x0f73);
  else if (n == 2) __builtin_amdgcn_s_waitcnt(0x0f72);
  else if (n == 1) __builtin_amdgcn_s_waitcnt(0x0f71);
  else __builtin_amdgcn_s_waitcnt(0x0f70);
}

template <int EPI>
__global__ __launch_bounds__(64 * R2W_WAVES, 1) void gemm_ring1w_kernel(GemmArgs p, int R) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = p.K / RG_BK;
  const int row0 = blockIdx.x * R;
  if (row0 >= p.M) return;
  const int row_end = row0 + R < p.M ? row0 + R : p.M, nvalid = row_end - row0;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u8*)smem);
  const unsigned ldsW = lds0 + 3 * R2W_A_BYTES;

  // DMA pieces of 8 rows x 128 B: A pieces {wave + 8 i}, weight pieces {wave + 8 i}, i = 0 .. 3 (all of one parity)
  const int lrow = lane >> 3;
  const int cg = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const unsigned offa = (unsigned)((lrow * (int)p.lda + cg * 8) * 2);
  const unsigned offw = (unsigned)((lrow * (int)p.ldb + cg * 8) * 2);
  int amode[4], nA = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = wave + R2W_WAVES * i;
    amode[i] = 8 * q + 8 <= nvalid ? 2 : (8 * q < nvalid ? 1 : 0);
    nA += amode[i] != 0;
  }
  const bool ROT = (EPI != EPI_RES_F32) && nk >= 8;
  const int kk0 = ROT ? (int)(blockIdx.x % (unsigned)nk) : 0;
  const bf16_t* const pa = A + (long long)row0 * p.lda;
  auto issue_a_piece = [&](int i, int kk, unsigned slot) __attribute__((always_inline)) {
    const int q = wave + R2W_WAVES * i;
    if (amode[i] == 2) {
      rg_dma16s(pa + (long long)8 * q * p.lda + kk * RG_BK, offa, slot + q * 1024);
    } else if (amode[i] == 1) {
      int r = row0 + 8 * q + lrow;
      r = r < row_end ? r : row_end - 1;
      rg_dma16(A + (long long)r * p.lda + kk * RG_BK + cg * 8, slot + q * 1024);
    }
  };
  auto issue_w2 = [&](int half, int kk, unsigned slot) __attribute__((always_inline)) {   // weight pieces 2 half, 2 half + 1 of this wave
    const bf16_t* wb = p.B + (long long)8 * (wave + R2W_WAVES * 2 * half) * p.ldb + kk * RG_BK;
    const long long w_p = 8LL * R2W_WAVES * p.ldb;
    rg_dma16s(wb, offw, slot + (wave + R2W_WAVES * 2 * half) * 1024);
    rg_dma16s(wb + w_p, offw, slot + (wave + R2W_WAVES * (2 * half + 1)) * 1024);
  };
  auto kwrap = [&](int k) { return k >= nk ? k - nk : k; };

  const int fr = lane & 31, kg = lane >> 5, swz = (fr >> 1) & 7;
  unsigned xo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xo[ks] = (unsigned)(fr * 128 + (((2 * ks + kg) ^ swz) << 4));
  const unsigned a_row = (unsigned)(wm * 64 * 128), w_row = (unsigned)(wn * 128 * 128);

  f32x16 acc[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float b = p.bias != nullptr ? p.bias[wn * 128 + j * 32 + fr] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][j][r] = b; acc[1][j][r] = b; }
  }

  // prologue: A(0), W(0), A(1)
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_a_piece(i, kk0, lds0);
  issue_w2(0, kk0, ldsW);
  issue_w2(1, kk0, ldsW);
  if (nk > 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a_piece(i, kwrap(kk0 + 1), lds0 + R2W_A_BYTES);
  }
  int sa = 0, sw = 0;
  int kk1 = kwrap(kk0 + 1), kk2 = kwrap(kk1 + 1);
  for (int s = 0; s < nk; ++s) {
    r2w_wait_vm(s + 1 < nk ? nA : 0);
    __builtin_amdgcn_s_barrier();
    const unsigned char* SA = smem + sa * R2W_A_BYTES;
    const unsigned char* SW = smem + 3 * R2W_A_BYTES + sw * RG_W_BYTES;
    const int sa2 = sa == 0 ? 2 : sa - 1;                        // (sa + 2) % 3
    {
      bf16x8 fa[2][2], fw[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(SA + a_row + i * 4096 + xo[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[0][j] = *reinterpret_cast<const bf16x8*>(SW + w_row + j * 4096 + xo[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        // issue order inside the stage: W(s + 1) at k steps 0 and 1, then the A pieces of stage s + 2 at k steps 2 and 3
        if (ks < 2) { if (s + 1 < nk) issue_w2(ks, kk1, ldsW + (sw ^ 1) * RG_W_BYTES); }
        else if (s + 2 < nk) { issue_a_piece(2 * (ks - 2), kk2, lds0 + sa2 * R2W_A_BYTES); issue_a_piece(2 * (ks - 2) + 1, kk2, lds0 + sa2 * R2W_A_BYTES); }
        if (ks < 3) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[nxt][i] = *reinterpret_cast<const bf16x8*>(SA + a_row + i * 4096 + xo[ks + 1]);
#pragma unroll
          for (int j = 0; j < 4; ++j) fw[nxt][j] = *reinterpret_cast<const bf16x8*>(SW + w_row + j * 4096 + xo[ks + 1]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fw[cur][j], acc[i][j], 0, 0, 0);
        if (ks < 3) {
#pragma unroll
          for (int g = 0; g < 6; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
      }
    }
    sa = sa == 2 ? 0 : sa + 1;
    sw ^= 1;
    kk1 = kk2;
    kk2 = kwrap(kk2 + 1);
  }

  // ---- write-out straight from the accumulator layout
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int wrow0 = row0 + wm * 64 + i * 32;
    const bool full = wm * 64 + (i + 1) * 32 <= nvalid;   // wave-uniform
    if (EPI == EPI_RES_F32) {
      const float* res = reinterpret_cast<const float*>(p.aux_in) + wn * 128 + fr;
      float* out = reinterpret_cast<float*>(p.C) + wn * 128 + fr;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float x[8][4];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int row = wrow0 + rg_frag_row(8 * h + r, lane), rc = (full || row < row_end) ? row : row_end - 1;
#pragma unroll
          for (int j = 0; j < 4; ++j) x[r][j] = res[(long long)rc * p.ld_aux + j * 32];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int row = wrow0 + rg_frag_row(8 * h + r, lane);
          if (full || row < row_end) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(long long)row * p.ldc + j * 32] = acc[i][j][8 * h + r] + x[r][j];
          }
        }
      }
    } else {
      bf16_t* out = reinterpret_cast<bf16_t*>(p.C) + wn * 128 + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wrow0 + rg_frag_row(r, lane);
        if (full || row < row_end) {
#pragma unroll
          for (int j = 0; j < 4; ++j) out[(long long)row * p.ldc + j * 32] = f2bf(acc[i][j][r]);
        }
      }
    }
  }
}

template <int EPI>
static int launch_ring1w_t(const GemmArgs& a, int R, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_ring1w_kernel<EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R2W_LDS_BYTES);
    if (e != hipSuccess) {
      coati_set_error("gemm_ring1w: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(a.M, R)), dim3(64 * R2W_WAVES), R2W_LDS_BYTES, s, a, R);
  COATI_LAUNCH_CHECK("gemm_ring1w");
  return COATI_OK;
}

// EPI_LNBWD exists in the one-round kernel only (the packed batch: 40 961 .. 57 344 rows); every other size keeps the two launches
// rows from which the one-round forms run (probe knob COATI_RING1_MINROWS; default: more than one round of 160-row blocks)
static int ring1_min_rows() {
  static const int v = []() { const char* e = getenv("COATI_RING1_MINROWS"); return e ? atoi(e) : 256 * 160 + 1; }();
  return v;
}

bool gemm_ring_lnbwd_supported(const GemmArgs& a, int* nwg) {
  static const bool off = getenv("COATI_NO_LNBWD_FUSE") != nullptr;   // A/B switch: bf16 product + stand-alone LayerNorm backward
  if (off || a.m_dev || a.N != 256 || a.K % RG_BK != 0 || a.K < 256) return false;
  if (130LL * a.lda >= (1LL << 30) || 260LL * a.ldb >= (1LL << 30)) return false;
  const int R = cdiv(cdiv(a.M, 256), 8) * 8;
  if (!(R <= R1_BR && a.M >= ring1_min_rows())) return false;
  if (nwg) *nwg = cdiv(a.M, R);
  return true;
}

int launch_gemm_ring256(const GemmArgs& a, int epi, hipStream_t s) {
  if (epi == EPI_LNBWD) {
    int nwg = 0;
    COATI_CHECK_SHAPE(gemm_ring_lnbwd_supported(a, &nwg), "gemm_ring: EPI_LNBWD needs N = 256 and 40 961 .. 57 344 rows (M=%d N=%d K=%d)", a.M, a.N, a.K);
    COATI_CHECK_ARG(a.lnb_x && a.lnb_mean && a.lnb_rstd && a.lnb_gamma && a.lnb_partial && a.C && a.bias == nullptr, "gemm_ring: EPI_LNBWD operands missing");
    COATI_CHECK_ARG(a.chain_W == nullptr || (a.aux_out != nullptr && a.chain_C != nullptr && a.chain_ldw % 8 == 0 && a.chain_ldc % 8 == 0 && 260LL * a.chain_ldw < (1LL << 30)),
                    "gemm_ring: EPI_LNBWD chained product needs the bf16 copy (aux_out), chain_C and aligned pitches");
    COATI_CHECK_SHAPE(a.lnb_ldx == a.ldc && (a.aux_in == nullptr || a.ld_aux == a.ldc) && ((long long)a.M + 8) * a.ldc * 4 < (1LL << 32), "gemm_ring: EPI_LNBWD wants one row pitch for x / dres / dx (ldx=%lld ld_aux=%lld ldc=%lld)", a.lnb_ldx, a.ld_aux, a.ldc);
    return launch_ring1_t<EPI_LNBWD>(a, cdiv(cdiv(a.M, 256), 8) * 8, s);
  }
  // block height: rows the busiest of the 256 persistent workgroups walks = rounds x block rows; ties go to the 160-row form
  // (less weight re-streaming per row).
  // more than one round of 160-row blocks, at most 224 rows per CU: the one-round kernel
  const int R = cdiv(cdiv(a.M, 256), 8) * 8;
  // 57 345 .. 65 536 rows (spans of 225 .. 256 rows): the 8-wave form with 64 x 128 wave tiles.  Its time hardly depends on the rows
  // (K = 1024, bf16 out: 39.4 / 38.8 / 39.2 us at 40 960 / 50 000 / 65 536 rows: two waves per SIMD leave the per-stage latency
  // exposed), so it only pays where the alternative is two rounds of 128-row blocks (65 536 rows: 42.6 us) -- at 50 000 rows the
  // 14-wave kernel below takes 33.8
  if (R <= R2W_BR && R > R1_BR) return epi == EPI_RES_F32 ? launch_ring1w_t<EPI_RES_F32>(a, R, s) : launch_ring1w_t<EPI_BF16>(a, R, s);
  if (R <= R1_BR && a.M >= ring1_min_rows())
    return epi == EPI_RES_F32 ? launch_ring1_t<EPI_RES_F32>(a, R, s) : launch_ring1_t<EPI_BF16>(a, R, s);
  const long long busiest160 = (long long)cdiv(cdiv(a.M, 160), 256) * 160, busiest128 = (long long)cdiv(cdiv(a.M, 128), 256) * 128;
  const bool small = busiest128 < busiest160;
  if (small) return epi == EPI_RES_F32 ? launch_ring_t<EPI_RES_F32, 4>(a, s) : launch_ring_t<EPI_BF16, 4>(a, s);
  return epi == EPI_RES_F32 ? launch_ring_t<EPI_RES_F32, 5>(a, s) : launch_ring_t<EPI_BF16, 5>(a, s);
}
