// Row-block ("A-stationary") bf16 GEMM for K = 256:  C[M,N] = epilogue(A[M,256] * B[N,256]^T + bias).
//
// Every Linear that reads the 256-wide stream has a tiny weight matrix (<= 512 KB, L2 resident) and a huge M (81,920
// rows): its arithmetic intensity is below the HBM ridge, so the tiled kernel's per-workgroup load -> MFMA -> store
// chain (two chains per CU) leaves the memory system half idle.  Here one workgroup per CU owns a block of rows for the
// whole kernel:
//   * each wave keeps its 32 x 256 slab of A as 16 MFMA A-fragments in registers (A is read from HBM exactly once);
//   * the weight tile [64 cols][256] streams through a double-buffered LDS image (rows padded to 528 B: conflict-free
//     ds_read_b128), the next tile's loads in flight behind the current tile's 32 MFMAs per wave, one barrier per tile;
//     the MFMA loop reads its weight fragments RB_PD - 1 k-steps ahead (explicit software pipeline);
//   * the bias is folded into the accumulator initialisation (next tile's two values prefetched with the weights), and
//     the rotary tables of the wave's 32 rows are staged in LDS once: the epilogue issues no global loads;
//   * every wave transposes its 32 x 64 accumulator block, 16 rows at a time, through a PRIVATE LDS region (no
//     workgroup barrier) and runs the fused epilogue (gemm_epi.h) on 8 consecutive columns per lane: 128-B row
//     segments, 16-B stores;
//   * W waves per workgroup is chosen on the host so that the grid is a whole number of 256-CU rounds.
// Probe with the phase ablations that led to this shape: tools/probes/rb_probe.hip.
#include <cstdlib>
#include "gemm_epi.h"

// Wave priority by phase: a wave raises its priority for its MFMA phase and drops it for the epilogue, so that on a SIMD
// the matrix core is fed first and the (issue-bound) epilogues of the other waves fill the gaps -- measured per step at
// M = 81,920: FC1 + GELU' 3.38 -> 3.20 ms, lm_head forward 0.73 -> 0.67, dlogits 0.70 -> 0.67, QKV 2.37 -> 2.31 (the reverse,
// RB_PRIO=1, is slower; RB_PRIO=0 switches it off)
#ifndef RB_PRIO
#define RB_PRIO 2
#endif
#ifndef RB_PRIO_HI
#define RB_PRIO_HI 3
#endif
#define RB_K 256
// One workgroup per CU on 64-column weight tiles (template BN, MAXW = the most waves an instantiation runs with: 7 / 10 / 12).
// (Two 5-wave workgroups per CU on 32-column tiles were measured in round 1 -- 51 -> 73 us -- and removed in round 3.)
#define RB_EFLOATS_MAX (16 * 68)          // per-wave transpose region, floats (16 rows x (BN + 4))
#define RB_ROPE_FLOATS (32 * 16)          // 2 KiB per wave: [32 rows][8 cos | 8 sin]
#define RB_AUX_BYTES 4096                 // per wave: the 32 x 64 bf16 block of saved pre-activations of the current tile
#define RB_MAX_W 10
#define RB_HALF_W 12                      // "8 + 4" shape: 8 waves of 32 rows + 4 waves of 16 rows = the same 320 rows per workgroup
#define RB_FEW_W 7                        // instantiation for 5 .. 7 waves per workgroup
#ifndef RB_PD
#define RB_PD 3                           // LDS read pipeline depth of the MFMA loop
#endif

// Probe build (-DCOATI_RB_TRACE, tools/probes/rb_trace.py): shader-clock totals per phase for the waves of the first 16
// workgroups of the LAST launch: [wg][wave][prologue, mfma, wait for the next tile / the staged operands, epilogue, barrier].
#ifdef COATI_RB_TRACE
__device__ unsigned long long rb_trace_buf[16 * RB_HALF_W * 8];
extern "C" int coati_rb_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rb_trace_buf), sizeof(rb_trace_buf)) == hipSuccess ? 0 : -3;
}
#define RB_T0() unsigned long long rb_t_last = __builtin_amdgcn_s_memtime(), rb_t_acc[5] = {0, 0, 0, 0, 0}
#define RB_T(i) do { const unsigned long long rb_t_now = __builtin_amdgcn_s_memtime(); rb_t_acc[i] += rb_t_now - rb_t_last; rb_t_last = rb_t_now; } while (0)
#define RB_TDUMP() do { if (blockIdx.x < 16 && lane == 0) { for (int i = 0; i < 5; ++i) rb_trace_buf[(blockIdx.x * RB_HALF_W + wave) * 8 + i] = rb_t_acc[i]; } } while (0)
#else
#define RB_T0() do { } while (0)
#define RB_T(i) do { } while (0)
#define RB_TDUMP() do { } while (0)
#endif

template <typename F>
__device__ __forceinline__ void rb_call_restrict(F&& f, int jt, int jn, const bf16_t* __restrict__ cur, bf16_t* __restrict__ nxt) {
  f(jt, jn, cur, nxt);
}

template <int EPI, int RB_BN, int MAXW, bool LN = false>
__global__ __launch_bounds__(64 * MAXW, (640 / (64 * MAXW)) > 0 ? (640 / (64 * MAXW)) : 1) void gemm_rb256_kernel(GemmArgs p, int W, int half_from, int rot) {
  // W waves; waves >= half_from own 16 rows instead of 32 (half_from >= W: none).  The kernel is issue-bound per SIMD (MFMA +
  // epilogue VALU of the waves that share it: tools/probes/rb_trace.py), and 10 full waves sit 3 / 3 / 2 / 2 on the four
  // SIMDs -- a fifth of the wave time went into the tile barrier.  8 full + 4 half waves put 2 + 1/2 slabs on every SIMD:
  // a half wave multiplies a whole 32-row MFMA block (rows 16..31 duplicate rows 0..15) but writes out only 16 rows.
  const int rows_wg = (half_from < W ? half_from : W) * 32 + (half_from < W ? (W - half_from) * 16 : 0);
  if (p.m_dev) {   // data-dependent row count (<= the M the grid was sized for): workgroups past the end leave before any barrier
    p.M = *p.m_dev;
    if ((int)blockIdx.x * rows_wg >= p.M) return;
  }
  constexpr int NACC = RB_BN / 32;                 // 32-column accumulator blocks per wave per tile
  constexpr int RB_TILE_HALFS = RB_BN * RB_K;      // [BN cols][256 k] bf16, unpadded, chunk-swizzled
  constexpr int RB_EPITCH = RB_BN + 4;             // floats per row of the per-wave transpose region
  constexpr int RB_EFLOATS = 16 * RB_EPITCH;       // 16 rows x BN columns
  constexpr int CGS = RB_BN / 8;                   // 8-column groups per row
  constexpr int TPH = 16 * CGS / 64;               // epilogue tasks per lane per 16-row half (2 or 1)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* const Bs = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* const Es = reinterpret_cast<float*>(smem + 2 * RB_TILE_HALFS * 2) + wave * RB_EFLOATS;
  float* const Rs = reinterpret_cast<float*>(smem + 2 * RB_TILE_HALFS * 2) + W * RB_EFLOATS + wave * RB_ROPE_FLOATS;
  constexpr int AUXB = (EPI == EPI_MUL_AUX) ? RB_AUX_BYTES / 2 : RB_AUX_BYTES;   // (one-byte codes: half the block)
  unsigned char* const Xs = smem + 2 * RB_TILE_HALFS * 2 + (size_t)W * RB_EFLOATS * 4 + wave * AUXB;   // DGELU / DSILU / MUL_AUX
  constexpr bool AUX = (EPI == EPI_DGELU || EPI == EPI_DSILU || EPI == EPI_MUL_AUX);
  constexpr bool EDGE = (EPI == EPI_EDGE_DPRE);   // per-column constants of the tile staged in Rs: [64 w1c | 64 b1]
  const bool halfw = wave >= half_from;
  const int m0 = blockIdx.x * rows_wg + (halfw ? half_from * 32 + (wave - half_from) * 16 : wave * 32);
  const int nrow = halfw ? 16 : 32;               // rows this wave owns
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  const int fra = fr & (nrow - 1);                // row this lane LOADS (a half wave's rows 16..31 repeat rows 0..15)
  RB_T0();

  // resident A slab: A-operand fragments, lane (i = lane&31 -> row, kg = lane>>5) holds k = ks*16 + kg*8 .. +7
  bf16x8 af[16];
  if constexpr (!LN) {
    const int rc = (m0 + fra) < p.M ? (m0 + fra) : p.M - 1;
    const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.A) + (long long)rc * p.lda + fk;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 16);
  } else {
    // LayerNorm fused into the slab load (reference basic_transformer.py:165-173: x + attn(ln_1(x)), ... mlp(ln_2(x))):
    // the lane pair (fr, kg = 0 / 1) holds one f32 row of 256 = 2 x 16 chunks of 8; two-pass statistics in registers
    // (same formulas as ln_fwd_kernel), normalised row -> bf16 fragments + the saved copy the weight gradient reads
    const int row = m0 + fra, rc = row < p.M ? row : p.M - 1;
    const float* xp = p.ln_x + (long long)rc * p.ln_ldx + fk;
    float xf[16][8];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float4 x0 = *reinterpret_cast<const float4*>(xp + ks * 16), x1 = *reinterpret_cast<const float4*>(xp + ks * 16 + 4);
      xf[ks][0] = x0.x; xf[ks][1] = x0.y; xf[ks][2] = x0.z; xf[ks][3] = x0.w;
      xf[ks][4] = x1.x; xf[ks][5] = x1.y; xf[ks][6] = x1.z; xf[ks][7] = x1.w;
    }
    // gamma | beta go through the second weight-tile buffer (idle until the first tile prefetches into it): their loads are
    // in flight together with the row loads, and the normalisation below reads them as LDS broadcasts
    float* const GB = reinterpret_cast<float*>(Bs + RB_TILE_HALFS);
    if (tid < 128) {
      const float* src = tid < 64 ? p.ln_gamma + 4 * tid : p.ln_beta + 4 * (tid - 64);
      *reinterpret_cast<float4*>(GB + 4 * tid) = *reinterpret_cast<const float4*>(src);
    }
    float sm = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) sm += xf[ks][i];
    sm += __shfl_xor(sm, 32, 64);
    const float mean = sm / (float)RB_K;
    float q2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = xf[ks][i] - mean;
        q2 += d * d;
      }
    q2 += __shfl_xor(q2, 32, 64);
    const float rstd = 1.0f / sqrtf(q2 / (float)RB_K + 1e-5f);
    if (lane < 32 && fr < nrow && row < p.M) {
      p.ln_mean[row] = mean;
      p.ln_rstd[row] = rstd;
    }
    bf16_t* op = reinterpret_cast<bf16_t*>(const_cast<void*>(p.A)) + (long long)rc * p.lda + fk;
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float4 g0 = *reinterpret_cast<const float4*>(GB + fk + ks * 16), g1 = *reinterpret_cast<const float4*>(GB + fk + ks * 16 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(GB + RB_K + fk + ks * 16), b1 = *reinterpret_cast<const float4*>(GB + RB_K + fk + ks * 16 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (xf[ks][i] - mean) * rstd * g[i] + bt[i];
      const uint4 u = pack8(o);
      af[ks] = __builtin_bit_cast(bf16x8, u);
      if (row < p.M && fr < nrow) *reinterpret_cast<uint4*>(op + ks * 16) = u;
    }
  }
  if (EPI == EPI_QKV_ROPE) {
    // lane -> (row = lane >> 1, cos | sin): 8 floats each
    const int r = lane >> 1;
    int t;
    if (p.rope_row_t != nullptr) {   // packed rows: the position comes from the row map (rows past M re-read the last one)
      const int mr = m0 + r < p.M ? m0 + r : p.M - 1;
      t = p.rope_row_t[mr];
    } else {
      t = (m0 + r) % p.rope_T;
    }
    const float* src = ((lane & 1) ? p.rope_sin : p.rope_cos) + t * 16;
    const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
    *reinterpret_cast<float4*>(Rs + r * 16 + (lane & 1) * 8) = x0;
    *reinterpret_cast<float4*>(Rs + r * 16 + (lane & 1) * 8 + 4) = x1;
  }

  const int ntiles = (p.N + RB_BN - 1) / RB_BN;
  // Weight tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write).  One wave
  // instruction fills 1 KiB = two 512-B tile rows in lane order, so rows are unpadded and the 16-B chunk c of tile row r
  // is stored at chunk position c ^ (r & 31): the 32 lanes of a fragment read (same k chunk, 32 different rows) then
  // hit 32 different 16-B columns (conflict-free ds_read_b128).  The swizzle is applied on the GLOBAL side: the lane
  // that writes position q of row r fetches chunk q ^ (r & 31).  Weight rows >= N are clamped to row N-1 (their output
  // columns are never stored).
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;
  auto load_tile = [&](int n0, bf16_t* S) {
    // RB_BN / 2 pieces over the W waves: 4 turns cover a 64-column tile from 8 waves on; the instantiation for 5 .. 7 waves
    // (RB_FEW_W: 6-7 slabs per workgroup, what packed rows bring) takes up to 7.  Compile-time turn count: a longer loop
    // with skipped turns in the 8 .. 12-wave kernels cost 10 % of the step (the uniform branches split the tile's schedule).
    constexpr int TURNS = MAXW >= 8 ? 4 : 7;
#pragma unroll
    for (int i = 0; i < TURNS; ++i) {
      const int k = wave + W * i;            // wave-uniform: which 1-KiB piece (two rows) of the tile
      if (k < RB_BN / 2) {
        const int r = 2 * k + (lane >> 5), q = lane & 31;
        const int g = n0 + r, gc = g < p.N ? g : p.N - 1;
        const bf16_t* src = p.B + (long long)gc * p.ldb + ((q ^ (r & 31)) * 8);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(S + k * 512), 16, 0, 0);
      }
    }
  };
  // saved pre-activations of this wave's 32 x 64 output block, DMA'd in epilogue-task order (task = lane + 64 i <->
  // row (task >> 3), columns (task & 7) * 8 .. +7): the epilogue reads its 16 B back with one ds_read_b128.  The block
  // is consumed after the MFMA phase of the same tile, behind the same vmcnt(0) as the next weight tile.
  auto load_aux = [&](int n0) {
    if constexpr (EPI == EPI_MUL_AUX) {
      // one-byte codes: a 16-B DMA piece = 16 columns; piece p = lane + 64 i <-> row p / PPR, columns (p % PPR) * 16 .. +15;
      // the epilogue task (row, 8-column group cg) reads its 8 B at piece (row * PPR + cg / 2), half cg & 1
      constexpr int PPR = RB_BN / 16;
#pragma unroll
      for (int i = 0; i < 32 * PPR / 64; ++i) {
        const int pc = lane + 64 * i, row = m0 + pc / PPR, col = n0 + (pc % PPR) * 16;
        const int rc = row < p.M ? row : p.M - 1, cc = col + 16 <= p.N ? col : 0;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.aux_in) + (long long)rc * p.ld_aux + cc;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(Xs + i * 1024), 16, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2 * TPH; ++i) {
      const int t = lane + 64 * i, row = m0 + t / CGS, col = n0 + (t % CGS) * 8;
      const int rc = row < p.M ? row : p.M - 1, cc = col + 8 <= p.N ? col : 0;
      const bf16_t* src = reinterpret_cast<const bf16_t*>(p.aux_in) + (long long)rc * p.ld_aux + cc;
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(Xs + i * 1024), 16, 0, 0);
    }
  };
  auto bias_at = [&](int col) { return p.bias[col < p.N ? col : p.N - 1]; };
  const bool has_bias = p.bias != nullptr;
  GemmArgs q = p;
  q.bias = nullptr;   // folded into the accumulator initialisation below

  // rot: every workgroup walks the weight tiles from its own starting tile instead of all 256 reading the same 32-KiB tile
  // out of the L2 and writing the same 128-B column of their power-of-two pitched output rows at the same time.  Every tile is
  // a complete sum over k and the CE epilogues index their partials by tile, so the order does not enter any result.
  const int j0 = rot ? (int)(blockIdx.x % (unsigned)ntiles) : 0;
  load_tile(j0 * RB_BN, Bs);
  float bz[NACC], bn[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) { bz[a] = has_bias ? bias_at(j0 * RB_BN + 32 * a + fr) : 0.f; bn[a] = 0.f; }
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();
  RB_T(0);

  const int sw = fr & 31, hk = lane >> 5;
  auto tile = [&](int jt, int jn, const bf16_t* cur, bf16_t* nxt) {   // jt: this tile, jn: the one to prefetch
    // prefetch for tile j + 1 (past the last tile: a clamped, unused re-read of the last rows).  Every wave finished
    // reading that buffer before the barrier that ended the previous iteration.
#if RB_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#elif RB_PRIO == 2
    __builtin_amdgcn_s_setprio(RB_PRIO_HI);
#endif
    load_tile(jn * RB_BN, nxt);
    if constexpr (AUX) load_aux(jt * RB_BN);
    if (has_bias) {
#pragma unroll
      for (int a = 0; a < NACC; ++a) bn[a] = bias_at(jn * RB_BN + 32 * a + fr);
    }
    float ew = 0.f, eb = 0.f;
    if constexpr (EDGE) {   // lane = column of this tile (clamped); written to LDS after the MFMA phase
      const int c = jt * RB_BN + (lane % RB_BN), cc = c < p.N ? c : p.N - 1;
      ew = p.w1c[(long long)cc * p.w1c_stride];
      eb = p.b1[cc];
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = bz[a];
    {
      const bf16_t* wp = cur + fr * RB_K;
      bf16x8 wf[RB_PD][NACC];
#pragma unroll
      for (int d = 0; d < RB_PD - 1; ++d)
#pragma unroll
        for (int a = 0; a < NACC; ++a) wf[d][a] = *reinterpret_cast<const bf16x8*>(wp + a * 32 * RB_K + (((2 * d + hk) ^ sw) * 8));
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks + RB_PD - 1 < 16) {
          const int kn = ks + RB_PD - 1;
#pragma unroll
          for (int a = 0; a < NACC; ++a) wf[kn % RB_PD][a] = *reinterpret_cast<const bf16x8*>(wp + a * 32 * RB_K + (((2 * kn + hk) ^ sw) * 8));
        }
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], wf[ks % RB_PD][a], acc[a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs (the scheduler sinks them otherwise)
      }
    }
    RB_T(1);
    // The next tile had the whole MFMA phase to land.  Waiting HERE -- before this tile's stores are issued -- lets the
    // stores stay in flight across the barrier and through the next MFMA phase (vmcnt completes in order).
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    RB_T(2);
#if RB_PRIO == 1
    __builtin_amdgcn_s_setprio(2);   // epilogue at raised priority
#elif RB_PRIO == 2
    __builtin_amdgcn_s_setprio(0);   // epilogue at low priority (MFMA phases first)
#endif
    if constexpr (EDGE) {   // read back after the lgkmcnt(0) + wave barrier below
      if (lane < RB_BN) { Rs[lane] = ew; Rs[64 + lane] = eb; }
    }
    // wave-private transpose, rows 0-15 then 16-31 of the slab (accumulator registers 0-7 / 8-15):
    // (lane = column, registers = rows) -> rows of 64 contiguous columns
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (hf == 1 && halfw) break;   // (wave-uniform) a half wave has no rows 16..31
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = hf * 8 + rr;
        const int row = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);   // 0..15 within this half
#pragma unroll
        for (int a = 0; a < NACC; ++a) Es[row * RB_EPITCH + 32 * a + fr] = acc[a][r];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
      auto task = [&](int i) {
        const int t = lane + 64 * i, rl = t / CGS, cg = t % CGS;
        const int row = hf * 16 + rl;
        float v[8];
        const float4 c0 = *reinterpret_cast<const float4*>(Es + rl * RB_EPITCH + cg * 8);
        const float4 c1 = *reinterpret_cast<const float4*>(Es + rl * RB_EPITCH + cg * 8 + 4);
        v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
        const void* staged = nullptr;
        if constexpr (EPI == EPI_QKV_ROPE) staged = Rs + row * 16;
        if constexpr (AUX) staged = Xs + (size_t)(lane + 64 * (TPH * hf + i)) * 16;
        if constexpr (EPI == EPI_MUL_AUX) staged = Xs + (size_t)(row * (RB_BN / 16) + (cg >> 1)) * 16 + (cg & 1) * 8;
        if constexpr (EDGE) staged = Rs + cg * 8;
        epilogue8<EPI, 1, RB_BN / 8, 16>(q, m0 + row, jt * RB_BN + cg * 8, v, (m0 + row) < p.M, jt, ntiles, staged);
      };
      // light epilogues run both tasks interleaved; the heavy ones (activation maths, extra operands) one after the
      // other, or their temporaries spill (the kernel lives at the 168-VGPR limit of 3 waves per SIMD)
      if constexpr (TPH == 1) {
        task(0);
      } else if constexpr (EPI == EPI_BF16 || EPI == EPI_F32) {
        task(0);
        task(1);
      } else {
#pragma nounroll
        for (int i = 0; i < TPH; ++i) task(i);
      }
      __builtin_amdgcn_wave_barrier();      // the next writes to Es stay behind these reads
    }
    RB_T(3);
    __syncthreads();
    RB_T(4);
#pragma unroll
    for (int a = 0; a < NACC; ++a) bz[a] = bn[a];
  };
  // cur / nxt reach the tile body as __restrict__ parameters (rb_call_restrict): the compiler waits for every pending
  // global_load_lds before an LDS read it cannot prove disjoint from the DMA's target
  for (int j = 0, jt = j0; j < ntiles; ++j) {
    const int jn = jt + 1 == ntiles ? 0 : jt + 1;   // (behind the last tile: an unused re-read of the first one)
    rb_call_restrict(tile, jt, jn, Bs + (j & 1) * RB_TILE_HALFS, Bs + ((j + 1) & 1) * RB_TILE_HALFS);
    jt = jn;
  }
  RB_TDUMP();
}

static int rb_waves(int M) {
  const int slabs = (M + 31) / 32;
  const int rounds = (slabs + 256 * RB_MAX_W - 1) / (256 * RB_MAX_W);
  int W = (slabs + 256 * rounds - 1) / (256 * rounds);
  if (W > RB_MAX_W) W = RB_MAX_W;
  return W;
}

bool gemm_rb256_ln_fusable(const GemmArgs& a, int epi) {
  if (epi != EPI_QKV_ROPE && epi != EPI_GELU_GRAD) return false;
  GemmArgs b = a;
  b.ln_x = nullptr;
  return gemm_rb256_supported(b, 0, epi) && a.K == RB_K;
}

int gemm_ce_tile_width(const GemmArgs& a) {
  // (the same order of questions as launch_gemm_nt's dispatch)
#ifdef COATI_EXPERIMENTAL
  if (gemm_t32_supported(a, 0, EPI_CE_PARTIAL)) return 64;
#endif
  return (gemm_rb16_supported(a, 0, EPI_CE_PARTIAL) || gemm_rb256_supported(a, 0, EPI_CE_PARTIAL)) ? 64 : 128;
}

// true when (a, epi) can run on the row-block kernel
bool gemm_rb256_supported(const GemmArgs& a, int a_f32, int epi) {
  if (a_f32 || a.K != RB_K) return false;
  if (epi == EPI_CE_PARTIAL && a.partial_tile != 64) return false;   // the caller's partial buffer is laid out for 128-column tiles
  if (epi == EPI_QKV_ROPE && a.rope_hs == 32) return false;   // the staged rotary rows are laid out for head size 16
  if (a.N % 16 != 0 && epi != EPI_CE_BWD && epi != EPI_CE_PARTIAL) return false;   // (those two write no N-wide rows)
  // small problems run on the tiled kernel.  (Packed rows: a full-size batch carries ~50 000 rows instead of 81 920, i.e.
  // 6-7 slabs per workgroup -- still one round of one workgroup per CU, with LayerNorm fused into the operand load.)
  if (rb_waves(a.M) < 5) return false;
  return true;
}

template <int EPI, int BN>
constexpr size_t rb_per_wave_bytes() {
  return (size_t)16 * (BN + 4) * 4 + ((EPI == EPI_QKV_ROPE || EPI == EPI_EDGE_DPRE) ? RB_ROPE_FLOATS * 4
                                      : (EPI == EPI_DGELU || EPI == EPI_DSILU) ? RB_AUX_BYTES : (EPI == EPI_MUL_AUX) ? RB_AUX_BYTES / 2 : 0);
}

template <int EPI, int BN, int MAXW, bool LN = false>
static int launch_rb_shape(const GemmArgs& a, int W, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_rb256_kernel<EPI, BN, MAXW, LN>;
  const size_t tile_bytes = (size_t)2 * BN * RB_K * 2;              // double-buffered weight tile
  constexpr size_t per_wave = rb_per_wave_bytes<EPI, BN>();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(tile_bytes + MAXW * per_wave));
    if (e != hipSuccess) {
      coati_set_error("gemm_rb256: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int blocks = cdiv(cdiv(a.M, 32), W);
  // rotated tile order: measured per epilogue (round 2), kept where it paid
  const int rot = (EPI == EPI_MUL_AUX);
  if constexpr (MAXW == RB_HALF_W) {   // 8 + 4: the same 320 rows per workgroup as 10 full waves
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * RB_HALF_W), tile_bytes + RB_HALF_W * per_wave, s, a, RB_HALF_W, 8, rot);
  } else {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), tile_bytes + W * per_wave, s, a, W, W, rot);
  }
  COATI_LAUNCH_CHECK("gemm_rb256");
  return COATI_OK;
}

template <int EPI>
static int launch_rb_t(const GemmArgs& a, hipStream_t s) {
  const int W = rb_waves(a.M);
  // 8 full + 4 half waves instead of 10 full ones: measured per epilogue (bench --all-sites, M = 81,920): FC1 + GELU/GELU' 3.65 ->
  // 3.51 ms/step, lm_head 0.78 -> 0.75 / 0.73 -> 0.72, but FC2 input gradient 2.41 -> 2.63 and QKV 2.37 -> 2.43 -- it pays only
  // where the epilogue outweighs the extra MFMA block of a half wave;
  // epilogues whose per-wave LDS does not fit 12 times keep 10 waves.
  const bool half_on = (EPI == EPI_GELU_GRAD || EPI == EPI_CE_PARTIAL || EPI == EPI_CE_BWD);
  constexpr bool half_fits = 2 * 64 * RB_K * 2 + RB_HALF_W * rb_per_wave_bytes<EPI, 64>() <= 160 * 1024;
  const bool half = half_on && half_fits && W == RB_MAX_W && a.m_dev == nullptr;
  if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_GELU_GRAD) {
    if (a.ln_x != nullptr) {
      if (W <= RB_FEW_W) return launch_rb_shape<EPI, 64, RB_FEW_W, true>(a, W, s);
      if constexpr (half_fits) { if (half) return launch_rb_shape<EPI, 64, RB_HALF_W, true>(a, W, s); }
      return launch_rb_shape<EPI, 64, RB_MAX_W, true>(a, W, s);
    }
  }
  if (a.ln_x != nullptr) {
    coati_set_error("gemm_rb256: epilogue %d has no fused-LayerNorm variant", (int)EPI);
    return COATI_EARG;
  }
  if (W <= RB_FEW_W) return launch_rb_shape<EPI, 64, RB_FEW_W>(a, W, s);
  if constexpr (half_fits) { if (half) return launch_rb_shape<EPI, 64, RB_HALF_W>(a, W, s); }
  return launch_rb_shape<EPI, 64, RB_MAX_W>(a, W, s);
}

int launch_gemm_rb256(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return launch_rb_t<EPI_BF16>(a, s);
    case EPI_F32: return launch_rb_t<EPI_F32>(a, s);
    case EPI_RES_F32: return launch_rb_t<EPI_RES_F32>(a, s);
    case EPI_GELU: return launch_rb_t<EPI_GELU>(a, s);
    case EPI_DGELU: return launch_rb_t<EPI_DGELU>(a, s);
    case EPI_GELU_GRAD: return launch_rb_t<EPI_GELU_GRAD>(a, s);
    case EPI_MUL_AUX: return launch_rb_t<EPI_MUL_AUX>(a, s);
    case EPI_SILU: return launch_rb_t<EPI_SILU>(a, s);
    case EPI_DSILU: return launch_rb_t<EPI_DSILU>(a, s);
    case EPI_ACC_F32: return launch_rb_t<EPI_ACC_F32>(a, s);
    case EPI_CE_BWD: return launch_rb_t<EPI_CE_BWD>(a, s);
    case EPI_CE_PARTIAL: return launch_rb_t<EPI_CE_PARTIAL>(a, s);
    case EPI_EDGE_DPRE: return launch_rb_t<EPI_EDGE_DPRE>(a, s);
    case EPI_QKV_ROPE: return launch_rb_t<EPI_QKV_ROPE>(a, s);
    default:
      coati_set_error("gemm_rb256: unsupported epilogue %d", epi);
      return COATI_EARG;
  }
}
