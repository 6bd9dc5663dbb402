// Transformer MLP forward at C = 256 as ONE kernel with paired waves (reference basic_transformer.py:157-174,
// RotaryBlock.mlpf: x + W2 NewGELU(W1 ln_2(x) + b1) + b2):
//
//   a   = LayerNorm(x)            bf16, saved (FC1's weight gradient reads it) -- mean / rstd saved too
//   g   = NewGELU(a W1^T + b1)    bf16, saved (FC2's weight gradient reads it)
//   d   = NewGELU'(a W1^T + b1)   8-bit fixed point, saved (the backward multiplies by it)
//   out = x + g W2^T + b2         f32 residual stream
//
// Why this shape.  The two separate launches are bound by different things (tools/probes/rb_trace.py): FC1 + GELU by the
// instruction issue of its epilogue (VALU) and MFMAs, FC2 + residual by HBM (it re-reads g, 2 KB per row).  Fusing them
// removes FC2's read of g, but a wave that keeps BOTH its LayerNorm rows (64 VGPRs) and a 32 x 256 f32 output block
// (128 VGPRs) does not fit 168 registers, and 16-row waves (gemm_mlp.hip) double the LDS fragment traffic.  Here the two
// products belong to two waves of a PAIR that share 32 rows:
//   P (producer):  the row-block FC1 of gemm_rb.hip on 32-unit tiles -- LayerNorm rows resident as 16 A fragments, the W1
//                  tile [32][256] streamed through LDS, GELU + derivative epilogue -- which ALSO drops its g tile
//                  [32 rows][32 units] (bf16) into LDS;
//   Q (consumer):  one tile later multiplies that g tile with the W2 slice [256][32] into its 32 x 256 accumulator
//                  (8 blocks = 128 VGPRs), which was initialised with x + b2 (the residual costs no extra pass).
// One barrier per tile; every buffer is double-buffered (W1 tile, W2 slice, g tiles).  10 waves = 5 pairs = 160 rows per
// workgroup (M = 81 920: 512 workgroups = 2 per CU); roles are placed so that the SIMDs (wave w -> SIMD w % 4) carry
// {P,Q,Q} {P,Q,Q} {P,P} {P,Q}: P costs ~3.4x a Q per tile.
#include <cstdlib>
#include "gemm_epi.h"

#ifndef M2_QPRIO
#define M2_QPRIO 3
#endif
#define M2_C 256
#define M2_BN 32                               // hidden units per tile
#define M2_ROWS 160                            // rows per workgroup
#define M2_W1T_BYTES (M2_BN * M2_C * 2)        // 16 KiB: W1 tile [32 units][256 k], 512-B rows, chunk c at c ^ (row & 31)
#define M2_W2S_BYTES (M2_C * M2_BN * 2)        // 16 KiB: W2 slice [256 c][32 units], 64-B rows, chunk c at c ^ ((row >> 2) & 3)
#define M2_GT_BYTES (32 * M2_BN * 2)           // 2 KiB: g tile [32 rows][32 units], laid out like a W2 slice
#define M2_EPITCH (M2_BN + 4)                  // floats per row of a P wave's transpose region
#define M2_ES_BYTES (16 * M2_EPITCH * 4)
#define M2_OFF_W1 0
#define M2_OFF_W2 (2 * M2_W1T_BYTES)
#define M2_OFF_G (M2_OFF_W2 + 2 * M2_W2S_BYTES)
#define M2_OFF_ES (M2_OFF_G + 2 * 5 * M2_GT_BYTES)
#define M2_LDS_BYTES (M2_OFF_ES + 5 * M2_ES_BYTES)   // 97,024 B

// Probe build (-DCOATI_RB_TRACE, tools/probes/rb_trace.py): shader-clock totals per phase, waves of the first 16 workgroups:
// P: [prologue, MFMA, vmcnt wait, epilogue, barrier]; Q: [prologue (x loads), MFMA, vmcnt wait, write-out, barrier]
#ifdef COATI_RB_TRACE
__device__ unsigned long long m2_trace_buf[16 * 10 * 8];
extern "C" int coati_m2_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(m2_trace_buf), sizeof(m2_trace_buf)) == hipSuccess ? 0 : -3;
}
#define M2_T0() unsigned long long m2_t_last = __builtin_amdgcn_s_memtime(), m2_t_acc[5] = {0, 0, 0, 0, 0}
#define M2_T(i) do { const unsigned long long m2_t_now = __builtin_amdgcn_s_memtime(); m2_t_acc[i] += m2_t_now - m2_t_last; m2_t_last = m2_t_now; } while (0)
#define M2_TDUMP() do { if (blockIdx.x < 16 && lane == 0) { for (int i = 0; i < 5; ++i) m2_trace_buf[(blockIdx.x * 10 + wave) * 8 + i] = m2_t_acc[i]; } } while (0)
#else
#define M2_T0() do { } while (0)
#define M2_T(i) do { } while (0)
#define M2_TDUMP() do { } while (0)
#endif

template <int V> struct M2Int { static constexpr int value = V; };
// wave -> (role, pair): P = {0, 1, 2, 3, 6}, Q = {4, 5, 7, 8, 9}
__device__ __forceinline__ bool m2_is_q(int wave) { return wave == 4 || wave == 5 || wave >= 7; }
__device__ __forceinline__ int m2_pair(int wave) { return wave < 4 ? wave : wave == 6 ? 4 : wave <= 5 ? wave - 4 : wave - 5; }

__global__ __launch_bounds__(640, 1) void mlp_pair_fwd_kernel(MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_q = m2_is_q(wave);
  const int pr = m2_pair(wave);                       // pair index 0..4 (also this wave's index inside its role)
  const int m0 = (int)blockIdx.x * M2_ROWS + 32 * pr;  // first row of the pair
  const int ntiles = p.Hd / M2_BN;
  const int fr = lane & 31, hk = lane >> 5;
  M2_T0();

  if (!is_q) {
    // =============================== P: LayerNorm -> FC1 -> NewGELU (+ derivative) ===============================
    float* const Es = reinterpret_cast<float*>(smem + M2_OFF_ES + pr * M2_ES_BYTES);
    const int fk = hk * 8;
    // LayerNorm fused into the slab load (same arithmetic as gemm_rb256_kernel<.., LN = true> / ln_fwd_kernel: two-pass
    // statistics in registers; the lane pair (fr, hk = 0 / 1) holds one f32 row as 2 x 16 chunks of 8)
    bf16x8 af[16];
    {
      const int row = m0 + fr, rc = row < p.M ? row : p.M - 1;
      const float* xp = p.x + (long long)rc * p.ldx + fk;
      float xf[16][8];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const float4 x0 = *reinterpret_cast<const float4*>(xp + ks * 16), x1 = *reinterpret_cast<const float4*>(xp + ks * 16 + 4);
        xf[ks][0] = x0.x; xf[ks][1] = x0.y; xf[ks][2] = x0.z; xf[ks][3] = x0.w;
        xf[ks][4] = x1.x; xf[ks][5] = x1.y; xf[ks][6] = x1.z; xf[ks][7] = x1.w;
      }
      float sm = 0.f;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) sm += xf[ks][i];
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm / (float)M2_C;
      float q2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = xf[ks][i] - mean;
          q2 += d * d;
        }
      q2 += __shfl_xor(q2, 32, 64);
      const float rstd = 1.0f / sqrtf(q2 / (float)M2_C + 1e-5f);
      if (lane < 32 && row < p.M) {
        p.mean[row] = mean;
        p.rstd[row] = rstd;
      }
      bf16_t* op = p.a + (long long)rc * p.lda + fk;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + fk + ks * 16), g1 = *reinterpret_cast<const float4*>(p.gamma + fk + ks * 16 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p.beta + fk + ks * 16), b1 = *reinterpret_cast<const float4*>(p.beta + fk + ks * 16 + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (xf[ks][i] - mean) * rstd * g[i] + bt[i];
        const uint4 u = pack8(o);
        af[ks] = __builtin_bit_cast(bf16x8, u);
        if (row < p.M) *reinterpret_cast<uint4*>(op + ks * 16) = u;
      }
    }
    // W1 tile j -> LDS: 16 pieces of 1 KiB (two 512-B tile rows each); P wave pr takes pieces pr, pr + 5, pr + 10 (, 15)
    auto load_w1 = [&](int j, unsigned char* S) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = pr + 5 * i;
        if (k < M2_BN / 2) {
          const int r = 2 * k + (lane >> 5), q = lane & 31;
          int gr = j * M2_BN + r;
          gr = gr < p.Hd ? gr : p.Hd - 1;
          const bf16_t* src = p.W1 + (long long)gr * p.ldw1 + ((q ^ (r & 31)) * 8);
          __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(S + k * 1024), 16, 0, 0);
        }
      }
    };
    load_w1(0, smem + M2_OFF_W1);
    float bz = p.b1[fr], bn = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __builtin_amdgcn_s_barrier();         // ---- barrier 0: W1 tile 0 is in LDS
    M2_T(0);

    // output pointers of this wave's rows (32-bit offsets inside the slab)
    const long long hrow0 = (long long)m0 * p.ldh;
    bf16_t* const gout = p.h + hrow0;
    unsigned char* const dout = reinterpret_cast<unsigned char*>(p.d) + hrow0;
    for (int j = 0; j < ntiles; ++j) {
      const unsigned char* cur = smem + M2_OFF_W1 + (j & 1) * M2_W1T_BYTES;
      load_w1(j + 1 < ntiles ? j + 1 : j, smem + M2_OFF_W1 + ((j + 1) & 1) * M2_W1T_BYTES);
      {
        const int c = (j + 1) * M2_BN + fr;
        bn = p.b1[c < p.Hd ? c : p.Hd - 1];
      }
      __builtin_amdgcn_s_setprio(M2_QPRIO);   // MFMA phase first, epilogue at the default priority (as in gemm_rb.hip)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bz;
      {
        const bf16_t* wp = reinterpret_cast<const bf16_t*>(cur) + fr * M2_C;
        bf16x8 wf[3];
        wf[0] = *reinterpret_cast<const bf16x8*>(wp + (((0 + hk) ^ fr) * 8));
        wf[1] = *reinterpret_cast<const bf16x8*>(wp + (((2 + hk) ^ fr) * 8));
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (ks + 2 < 16) wf[(ks + 2) % 3] = *reinterpret_cast<const bf16x8*>(wp + (((2 * (ks + 2) + hk) ^ fr) * 8));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], wf[ks % 3], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      M2_T(1);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the next W1 tile has landed (it had the MFMA phase); this tile's stores go out behind it
      M2_T(2);
      unsigned char* const gt = smem + M2_OFF_G + ((j & 1) * 5 + pr) * M2_GT_BYTES;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int row = (rr & 3) + 8 * (rr >> 2) + 4 * hk;   // 0..15 within this half
          Es[row * M2_EPITCH + fr] = acc[hf * 8 + rr];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        {
          const int rl = lane >> 2, cg = lane & 3, row = hf * 16 + rl;
          const float4 c0 = *reinterpret_cast<const float4*>(Es + rl * M2_EPITCH + cg * 8);
          const float4 c1 = *reinterpret_cast<const float4*>(Es + rl * M2_EPITCH + cg * 8 + 4);
          const float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
          float o[8], d[8];
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            coati_v2f hh, dd;
            gelu_and_grad_f2(coati_v2f{v[e], v[e + 1]}, hh, dd);
            o[e] = hh.x; o[e + 1] = hh.y;
            d[e] = dd.x; d[e + 1] = dd.y;
          }
          const uint4 hv = pack8(o);
          // the pair's g tile: row `row`, 16-B chunk cg at position cg ^ ((row >> 2) & 3)
          *reinterpret_cast<uint4*>(gt + row * 64 + ((cg ^ ((row >> 2) & 3)) << 4)) = hv;
          if (m0 + row < p.M) {
            const unsigned off = (unsigned)row * (unsigned)p.ldh + (unsigned)(j * M2_BN + cg * 8);
            *reinterpret_cast<uint4*>(gout + off) = hv;
            *reinterpret_cast<uint2*>(dout + off) = packq8(d);
          }
        }
        __builtin_amdgcn_wave_barrier();      // the next writes to Es stay behind these reads
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): the g tile is written
      M2_T(3);
      __builtin_amdgcn_s_barrier();           // ---- barrier j + 1
      M2_T(4);
      bz = bn;
    }
    __builtin_amdgcn_s_barrier();             // ---- barrier ntiles + 1 (the consumers' last tile)
    M2_T(4);
    M2_TDUMP();
  } else {
    // =============================== Q: FC2 + residual ===============================
    if (p.M >= 0) __builtin_amdgcn_s_setprio(M2_QPRIO);   // consumers are short MFMA bursts: let them go first (A/B: -DM2_QPRIO=0)
    // accumulator = x + b2: lane = output column inside a 32-column block, registers = rows (frag_row)
    f32x16 acc2[8];
    // The residual x (128 values per lane) is NOT loaded up front -- 128 dword loads per consumer wave next to the producers'
    // LayerNorm loads made the prologue a quarter of the kernel -- but four values per interval, added behind that interval's
    // MFMAs: interval J brings rows 8 (J & 3) + 4 hk + 0..3 of column block J >> 2 (32 intervals cover the 8 x 16 registers).
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const float b = p.b2[32 * n + fr];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[n][r] = b;
    }
    const float* const xq = p.x + fr;
    float xr[4];
    auto x_load = [&](int J) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = m0 + 8 * (J & 3) + 4 * hk + k, rc = row < p.M ? row : p.M - 1;
        xr[k] = xq[(long long)rc * p.ldx + 32 * (J >> 2)];
      }
    };
    // W2 slice j -> LDS: 16 pieces of 1 KiB (sixteen 64-B rows each); Q wave pr takes pieces pr, pr + 5, pr + 10 (, 15)
    auto load_w2 = [&](int j, unsigned char* S) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = pr + 5 * i;
        if (k < 16) {
          const int r = 16 * k + (lane >> 2), q = lane & 3;
          const bf16_t* src = p.W2 + (long long)r * p.ldw2 + j * M2_BN + ((q ^ ((r >> 2) & 3)) * 8);
          __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(S + k * 1024), 16, 0, 0);
        }
      }
    };
    __builtin_amdgcn_s_barrier();         // ---- barrier 0 (the producers' first W1 tile)
    M2_T(0);
    // fragment addresses inside a g tile / a W2 slice: row (lane & 31), chunk 2 ks + hk at position chunk ^ ((row >> 2) & 3)
    const int sw = (fr >> 2) & 3;
    const unsigned fo0 = (unsigned)(fr * 64 + (((0 + hk) ^ sw) << 4)), fo1 = (unsigned)(fr * 64 + (((2 + hk) ^ sw) << 4));
    // one interval (= one barrier): fetch W2 slice j, multiply g tile j - 1 by slice j - 1; XI >= 0: also bring residual group XI
    auto interval = [&](int j, auto xi_c) __attribute__((always_inline)) {
      constexpr int XI = decltype(xi_c)::value;
      // slice j is multiplied in interval j + 1: it is fetched now, into the buffer that slice j - 2 left after interval j - 1
      if (j < ntiles) load_w2(j, smem + M2_OFF_W2 + (j & 1) * M2_W2S_BYTES);
      if constexpr (XI >= 0) x_load(XI);
      if (j >= 1) {
        const unsigned char* gt = smem + M2_OFF_G + (((j - 1) & 1) * 5 + pr) * M2_GT_BYTES;
        const unsigned char* ws = smem + M2_OFF_W2 + ((j - 1) & 1) * M2_W2S_BYTES;
        const bf16x8 ga0 = *reinterpret_cast<const bf16x8*>(gt + fo0), ga1 = *reinterpret_cast<const bf16x8*>(gt + fo1);
        bf16x8 wb[2][2];   // (a third fragment set does not fit beside the 128 accumulator registers)
        wb[0][0] = *reinterpret_cast<const bf16x8*>(ws + fo0);
        wb[0][1] = *reinterpret_cast<const bf16x8*>(ws + fo1);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          if (n + 1 < 8) {
            wb[(n + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(ws + (n + 1) * 32 * 64 + fo0);
            wb[(n + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(ws + (n + 1) * 32 * 64 + fo1);
          }
          acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga0, wb[n & 1][0], acc2[n], 0, 0, 0);
          acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga1, wb[n & 1][1], acc2[n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      M2_T(1);
      __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) + lgkmcnt(0): the next W2 slice has landed, this wave's LDS reads are complete
      M2_T(2);
      if constexpr (XI >= 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc2[XI >> 2][4 * (XI & 3) + k] += xr[k];
      }
      __builtin_amdgcn_s_barrier();         // ---- barrier j + 1
      M2_T(4);
    };
    // the first 32 intervals are unrolled (the residual group of an interval names accumulator registers), the rest loop
    int j = 0;
#define M2_IV(J) if (j <= ntiles) { interval(j, M2Int<J>()); ++j; }
    M2_IV(0) M2_IV(1) M2_IV(2) M2_IV(3) M2_IV(4) M2_IV(5) M2_IV(6) M2_IV(7) M2_IV(8) M2_IV(9) M2_IV(10) M2_IV(11) M2_IV(12) M2_IV(13) M2_IV(14) M2_IV(15)
    M2_IV(16) M2_IV(17) M2_IV(18) M2_IV(19) M2_IV(20) M2_IV(21) M2_IV(22) M2_IV(23) M2_IV(24) M2_IV(25) M2_IV(26) M2_IV(27) M2_IV(28) M2_IV(29) M2_IV(30) M2_IV(31)
#undef M2_IV
    for (; j <= ntiles; ++j) interval(j, M2Int<-1>());
    // fewer than 31 tiles: the rest of the residual
#define M2_XR(J) if (J > ntiles) { x_load(J); _Pragma("unroll") for (int k = 0; k < 4; ++k) acc2[(J) >> 2][4 * ((J) & 3) + k] += xr[k]; }
    M2_XR(1) M2_XR(2) M2_XR(3) M2_XR(4) M2_XR(5) M2_XR(6) M2_XR(7) M2_XR(8) M2_XR(9) M2_XR(10) M2_XR(11) M2_XR(12) M2_XR(13) M2_XR(14) M2_XR(15) M2_XR(16)
    M2_XR(17) M2_XR(18) M2_XR(19) M2_XR(20) M2_XR(21) M2_XR(22) M2_XR(23) M2_XR(24) M2_XR(25) M2_XR(26) M2_XR(27) M2_XR(28) M2_XR(29) M2_XR(30) M2_XR(31)
#undef M2_XR
    // write-out: for one register the 32 lanes of a half-wave hold 32 consecutive columns of one row (a 128-B line)
    float* const out = reinterpret_cast<float*>(p.out) + fr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + frag_row(r, lane);
      if (row < p.M) {
#pragma unroll
        for (int n = 0; n < 8; ++n) out[(long long)row * p.ldo + 32 * n] = acc2[n][r];
      }
    }
    M2_T(3);
    M2_TDUMP();
  }
}

bool mlp_pair_supported(const MlpArgs& a) {
  return a.C == M2_C && a.Hd % M2_BN == 0 && a.Hd >= M2_BN && a.M > 0 && a.lda % 8 == 0 && a.ldh % 8 == 0 && a.ldx % 4 == 0 && a.ldo % 4 == 0 &&
         a.ldw1 % 8 == 0 && a.ldw2 % 8 == 0 && 40LL * a.ldh < (1LL << 31);
}

int launch_mlp_pair_fwd(const MlpArgs& a, hipStream_t s) {
  COATI_CHECK_ARG(a.x && a.gamma && a.beta && a.W1 && a.b1 && a.W2 && a.b2 && a.a && a.h && a.d && a.out && a.mean && a.rstd, "mlp_pair_fwd: null operand");
  COATI_CHECK_SHAPE(mlp_pair_supported(a), "mlp_pair_fwd: unsupported shape C=%d Hd=%d", a.C, a.Hd);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_pair_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, M2_LDS_BYTES);
    if (e != hipSuccess) {
      coati_set_error("mlp_pair_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(mlp_pair_fwd_kernel, dim3(cdiv(a.M, M2_ROWS)), dim3(640), M2_LDS_BYTES, s, a);
  COATI_LAUNCH_CHECK("mlp_pair_fwd");
  return COATI_OK;
}
