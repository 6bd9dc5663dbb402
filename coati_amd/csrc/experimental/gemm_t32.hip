// Row-block GEMM for K = 256 on 32-row slabs, TRANSPOSED product (round 5): C[M, N] = A[M, 256] W[N, 256]^T for the transformer's
// K = 256 products at packed-batch sizes -- QKV (+ ln_1, RoPE), FC1 (+ ln_2, NewGELU + derivative), the FC2 input gradient (x the saved
// derivative), the lm_head's partial cross-entropy and its gradient.
//
// Why another form.  The 16-row-slab kernel (gemm_rb16.hip) runs v_mfma_f32_16x16x32_bf16 with the weight fragment from LDS and the
// slab from registers: ONE 1-KiB LDS fragment read per 16-cycle MFMA, i.e. twice what the LDS delivers per MFMA cycle -- it is bound
// by LDS reads at half the matrix core's rate (its header: 13.6 us of MFMA issue inside a 41-us launch).  The phase trace of the fused
// attention kernel (attn_block.hip, profiles/r05_attn_block_trace.txt) measured the same product in another form at 70 % of the matrix
// core inside its phase: D^T = W a^T on v_mfma_f32_32x32x16_bf16 with the WEIGHT rows as the MFMA's row operand (from LDS) and the
// resident 32-row slab as its column operand (registers): one 1-KiB fragment read per 32-cycle MFMA, and the lane ends with ONE
// token row and, per 32-feature block, 16 features of it -- bias, RoPE, activations run in registers, a v_permlane32_swap per pair
// of registers regroups them into 8 consecutive columns per lane for 16-B stores.  This file is that phase as a kernel of its own.
//
//   * workgroup = W waves (4..8), wave w owns rows 32 (W blockIdx + w) .. + 31 and keeps them as 16 MFMA fragments (64 VGPRs);
//     W = ceil(rows / 32 / 256): one round of one workgroup per CU, 2 waves per SIMD, 256 VGPRs per wave
//   * LayerNorm in the slab load (QKV, FC1): the lane pair (row, half) holds the row's 256 values in operand order; statistics in
//     registers + one v_permlane32_swap; the normalised row -> fragments + the bf16 copy the weight gradient reads + mean / rstd
//   * weight tiles [64 features][256 k] (32 KiB) L2 -> LDS by global_load_lds into a ring of 3, two tiles ahead; the 16-B chunk c of
//     row f sits at chunk slot c ^ (f & 31): the 16 lanes of a fragment read hit 16 different bank groups
//   * per tile and wave: 32 MFMAs (2 feature blocks x 16 k steps, two independent accumulators), then the write-out
//   * one barrier per tile; before a tile's stores are issued the wave waits for ITS pieces of the next tile (vmcnt counts the
//     tile after that as still in flight), so the barrier never waits for a store acknowledgement
#include <cstdlib>
#include "../gemm_epi.h"

#define T32_K 256
#define T32_BN 64
#define T32_STAGE 32768
#define T32_NS 3
#define T32_MAXW 8
#define T32_BIAS_MAX 4096

typedef unsigned t32_v2u __attribute__((ext_vector_type(2)));

#ifdef COATI_T32_TRACE
// Probe build (-DCOATI_T32_TRACE, tools/probes/t32_trace.py): shader-clock totals per phase of the waves of the first 16 workgroups:
// [wg][wave][slab load (+ LayerNorm), tile barrier, MFMA loop, wait for the next tile's pieces, write-out, -, -, whole kernel]
__device__ unsigned long long t32_trace_buf[16 * 8 * 8];
extern "C" int coati_t32_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(t32_trace_buf), sizeof(t32_trace_buf)) == hipSuccess ? 0 : -3;
}
#define T32_T0() unsigned long long t32_t_last = __builtin_amdgcn_s_memtime(), t32_t_first = t32_t_last, t32_t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define T32_T(i) do { const unsigned long long t32_t_now = __builtin_amdgcn_s_memtime(); t32_t_acc[i] += t32_t_now - t32_t_last; t32_t_last = t32_t_now; } while (0)
#define T32_TDUMP() do { t32_t_acc[7] = t32_t_last - t32_t_first; if (blockIdx.x < 16 && lane == 0) { for (int i = 0; i < 8; ++i) t32_trace_buf[(blockIdx.x * 8 + wave) * 8 + i] = t32_t_acc[i]; } } while (0)
#else
#define T32_T0() do { } while (0)
#define T32_T(i) do { } while (0)
#define T32_TDUMP() do { } while (0)
#endif

__device__ __forceinline__ void t32_wait_vm(int n) {   // vmcnt(n), n wave-uniform, for the counts that occur (4..16); anything else waits for everything
  switch (n) {
    case 4: __builtin_amdgcn_s_waitcnt(0x0f74); break;
    case 5: __builtin_amdgcn_s_waitcnt(0x0f75); break;
    case 6: __builtin_amdgcn_s_waitcnt(0x0f76); break;
    case 7: __builtin_amdgcn_s_waitcnt(0x0f77); break;
    case 8: __builtin_amdgcn_s_waitcnt(0x0f78); break;
    case 9: __builtin_amdgcn_s_waitcnt(0x0f79); break;
    case 10: __builtin_amdgcn_s_waitcnt(0x0f7a); break;
    case 11: __builtin_amdgcn_s_waitcnt(0x0f7b); break;
    case 12: __builtin_amdgcn_s_waitcnt(0x0f7c); break;
    case 13: __builtin_amdgcn_s_waitcnt(0x0f7d); break;
    case 14: __builtin_amdgcn_s_waitcnt(0x0f7e); break;
    case 15: __builtin_amdgcn_s_waitcnt(0x0f7f); break;
    case 16: __builtin_amdgcn_s_waitcnt(0x4f70); break;
    default: __builtin_amdgcn_s_waitcnt(0x0f70); break;
  }
}
__device__ __forceinline__ void t32_barrier() {
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
}

// cur / nxt reach the tile body as __restrict__ parameters (gemm_rb16.hip: the compiler otherwise waits for every pending
// global_load_lds before an LDS read it cannot prove disjoint from the DMA's target)
template <typename F>
__device__ __forceinline__ void t32_call_restrict(F&& f, int j, const unsigned char* __restrict__ cur, unsigned char* __restrict__ nxt) {
  f(j, cur, nxt);
}

template <int EPI, bool LN>
__global__ __launch_bounds__(64 * T32_MAXW, 1) void gemm_t32_kernel(GemmArgs p, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const BiasS = reinterpret_cast<float*>(smem + T32_NS * T32_STAGE);
  float* const GB = BiasS + T32_BIAS_MAX;   // LN variants: ln gamma | beta (512 floats), read as LDS broadcasts
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, half = lane >> 5;
  const int ntiles = (p.N + T32_BN - 1) / T32_BN;
  T32_T0();
  const int row = (blockIdx.x * W + wave) * 32 + tl;
  const bool rowok = row < p.M;
  const int rc = rowok ? row : p.M - 1;
  const bool has_bias = p.bias != nullptr;
  if (has_bias)
    for (int c = tid; c < ntiles * T32_BN; c += blockDim.x) BiasS[c] = c < p.N ? p.bias[c] : 0.f;

  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;
  // weight tile: 32 pieces of 1 KiB (two 512-B rows), piece i by wave i % W: lane -> row f = 2 i + (lane >> 5), chunk slot lane & 31,
  // global chunk (lane & 31) ^ (f & 31); rows past N re-read row N - 1 (their columns are masked / never stored)
  const int my_dmas = (32 - wave + W - 1) / W;
  auto load_tile = [&](int n0, unsigned char* S) {
#pragma unroll
    for (int i8 = 0; i8 < 8; ++i8) {
      const int i = wave + W * i8;
      if (i < 32) {
        const int f = 2 * i + (lane >> 5), sl = lane & 31;
        const int g = n0 + f, gc = g < p.N ? g : p.N - 1;
        const bf16_t* src = p.B + (long long)gc * p.ldb + ((sl ^ (f & 31)) << 3);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(S + i * 1024), 16, 0, 0);
      }
    }
  };
  load_tile(0, smem);
  if (ntiles > 1) load_tile(T32_BN, smem + T32_STAGE);
  if constexpr (LN) {
    if (tid < 128) {
      const float* src = tid < 64 ? p.ln_gamma + 4 * tid : p.ln_beta + 4 * (tid - 64);
      *reinterpret_cast<float4*>(GB + 4 * tid) = *reinterpret_cast<const float4*>(src);
    }
  }

  // ---- resident slab: fragment ks = k 16 ks + 8 half .. + 7 of this lane's row (the MFMA's column operand)
  bf16x8 af[16];
  if constexpr (!LN) {
    const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.A) + (long long)rc * p.lda + half * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 16);
  } else {
    // LayerNorm inside the slab load (basic_transformer.py:165-173): two-pass statistics over the row's 256 values, 128 in this
    // lane, 128 in lane ^ 32; gamma / beta as LDS broadcasts
    const float* xp = p.ln_x + (long long)rc * p.ln_ldx + half * 8;
    float xf[16][8];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float4 x0 = *reinterpret_cast<const float4*>(xp + ks * 16), x1 = *reinterpret_cast<const float4*>(xp + ks * 16 + 4);
      xf[ks][0] = x0.x; xf[ks][1] = x0.y; xf[ks][2] = x0.z; xf[ks][3] = x0.w; xf[ks][4] = x1.x; xf[ks][5] = x1.y; xf[ks][6] = x1.z; xf[ks][7] = x1.w;
    }
    float sm = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) sm += xf[ks][i];
    const float mean = half_xchg_sum(sm) * (1.0f / T32_K);
    float vs = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = xf[ks][i] - mean; vs = fmaf(d, d, vs); }
    const float rstd = rsqrtf(half_xchg_sum(vs) * (1.0f / T32_K) + 1e-5f);
    if (half == 0 && rowok) { p.ln_mean[row] = mean; p.ln_rstd[row] = rstd; }
    t32_barrier();   // gamma / beta are in LDS
    bf16_t* op = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.A)) + (long long)rc * p.lda + half * 8;
    const float* gp = GB + half * 8;
    const float* bp = GB + 256 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float4 g0 = *reinterpret_cast<const float4*>(gp + ks * 16), g1 = *reinterpret_cast<const float4*>(gp + ks * 16 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(bp + ks * 16), b1 = *reinterpret_cast<const float4*>(bp + ks * 16 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (xf[ks][i] - mean) * rstd * g[i] + bt[i];
      const uint4 u = pack8(o);
      af[ks] = __builtin_bit_cast(bf16x8, u);
      if (rowok) *reinterpret_cast<uint4*>(op + ks * 16) = u;
    }
  }
  // rotary table row of this lane's token: c_i, s_i for i = 4 half .. + 3 (entries i and i + 8 of the [n_seq, 16] tables are equal)
  float rc_[4], rs_[4];
  if constexpr (EPI == EPI_QKV_ROPE) {
    const int t = p.rope_row_t != nullptr ? p.rope_row_t[rc] : rc % p.rope_T;
    const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + t * 16 + 4 * half);
    const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sin + t * 16 + 4 * half);
    rc_[0] = c4.x; rc_[1] = c4.y; rc_[2] = c4.z; rc_[3] = c4.w; rs_[0] = s4.x; rs_[1] = s4.y; rs_[2] = s4.z; rs_[3] = s4.w;
  }
  GemmArgs q = p;
  q.bias = nullptr;   // folded into the accumulator initialisation
  // write-out addresses of this lane's row (element offsets fit 32 bits: launch_gemm_nt refuses M * ld >= 2^32); lane (row, half)
  // stores the 8 columns 8 half .. + 7 of every 16-wide group
  bf16_t* const Crow = reinterpret_cast<bf16_t*>(p.C) + (long long)((unsigned)rc * (unsigned)p.ldc + 8u * (unsigned)half);
  unsigned char* const Xrow = (EPI == EPI_GELU_GRAD) ? reinterpret_cast<unsigned char*>(p.aux_out) + (long long)((unsigned)rc * (unsigned)p.ld_aux + 8u * (unsigned)half) : nullptr;
  float ce_l = 0.f, ce_inv = 0.f;
  int ce_t = -1;        // target column relative to 4 half (never matches when the row has no target)
  if constexpr (EPI == EPI_CE_BWD) {
    const long long tgt = p.target[rc];
    const float cnt = p.scal[1];
    ce_inv = (tgt >= 0 && cnt > 0.f) ? 1.0f / cnt : 0.f;
    ce_l = p.lse[rc];
    ce_t = tgt >= 0 ? (int)tgt - 4 * half : -(1 << 30);
  }
  const int n_st = p.n_store > p.N ? p.n_store : p.N;
  const bool wave_live = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * W + wave) * 32 < p.M)) != 0;   // (its stores are issued at all)
  int st_prev = 0;   // vector-memory operations of the previous tile's write-out

  // the first tile has landed (the second may still be in flight)
  if (ntiles > 1) t32_wait_vm(my_dmas); else __builtin_amdgcn_s_waitcnt(0x0f70);

  T32_T(0);
  auto tile = [&](int j, const unsigned char* cur, unsigned char* nxt) {
    t32_barrier();     // tile j has landed (every wave waited for its pieces), every wave is done with tile j - 1
    T32_T(1);
    uint2 codes[4];
    if constexpr (EPI == EPI_MUL_AUX) {
      // the saved NewGELU' codes of this lane's 4 x 8 columns of the tile, issued BEFORE the DMA below: the wait for them then
      // leaves the DMA in flight
      const unsigned char* X = reinterpret_cast<const unsigned char*>(p.aux_in) + (long long)rc * p.ld_aux + j * T32_BN + 8 * half;
#pragma unroll
      for (int hd = 0; hd < 4; ++hd) codes[hd] = (j * T32_BN + 16 * hd + 8 * half + 8 <= p.N) ? *reinterpret_cast<const uint2*>(X + 16 * hd) : make_uint2(0, 0);
    }
    if (j + 2 < ntiles) load_tile((j + 2) * T32_BN, nxt);
    // ---- D^T[feature][token] = W_tile a^T: 2 blocks of 32 features x this wave's 32 rows, K = 256 -----------------------------
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (has_bias) {
        const float* bq = BiasS + j * T32_BN + 32 * b + 4 * half;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float4 b4 = *reinterpret_cast<const float4*>(bq + 8 * rg);
          acc[b][4 * rg] = b4.x; acc[b][4 * rg + 1] = b4.y; acc[b][4 * rg + 2] = b4.z; acc[b][4 * rg + 3] = b4.w;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      }
    }
    {
      // Four fragment sets in rotation, the reads of k step ks + 3 issued in front of the MFMAs of step ks.  The reads and their waits
      // are inline assembly: left to the compiler, the pre-RA scheduler sinks every read to its use (one fragment set,
      // `s_waitcnt lgkmcnt(0)` in front of every MFMA pair -- the LDS latency of every k step exposed: 39 vs 32 us against the
      // 16-row-slab kernel on the plain product).  LDS address of chunk (2 ks + half) ^ tl of row tl: P ^ (ks << 5).
      __builtin_amdgcn_sched_barrier(0);   // (the bias reads above stay above: their wait must not cover the reads below)
      typedef __attribute__((address_space(3))) unsigned char lds_u8;
      const unsigned P = (unsigned)(size_t)(lds_u8*)smem + (unsigned)(cur - smem) +
                         (unsigned)(tl * 512 + (((tl >> 1) << 5) | ((half ^ (tl & 1)) << 4)));
      bf16x8 wf[4][2];
#define T32_RD(i, ksv)                                                                                                   \
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16384" : "=&v"(wf[i][0]), "=&v"(wf[i][1]) : "v"(P ^ ((unsigned)(ksv) << 5)))
#define T32_WAIT(n, i) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(wf[i][0]), "+v"(wf[i][1]))
      T32_RD(0, 0);
      T32_RD(1, 1);
      T32_RD(2, 2);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        switch ((ks + 3) & 3) {   // (compile-time after unrolling; the asm operands must be named registers)
          case 0: if (ks + 3 < 16) T32_RD(0, ks + 3); break;
          case 1: if (ks + 3 < 16) T32_RD(1, ks + 3); break;
          case 2: if (ks + 3 < 16) T32_RD(2, ks + 3); break;
          default: if (ks + 3 < 16) T32_RD(3, ks + 3); break;
        }
        // reads of the steps behind ks may stay in flight: 2 per step, at most 3 steps
        if (ks <= 12) { switch (ks & 3) { case 0: T32_WAIT(6, 0); break; case 1: T32_WAIT(6, 1); break; case 2: T32_WAIT(6, 2); break; default: T32_WAIT(6, 3); break; } }
        else if (ks == 13) T32_WAIT(4, 1);
        else if (ks == 14) T32_WAIT(2, 2);
        else T32_WAIT(0, 3);
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 3][b], af[ks], acc[b], 0, 0, 0);
      }
#undef T32_RD
#undef T32_WAIT
    }
    T32_T(2);
    // this wave's pieces of tile j + 1 have landed -- waited for BEFORE this tile's stores are issued.  vmcnt counts in order:
    // behind those pieces sit the previous tile's stores (st_prev of them, when that tile took the lean write-out and the wave holds
    // a valid row) and the pieces of tile j + 2 -- none of which is waited for
    if (j + 2 < ntiles) t32_wait_vm(my_dmas + st_prev); else __builtin_amdgcn_s_waitcnt(0x0f70);
    T32_T(3);

    // ---- write-out.  Register r = 4 rg + jj of block b: feature 32 b + 8 rg + 4 half + jj of the tile = head 2 b + (rg >> 1) of the
    // tile's four 16-wide heads, dim 8 (rg & 1) + 4 half + jj
    if constexpr (EPI == EPI_CE_PARTIAL) {
      // (max, sum exp) over the tile's 64 columns of this row: 32 in this lane, 32 in lane ^ 32
      const int c_base = j * T32_BN + 4 * half;
      const bool full = (j + 1) * T32_BN <= p.N;
      float mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = c_base + 32 * b + 8 * (r >> 2) + (r & 3);
          if (full || c < p.N) mx = fmaxf(mx, acc[b][r]);
        }
      mx = half_xchg_max(mx);
      float sm = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = c_base + 32 * b + 8 * (r >> 2) + (r & 3);
          if (full || c < p.N) sm += __expf(acc[b][r] - mx);   // (subtraction first: see EPI_CE_BWD in gemm_epi.h)
        }
      sm = half_xchg_sum(sm);
      if (rowok && half == 0) p.partial[(long long)row * ntiles + j] = make_float2(mx, sm);
      st_prev = wave_live ? 1 : 0;
    } else if ((j + 1) * T32_BN <= (EPI == EPI_CE_BWD ? n_st : p.N)) {
      // Every 16-B store of the tile lies inside the row: write-out in the accumulator's own layout.  The general 8-column epilogue
      // (gemm_epi.h) spent ~ 1400 clocks per tile and wave here (tools/probes/t32_trace.py: 24 % of the kernel on the plain product,
      // 46 % with NewGELU) on per-chunk address arithmetic and bound checks; the MFMA pipe and the VALU of a SIMD do not overlap.
      // Activations run on (lo, hi) = dims 4 half + jj, 8 + 4 half + jj of head hd; the bf16 PAIRS are exchanged (2 v_permlane32_swap
      // per 16-B store): lanes < 32 end with dims 0..7, lanes >= 32 with dims 8..15.
      const bool fullN = (j + 1) * T32_BN <= p.N;
      bf16_t* const cp = Crow + j * T32_BN;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int hd = 2 * b + hh;
          float lo[4], hi[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { lo[jj] = acc[b][8 * hh + jj]; hi[jj] = acc[b][8 * hh + 4 + jj]; }
          if constexpr (EPI == EPI_QKV_ROPE) {
            // RotaryEmbedding.rotary_embed (basic_transformer.py:83-100): y_i = x_i c_i - x_{i+8} s_i, y_{i+8} = x_{i+8} c_i + x_i s_i;
            // q and k tiles only (a 64-column tile never straddles 2 C: rope_C % 64 == 0)
            if (j * T32_BN < 2 * p.rope_C) {
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const float l = lo[jj], h = hi[jj];
                lo[jj] = fmaf(-h, rs_[jj], l * rc_[jj]);
                hi[jj] = fmaf(l, rs_[jj], h * rc_[jj]);
              }
            }
          } else if constexpr (EPI == EPI_GELU_GRAD) {
            float dl[4], dh[4];
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              coati_v2f h2, d2;
              gelu_and_grad_f2(coati_v2f{lo[e], lo[e + 1]}, h2, d2);
              lo[e] = h2.x; lo[e + 1] = h2.y; dl[e] = d2.x; dl[e + 1] = d2.y;
              gelu_and_grad_f2(coati_v2f{hi[e], hi[e + 1]}, h2, d2);
              hi[e] = h2.x; hi[e + 1] = h2.y; dh[e] = d2.x; dh[e + 1] = d2.y;
            }
            unsigned ql = 0, qh = 0;   // NewGELU' as 8-bit fixed point (common.h, packq8)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              ql = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(dl[i], COATI_DQ_SCALE, COATI_DQ_OFF), i, ql);
              qh = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(dh[i], COATI_DQ_SCALE, COATI_DQ_OFF), i, qh);
            }
            const t32_v2u r = __builtin_amdgcn_permlane32_swap(ql, qh, false, false);
            if (rowok) *reinterpret_cast<uint2*>(Xrow + j * T32_BN + 16 * hd) = make_uint2(r.x, r.y);
          } else if constexpr (EPI == EPI_MUL_AUX) {
            // the codes were loaded in the store layout (dims 8 half .. + 7): one exchange brings (lo, hi)'s four each
            const t32_v2u r = __builtin_amdgcn_permlane32_swap(codes[hd].x, codes[hd].y, false, false);
            const float sc = 1.0f / COATI_DQ_SCALE, of = -COATI_DQ_OFF / COATI_DQ_SCALE;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              lo[jj] *= fmaf((float)((r.x >> (8 * jj)) & 0xffu), sc, of);
              hi[jj] *= fmaf((float)((r.y >> (8 * jj)) & 0xffu), sc, of);
            }
          } else if constexpr (EPI == EPI_CE_BWD) {
            // (softmax - onehot) / count; exp(v - lse) keeps the subtraction first (see EPI_CE_BWD in gemm_epi.h)
            const int rel = ce_t - (j * T32_BN + 16 * hd);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              lo[jj] = (__expf(lo[jj] - ce_l) - (rel == jj ? 1.0f : 0.0f)) * ce_inv;
              hi[jj] = (__expf(hi[jj] - ce_l) - (rel == 8 + jj ? 1.0f : 0.0f)) * ce_inv;
            }
            if (!fullN) {
              const int c = j * T32_BN + 16 * hd + 4 * half;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                if (c + jj >= p.N) lo[jj] = 0.f;
                if (c + 8 + jj >= p.N) hi[jj] = 0.f;
              }
            }
          }
          const t32_v2u r0 = __builtin_amdgcn_permlane32_swap(pack2bf(lo[0], lo[1]), pack2bf(hi[0], hi[1]), false, false);
          const t32_v2u r1 = __builtin_amdgcn_permlane32_swap(pack2bf(lo[2], lo[3]), pack2bf(hi[2], hi[3]), false, false);
          if (rowok) *reinterpret_cast<uint4*>(cp + 16 * hd) = make_uint4(r0.x, r1.x, r0.y, r1.y);
        }
      st_prev = (EPI == EPI_MUL_AUX || !wave_live) ? 0 : (EPI == EPI_GELU_GRAD ? 8 : 4);   // (MUL_AUX: the code loads sit in the queue too; not relaxed)
    } else {
      st_prev = 0;
      // a tile that crosses the end of the row: the general 8-column epilogue
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int hd = 2 * b + hh;
          float lo[4], hi[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { lo[jj] = acc[b][8 * hh + jj]; hi[jj] = acc[b][8 * hh + 4 + jj]; }
          if constexpr (EPI == EPI_QKV_ROPE) {
            if (j * T32_BN < 2 * p.rope_C) {
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const float l = lo[jj], h = hi[jj];
                lo[jj] = fmaf(-h, rs_[jj], l * rc_[jj]);
                hi[jj] = fmaf(l, rs_[jj], h * rc_[jj]);
              }
            }
          }
          float v8[8];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const t32_v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[jj]), __float_as_uint(hi[jj]), false, false);
            v8[jj] = __uint_as_float(r.x);
            v8[4 + jj] = __uint_as_float(r.y);
          }
          const int c0 = j * T32_BN + 16 * hd + 8 * half;
          if constexpr (EPI == EPI_QKV_ROPE) {
            epilogue8<EPI_BF16>(q, row, c0, v8, rowok, j, ntiles);
          } else if constexpr (EPI == EPI_MUL_AUX) {
            epilogue8<EPI>(q, row, c0, v8, rowok, j, ntiles, &codes[hd]);
          } else {
            epilogue8<EPI>(q, row, c0, v8, rowok, j, ntiles);
          }
        }
    }
    T32_T(4);
  };
  for (int j = 0; j < ntiles; ++j)
    t32_call_restrict(tile, j, smem + (j % T32_NS) * T32_STAGE, smem + ((j + 2) % T32_NS) * T32_STAGE);
  T32_TDUMP();
}

// waves per workgroup for M rows: one round of one workgroup per CU
static int t32_waves(int M) { return (cdiv(M, 32) + 255) / 256; }

bool gemm_t32_supported(const GemmArgs& a, int a_f32, int epi) {
  // Opt-in (COATI_T32=1).  Measured at 50 000 rows (tools/rb16_bench.py, profiles/r05_t32_microbench.txt) against the 16-row-slab
  // kernel: plain bf16 N = 768 36.6 vs 32.6 us, N = 1024 45.0 vs 40.5, FC1 NewGELU 70.2 vs 65.2, FC2 input gradient 60.8 vs 43.3,
  // QKV + RoPE 39.0 vs 36.1 -- slower on every shape.  The phase trace (tools/probes/t32_trace.py, profiles/r05_t32_trace.txt): the
  // MFMA loop runs at 85 % of the pipe while two waves share a SIMD (2 414 clocks per tile against 2 048), but it is 46 % of a wave's
  // time; slab load 15 %, tile barrier 16 %, wait for the next tile's DMA 10 %, write-out 13 % (38 % with NewGELU): with 7 waves in
  // lock step per CU nothing runs underneath those phases, and 176-190 VGPRs (64 of them the resident slab) leave no room for more.
  static const bool on = getenv("COATI_T32") != nullptr;
  if (!on || a_f32 || a.K != T32_K || a.m_dev != nullptr) return false;
  if (epi != EPI_BF16 && epi != EPI_QKV_ROPE && epi != EPI_GELU_GRAD && epi != EPI_MUL_AUX && epi != EPI_CE_PARTIAL && epi != EPI_CE_BWD) return false;
  if (epi == EPI_CE_PARTIAL && a.partial_tile != 64) return false;
  if (epi == EPI_QKV_ROPE && (a.rope_hs == 32 || a.rope_C % 64 != 0 || a.rope_pos != nullptr)) return false;
  if (a.N % 16 != 0 && epi != EPI_CE_BWD && epi != EPI_CE_PARTIAL) return false;
  if (a.bias != nullptr && cdiv(a.N, T32_BN) * T32_BN > T32_BIAS_MAX) return false;
  if (a.q8_out != nullptr) return false;
  if (epi == EPI_GELU_GRAD && a.n_store > a.N) return false;
  if (a.ldb != T32_K) return false;
  const int W = t32_waves(a.M);
  return W >= 4 && W <= T32_MAXW;   // 24 577 .. 65 536 rows
}

template <int EPI, bool LN>
static int launch_t32_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_t32_kernel<EPI, LN>;
  constexpr size_t ring = (size_t)T32_NS * T32_STAGE;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ring + T32_BIAS_MAX * 4 + 2048)) != hipSuccess) {
      coati_set_error("gemm_t32: hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int W = t32_waves(a.M);
  const int blocks = cdiv(cdiv(a.M, 32), W);
  const size_t extra = LN ? (size_t)T32_BIAS_MAX * 4 + 2048 : (a.bias != nullptr ? (size_t)cdiv(a.N, T32_BN) * T32_BN * 4 : 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), ring + extra, s, a, W);
  COATI_LAUNCH_CHECK("gemm_t32");
  return COATI_OK;
}

int launch_gemm_t32(const GemmArgs& a, int epi, hipStream_t s) {
  const bool ln = a.ln_x != nullptr;
  if (ln && epi != EPI_QKV_ROPE && epi != EPI_GELU_GRAD) {
    coati_set_error("gemm_t32: epilogue %d has no fused-LayerNorm variant", epi);
    return COATI_EARG;
  }
  switch (epi) {
    case EPI_BF16: return launch_t32_t<EPI_BF16, false>(a, s);
    case EPI_QKV_ROPE: return ln ? launch_t32_t<EPI_QKV_ROPE, true>(a, s) : launch_t32_t<EPI_QKV_ROPE, false>(a, s);
    case EPI_GELU_GRAD: return ln ? launch_t32_t<EPI_GELU_GRAD, true>(a, s) : launch_t32_t<EPI_GELU_GRAD, false>(a, s);
    case EPI_MUL_AUX: return launch_t32_t<EPI_MUL_AUX, false>(a, s);
    case EPI_CE_PARTIAL: return launch_t32_t<EPI_CE_PARTIAL, false>(a, s);
    case EPI_CE_BWD: return launch_t32_t<EPI_CE_BWD, false>(a, s);
    default:
      coati_set_error("gemm_t32: unsupported epilogue %d", epi);
      return COATI_EARG;
  }
}
