// The attention half of a RotaryBlock as ONE sequence-stationary kernel (round 5):
//     xmid = x + c_proj(causal_attention(RoPE(c_attn(ln_1(x)))))            basic_transformer.py:126-154, 171-172
// for d = 256, 16 heads of 16.  The unfused path is three launches per layer (gemm_rb16<QKV_ROPE, LN>, attn_fwd_varlen,
// gemm_ring<RES_F32>) that exist only to round-trip qkv and y through HBM; rows of different sequences never interact and a
// packed sequence is <= 128 rows, so here a workgroup owns a GROUP of whole consecutive sequences (<= 128 rows, built on the
// device by attn_groups_kernel), keeps ln_1(x) in registers, and walks over the 4 head groups (4 heads = 64 c_attn columns each of
// q, k, v):
//
//   per group:   x rows -> LayerNorm (16 rows per wave, lane quad per row) -> a1 (bf16) to HBM (the weight gradient reads it) and,
//                through an LDS staging area, into every wave's RESIDENT slab: 32 rows x 256 k as 16 MFMA B-fragments (64 VGPRs)
//   per head group g (4 weight stages of 32 KiB through a 3-slot LDS ring, global_load_lds, 2 stages ahead):
//     stage Q_g / K_g / V_g  [64 features][256 k]:  D^T = W a1^T on v_mfma_f32_32x32x16_bf16, wave (rb, fh) = row block rb x
//                features 32 fh .. + 31 (two heads): the lane ends with ONE token row and 16 features of it, so bias + RoPE run
//                in registers.  q stays in registers as the B fragment of S^T = K Q^T (v_permlane32_swap regroups the dims) and is
//                dropped into a staging image for the coalesced HBM copy; k, v go to row-major [128][16] LDS images per head.
//     attention (static: wave (rb, fh) owns query rows 32 rb .. + 31 of heads 2 fh, 2 fh + 1 of the group): key blocks from the
//                block that holds the start of the first query's sequence up to rb; S^T = K Q^T (lane = query), block-diagonal
//                causal mask (sequence start <= key <= query), online softmax, O^T += V^T P^T with the LDS transpose read --
//                the loop body of attention.hip.  y (bf16) -> Y image, log-sum-exp -> LDS table.
//     stage P_g  [256 features][64 k]:  xmid_acc[32 rows][128 features] += y_g Wproj[:, 64 g ..]^T  (A = y fragments from the Y
//                image, one head = one k step; 64 accumulator registers live across the 4 head groups)
//   group end:   xmid = x + acc (+ bias) straight from the accumulator layout: one register = one 128-B line of an f32 row.
//   HBM traffic per row: x 1 KiB in (+ 1 KiB re-read from L2 for the residual), a1 512 B, qkv 1.5 KiB, y 512 B, xmid 1 KiB out --
//   qkv and y are written once (the backward reads them) and never read back.
//
// Roles: waves 0..3 issue the weight DMA (8 x 1 KiB per stage each; they issue no global store inside the stage loop, so their
// vmcnt queue holds DMAs only and `s_waitcnt vmcnt(8)` is exact), waves 4..7 copy the finished images (q, k, v, y) and the
// log-sum-exp table to HBM as whole 128-B row segments.  Everything else is symmetric.  Barriers are bare s_barrier behind
// lgkmcnt(0) (a __syncthreads() would drain the DMA queue).
//
// LDS (161 280 B): images K[4] V[4] Y[4] of [128][16] bf16 (48 KiB) | ring 3 x 32 KiB | c_attn bias, ln_1 gamma / beta,
// log-sum-exp table [128][16], its row index.  The a1 staging ([128][256] bf16, 64 KiB) lies over the images (rows 0..95) and the
// ring slot of the stage consumed last (rows 96..127): both are idle between two groups.
#include <cstdlib>
#include "../kernels.h"

#define AB_R 128
#define AB_STAGE 32768
#define AB_NS 3
#define AB_IMG 4096                                     // one head image: 128 rows x 32 B
#define AB_OFF_K 0
#define AB_OFF_V (4 * AB_IMG)
#define AB_OFF_Y (8 * AB_IMG)
#define AB_OFF_RING (12 * AB_IMG)                       // 49 152
#define AB_OFF_BIAS (AB_OFF_RING + AB_NS * AB_STAGE)    // 147 456: c_attn bias, 768 f32
#define AB_OFF_GB (AB_OFF_BIAS + 3072)                  // ln_1 gamma | beta, 512 f32
#define AB_OFF_LSE (AB_OFF_GB + 2048)                   // [128][16] f32
#define AB_OFF_LSEB (AB_OFF_LSE + 8192)                 // [128] int: index of (row, head 0) in lse, -1 = no such row
#define AB_OFF_SQ (AB_OFF_LSEB + 512)                   // [128] int: first row (of the group) of the row's sequence
#define AB_OFF_BP (AB_OFF_SQ + 512)                     // c_proj bias, 256 f32
#define AB_OFF_CNT (AB_OFF_BP + 1024)                   // [4] int: next attention task of each head group
#define AB_LDS_BYTES (AB_OFF_CNT + 16)                  // 162 832
#define AB_STAGES_PER_GROUP 16
#define AB_LOG2E 1.4426950408889634f
#define AB_SCALE 0.25f
#define AB_SCALE_LOG2E 0.36067376022224085f

typedef short ab_v4s __attribute__((ext_vector_type(4)));
typedef short ab_v8s __attribute__((ext_vector_type(8)));
typedef unsigned ab_v2u __attribute__((ext_vector_type(2)));

#ifdef COATI_AB_TRACE
// Probe build (-DCOATI_AB_TRACE, tools/probes/ab_trace.py): shader-clock totals per phase of the waves of the first 16 workgroups:
// [wg][wave][ln, stage wait + barrier, gemm1 (q, k, v products + write-outs), attention, gemm2, image copies, write-out, other barriers]
__device__ unsigned long long ab_trace_buf[16 * 8 * 8];
extern "C" int coati_ab_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ab_trace_buf), sizeof(ab_trace_buf)) == hipSuccess ? 0 : -3;
}
#define AB_T0() unsigned long long ab_t_last = __builtin_amdgcn_s_memtime(), ab_t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define AB_T(i) do { const unsigned long long ab_t_now = __builtin_amdgcn_s_memtime(); ab_t_acc[i] += ab_t_now - ab_t_last; ab_t_last = ab_t_now; } while (0)
#define AB_TDUMP() do { if (blockIdx.x < 16 && lane == 0) { for (int i = 0; i < 8; ++i) ab_trace_buf[(blockIdx.x * 8 + wave) * 8 + i] = ab_t_acc[i]; } } while (0)
#else
extern "C" int coati_ab_trace_read(unsigned long long*) { return -1; }
#define AB_T0() do { } while (0)
#define AB_T(i) do { } while (0)
#define AB_TDUMP() do { } while (0)
#endif


__device__ __forceinline__ void ab_dma16(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
// every LDS operation of this wave has completed, then the workgroup barrier (no vmcnt drain: see the header)
__device__ __forceinline__ void ab_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void ab_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void ab_wait_vm8() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }

__device__ __forceinline__ int ab_arow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ f32x16 ab_zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// [128][16] bf16 image: the 16-B chunk c (dims 8 c .. + 7) of row t sits at chunk position c ^ ((t >> 3) & 1) (attention.hip img_chunk)
__device__ __forceinline__ unsigned ab_img_off(int row, int c) { return (unsigned)(row * 32 + ((c ^ ((row >> 3) & 1)) << 4)); }

// A / B fragment of 32 image rows base .. base + 31 (base % 32 == 0): lane (r = lane & 31, h = lane >> 5) -> row base + r, dims 8 h .. + 7
__device__ __forceinline__ bf16x8 ab_rfrag(const unsigned char* img, int base, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + ab_img_off(base + (lane & 31), lane >> 5));
}
// A fragment X^T[d][row] with the accumulator's row permutation via the LDS transpose read (attention.hip tfrag, HS = 16); base % 16 == 0
__device__ __forceinline__ bf16x8 ab_tfrag(const unsigned char* img, int base, int lane) {
  typedef __attribute__((address_space(3))) ab_v4s lds_v4;
  const int tl = 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int c = (lane & 3) >> 1, h8 = 8 * (lane & 1);   // bytes
  const unsigned char* p = img + ab_img_off(base + tl, c) + h8;
  const unsigned char* p2 = img + ab_img_off(base + tl + 8, c) + h8;
  const ab_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p);
  const ab_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p2);
  const ab_v8s r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}
// The accumulator of a transposed product holds, per lane (token row = lane & 31, h = lane >> 5), dims 4 h + j (u0, u1 packed) and
// 8 + 4 h + j (u2, u3) of one head; the MFMA operand wants dims 8 h .. + 7.  v_permlane32_swap_b32 exchanges the upper half of its
// first operand with the lower half of its second: afterwards lanes < 32 hold dims 0..7, lanes >= 32 dims 8..15, in order.
__device__ __forceinline__ bf16x8 ab_regroup(unsigned u0, unsigned u1, unsigned u2, unsigned u3) {
  const ab_v2u a = __builtin_amdgcn_permlane32_swap(u0, u2, false, false);
  const ab_v2u b = __builtin_amdgcn_permlane32_swap(u1, u3, false, false);
  const uint4 v = make_uint4(a.x, b.x, a.y, b.y);
  return __builtin_bit_cast(bf16x8, v);
}
// max / sum of a value with the one the other half-wave (lane ^ 32) holds: one v_permlane32_swap instead of a trip through the LDS crossbar
__device__ __forceinline__ float ab_half_max(float v) {
  const ab_v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float ab_half_sum(float v) {
  const ab_v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
extern "C" __global__ void ab_probe_swap_kernel(unsigned* out) {   // tests/test_gpu_attn_block.py: pins the semantics assumed above
  const unsigned l = threadIdx.x;
  const ab_v2u a = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
  out[l] = a.x;
  out[64 + l] = a.y;
}

// ---- weight stage (g, which) = 4 g + which of a group -> ring slot at LDS byte address `slot`: DMA wave wd issues pieces wd, wd + 4, ...
// Addressing is a scalar base per piece + ONE 32-bit lane offset per stage kind (offq / offp, computed once per kernel):
//   c_attn stage: rows which * 256 + 64 g + f, f = 0..63: [64][256] bf16, row f at f * 512, its 16-B chunk c at slot c ^ (f & 31).
//     Piece i = rows 2 i, 2 i + 1: lane -> row f = 2 i + (lane >> 5), slot lane & 31, global chunk (lane & 31) ^ (f & 31).  With
//     i = wd + 4 n: f = f0 + 8 n, f0 = 2 wd + (lane >> 5) < 8, so f & 31 = f0 + 8 (n & 3) and the global byte offset is
//     (f0 * 512 + (((lane & 31) ^ f0) << 4)) ^ ((n & 3) << 7), + n * 4096 on the scalar side.
//   c_proj stage: columns 64 g .. + 63 of all 256 rows: [256][64] bf16, row r at r * 128, chunk c at slot c ^ ((r >> 1) & 7).  Piece i =
//     rows 8 i .. + 7: lane -> row 8 i + (lane >> 3), slot lane & 7; (r >> 1) & 7 does not depend on n: one offset, + n * 16384 scalar.
__device__ __forceinline__ void ab_dma16s(const void* base, unsigned off, unsigned lds) {   // scalar base + 32-bit lane byte offset
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
template <int WHICH>
__device__ __forceinline__ void ab_issue_stage(const AttnBlockArgs& p, int g, unsigned slot, int wd, unsigned offq, unsigned offp) {
  if constexpr (WHICH < 3) {
    const char* W = reinterpret_cast<const char*>(p.Wqkv + (long long)(WHICH * 256 + g * 64) * 256);
#pragma unroll
    for (int n = 0; n < 8; ++n) ab_dma16s(W + n * 4096, offq ^ ((n & 3) << 7), slot + (wd + 4 * n) * 1024);
  } else {
    const char* W = reinterpret_cast<const char*>(p.Wproj + g * 64);
#pragma unroll
    for (int n = 0; n < 8; ++n) ab_dma16s(W + n * 16384, offp, slot + (wd + 4 * n) * 1024);
  }
}

__global__ __launch_bounds__(512, 1) void attn_block_fwd_kernel(AttnBlockArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave & 3, fh = wave >> 2;      // row block of 32; feature half (heads 2 fh, 2 fh + 1 of a head group)
  const int tl = lane & 31, half = lane >> 5;
  const bool dma_wave = wave < 4;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u8*)smem);
  float* const biasS = reinterpret_cast<float*>(smem + AB_OFF_BIAS);
  float* const GB = reinterpret_cast<float*>(smem + AB_OFF_GB);
  float* const lseT = reinterpret_cast<float*>(smem + AB_OFF_LSE);
  int* const lseB = reinterpret_cast<int*>(smem + AB_OFF_LSEB);
  int* const sqT = reinterpret_cast<int*>(smem + AB_OFF_SQ);
  // (Walking the head groups from a different one in every workgroup -- so that the CUs of an XCD do not all pull the same weight lines
  //  at the same time -- measured nothing, and it makes a row's c_proj sum depend on which workgroup its group lands on: not kept.)
  constexpr int rot = 0;

  const int ngroups = p.grp[0];
  const int G = gridDim.x;
  const int mine = (ngroups - (int)blockIdx.x + G - 1) / G;   // groups blockIdx.x, + G, ...
  if (mine <= 0) return;
  const int total_stages = mine * AB_STAGES_PER_GROUP;

  for (int c = tid; c < 768; c += 512) biasS[c] = p.bqkv[c];
  GB[tid] = tid < 256 ? p.ln_g[tid] : p.ln_b[tid - 256];
  if (tid < 256) reinterpret_cast<float*>(smem + AB_OFF_BP)[tid] = p.bproj[tid];
  // DMA lane offsets (see ab_issue_stage); the first two stages of the first group (ring slots 0, 1) are in flight during the first LayerNorm
  unsigned offq, offp;
  {
    const int f0 = 2 * (wave & 3) + (lane >> 5);
    offq = (unsigned)(f0 * 512 + (((lane & 31) ^ f0) << 4));
    const int r0 = 8 * (wave & 3) + (lane >> 3);
    offp = (unsigned)(r0 * 512 + (((lane & 7) ^ ((r0 >> 1) & 7)) << 4));
  }
  if (dma_wave) {
    ab_issue_stage<0>(p, rot, lds0 + AB_OFF_RING, wave, offq, offp);
    ab_issue_stage<1>(p, rot, lds0 + AB_OFF_RING + AB_STAGE, wave, offq, offp);
  }
  AB_T0();

  int S = 0;   // stages consumed so far by this workgroup (ring slot = S % 3)
  for (int gi = 0; gi < mine; ++gi) {
    const int grp = blockIdx.x + gi * G;
    const int row0 = p.grp[1 + grp], nrows = p.grp[2 + grp] - row0;   // 1 .. 128 rows

    // lane / thread index the compiler cannot see through: index arithmetic of the once-per-group phases (LayerNorm rows, image copies,
    // write-out) is recomputed per group instead of being hoisted out of the loop and spilled (the MFMA phases hold ~ 230 VGPRs)
    int lane_o = lane, tid_o = tid;
    asm volatile("" : "+v"(lane_o), "+v"(tid_o));

    // ================= LayerNorm: wave w normalises rows 16 w .. + 15 (lane quad (fr, kq) per row, as gemm_rb16.hip) =============
    const int fr = lane_o & 15, kq = lane_o >> 4;
    const int lrow = 16 * wave + fr;
    const bool lrow_ok = lrow < nrows;
    const long long lgrow = row0 + (lrow_ok ? lrow : nrows - 1);
    float xf[8][8];
    {
      const float* xp = p.x + lgrow * 256 + kq * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const float4 x0 = *reinterpret_cast<const float4*>(xp + ks * 32), x1 = *reinterpret_cast<const float4*>(xp + ks * 32 + 4);
        xf[ks][0] = x0.x; xf[ks][1] = x0.y; xf[ks][2] = x0.z; xf[ks][3] = x0.w; xf[ks][4] = x1.x; xf[ks][5] = x1.y; xf[ks][6] = x1.z; xf[ks][7] = x1.w;
      }
    }
    // per-lane metadata of the MFMA phases: token row 32 rb + tl of the group
    const int mrow = 32 * rb + tl;
    const bool mrow_ok = mrow < nrows;
    const long long mgrow = row0 + (mrow_ok ? mrow : nrows - 1);
    int t_pos = 0, lse_idx = -1;
    {
      const int src = p.row_src != nullptr ? p.row_src[mgrow] : (int)mgrow;
      const int b = src / p.Tl;
      if (mrow_ok) { t_pos = src - b * p.Tl; lse_idx = b * 16 * p.Tl + t_pos; }
    }
    const int s_q = mrow - t_pos;   // first row (of the group) of this query's sequence; invalid rows attend to themselves only
    float rc_[4], rs_[4];
    {
      const float4 c4 = *reinterpret_cast<const float4*>(p.cos_t + t_pos * 16 + 4 * half);
      const float4 s4 = *reinterpret_cast<const float4*>(p.sin_t + t_pos * 16 + 4 * half);
      rc_[0] = c4.x; rc_[1] = c4.y; rc_[2] = c4.z; rc_[3] = c4.w; rs_[0] = s4.x; rs_[1] = s4.y; rs_[2] = s4.z; rs_[3] = s4.w;
    }
    float mean, rstd;
    {
      float sm = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) sm += xf[ks][i];
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      mean = sm * (1.0f / 256);
      float vs = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = xf[ks][i] - mean; vs = fmaf(d, d, vs); }
      vs += __shfl_xor(vs, 16, 64);
      vs += __shfl_xor(vs, 32, 64);
      rstd = rsqrtf(vs * (1.0f / 256) + 1e-5f);
    }
    // every wave is done with the previous group's images, ring slot and tables (their copies to HBM included)
    ab_barrier();
    AB_T(7);
    if (lrow_ok && kq == 0) { p.mean[lgrow] = mean; p.rstd[lgrow] = rstd; }
    if (fh == 0 && half == 0) { lseB[mrow] = lse_idx; sqT[mrow] = s_q; }
    if (tid < 4) reinterpret_cast<int*>(smem + AB_OFF_CNT)[tid] = 0;
    {
      // staging: row r at (r < 96 ? images : free ring slot) + r' * 512, chunk c (k = 8 c .. + 7) at slot c ^ (r & 31)
      const int free_slot = (S + 2) % AB_NS;
      unsigned char* const stg = lrow < 96 ? smem + lrow * 512 : smem + AB_OFF_RING + free_slot * AB_STAGE + (lrow - 96) * 512;
      bf16_t* const op = p.a1 + lgrow * 256 + kq * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const float4 g0 = *reinterpret_cast<const float4*>(GB + ks * 32 + kq * 8), g1 = *reinterpret_cast<const float4*>(GB + ks * 32 + kq * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(GB + 256 + ks * 32 + kq * 8), b1 = *reinterpret_cast<const float4*>(GB + 256 + ks * 32 + kq * 8 + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (xf[ks][i] - mean) * rstd * g[i] + bt[i];
        uint4 u = pack8(o);
        if (lrow_ok) *reinterpret_cast<uint4*>(op + ks * 32) = u;
        else u = make_uint4(0, 0, 0, 0);   // rows past the group: zero operand rows (their products stay finite, nothing of them is stored)
        *reinterpret_cast<uint4*>(stg + (((4 * ks + kq) ^ (lrow & 31)) << 4)) = u;
      }
    }
    // the residual x in the accumulator layout of the c_proj product (register r = row 32 rb + arow(r), lane = feature 128 fh + 32 j + tl):
    // read here (L2 hits: the LayerNorm has just read these rows), in flight during the slab load and the first stages, so that the
    // write-out at the end of the group waits for nothing (the bias is added there, from LDS)
    f32x16 acc2[4];
    {
      const char* const res = reinterpret_cast<const char*>(p.x + (long long)row0 * 256 + fh * 128);
      const int tl_o = lane_o & 31, rbase = 32 * rb + 4 * (lane_o >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        const unsigned ro = (unsigned)(((row < nrows ? row : nrows - 1) * 256 + tl_o) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[j][r] = *reinterpret_cast<const float*>(res + ro + j * 128);
      }
    }
    ab_barrier();
    // resident slab: fragment ks = k 16 ks + 8 half .. + 7 of row 32 rb + tl
    bf16x8 a1f[16];
    {
      const int free_slot = (S + 2) % AB_NS;
      const unsigned char* const stg = mrow < 96 ? smem + mrow * 512 : smem + AB_OFF_RING + free_slot * AB_STAGE + (mrow - 96) * 512;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) a1f[ks] = *reinterpret_cast<const bf16x8*>(stg + (((2 * ks + half) ^ tl) << 4));
    }
    AB_T(0);

#pragma unroll 1
    for (int gl = 0; gl < 4; ++gl) {
    const int g = (gl + rot) & 3;
#pragma unroll
    for (int which = 0; which < 4; ++which) {
      const int s = 4 * gl + which;
      // ---- stage S has landed (the DMA waves' pieces; the barrier covers the rest); every wave is done with stage S - 1
      if (dma_wave) {
        if (s == 0) ab_wait_vm0();                       // (+ this wave's a1 / statistics stores of the LayerNorm phase)
        else if (s >= 2) { if (S + 1 < total_stages) ab_wait_vm8(); else ab_wait_vm0(); }
      }
      ab_barrier();
      if (dma_wave && S + 2 < total_stages) {
        // two stages ahead: (g, which + 2) or (g + 1, which - 2) -- the next group's first stages at s = 14, 15
        const int g2 = (g + (which >= 2 ? 1 : 0)) & 3;   // (the walk is cyclic: behind the last head group comes the next group's first = rot)
        const unsigned slot2 = lds0 + AB_OFF_RING + ((S + 2) % AB_NS) * AB_STAGE;
        switch ((which + 2) & 3) {   // (compile-time after unrolling)
          case 0: ab_issue_stage<0>(p, g2, slot2, wave, offq, offp); break;
          case 1: ab_issue_stage<1>(p, g2, slot2, wave, offq, offp); break;
          case 2: ab_issue_stage<2>(p, g2, slot2, wave, offq, offp); break;
          default: ab_issue_stage<3>(p, g2, slot2, wave, offq, offp); break;
        }
      }
      AB_T(1);
      const unsigned char* const St = smem + AB_OFF_RING + (S % AB_NS) * AB_STAGE;
      ++S;
      if (which < 3) {
        // ======== D^T[feature][token] = W_stage a1^T: this wave's 32 features (2 heads) x its 32 token rows, K = 256 =========
        if (which == 1 && !dma_wave) {
          // the q staging image of this head group is complete (barrier above): its coalesced copy to HBM.  Thread -> (row = t >> 3
          // (+ 32 per turn), head (t >> 1) & 3, 16-B chunk t & 1): uniform base + one 32-bit offset (saddr form), nothing hoisted
          {
            const int t = tid_o - 256, row = t >> 3, hd = (t >> 1) & 3, ch = t & 1;
            const unsigned char* const src = smem + AB_OFF_Y + hd * AB_IMG + ab_img_off(row, ch);
            const unsigned go = (unsigned)((row * 768 + hd * 16 + ch * 8) * 2);
            char* const dst = reinterpret_cast<char*>(p.qkv + (long long)row0 * 768 + g * 64);
            uint4 v[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const uint4*>(src + it * 1024);
#pragma unroll
            for (int it = 0; it < 4; ++it)
              if (row + 32 * it < nrows) *reinterpret_cast<uint4*>(dst + it * (32 * 768 * 2) + go) = v[it];
          }
          AB_T(5);
        }
        f32x16 c0, c1 = ab_zero16();
        {
          const float* bq = biasS + which * 256 + g * 64 + fh * 32 + 4 * half;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const float4 b4 = *reinterpret_cast<const float4*>(bq + 8 * rg);
            c0[4 * rg] = b4.x; c0[4 * rg + 1] = b4.y; c0[4 * rg + 2] = b4.z; c0[4 * rg + 3] = b4.w;
          }
        }
        {
          const unsigned char* const wrow = St + (32 * fh + tl) * 512;
          bf16x8 wf[4];
#pragma unroll
          for (int i = 0; i < 3; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wrow + (((2 * i + half) ^ tl) << 4));
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            if (ks + 3 < 16) wf[(ks + 3) & 3] = *reinterpret_cast<const bf16x8*>(wrow + (((2 * (ks + 3) + half) ^ tl) << 4));
            if (ks & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 3], a1f[ks], c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 3], a1f[ks], c0, 0, 0, 0);
          }
        }
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = c0[r] + c1[r];
        // register r = 4 rg + j: feature 8 rg + 4 half + j of the 32 = head (rg >> 1), dim 8 (rg & 1) + 4 half + j
        if (which < 2) {
          // RotaryEmbedding.rotary_embed (basic_transformer.py:83-100): y_i = x_i c_i - x_{i+8} s_i, y_{i+8} = x_{i+8} c_i + x_i s_i
#pragma unroll
          for (int hd = 0; hd < 2; ++hd)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float lo = v[8 * hd + j], hi = v[8 * hd + 4 + j];
              v[8 * hd + j] = fmaf(-hi, rs_[j], lo * rc_[j]);
              v[8 * hd + 4 + j] = fmaf(lo, rs_[j], hi * rc_[j]);
            }
        }
#pragma unroll
        for (int hd = 0; hd < 2; ++hd) {
          const unsigned u0 = pack2bf(v[8 * hd], v[8 * hd + 1]), u1 = pack2bf(v[8 * hd + 2], v[8 * hd + 3]);
          const unsigned u2 = pack2bf(v[8 * hd + 4], v[8 * hd + 5]), u3 = pack2bf(v[8 * hd + 6], v[8 * hd + 7]);
          if (which == 0) {
            *reinterpret_cast<bf16x8*>(smem + AB_OFF_Y + (2 * fh + hd) * AB_IMG + ab_img_off(mrow, half)) = ab_regroup(u0, u1, u2, u3);
          } else {
            unsigned char* const img = smem + (which == 1 ? AB_OFF_K : AB_OFF_V) + (2 * fh + hd) * AB_IMG;
            *reinterpret_cast<uint2*>(img + ab_img_off(mrow, 0) + half * 8) = make_uint2(u0, u1);
            *reinterpret_cast<uint2*>(img + ab_img_off(mrow, 1) + half * 8) = make_uint2(u2, u3);
          }
        }
        AB_T(2);
        if (which == 2) {
          // ======== attention of head group g: query rows 32 rb .. + 31, heads 2 fh, 2 fh + 1; k, v images complete behind this barrier
          ab_barrier();
          AB_T(7);
          // 16 tasks (query block, head) per head group; a task costs one (S, PV) pair per key block between the start of its first
          // query's sequence and itself -- 1 for block 0, up to 3 - 4 for the later ones.  The waves draw them from an LDS counter,
          // costly blocks first (the waves that also carry the image copies simply end up with fewer).  q comes back from its staging
          // image; y replaces it there (only this task touches these rows of this head)
          int* const cnt = reinterpret_cast<int*>(smem + AB_OFF_CNT) + gl;
          for (;;) {
            int task = 0;
            if (lane == 0) task = atomicAdd(cnt, 1);
            task = __builtin_amdgcn_readfirstlane(task);
            if (task >= 16) break;
            const int qb = 3 - (task >> 2), hq = task & 3;
            const unsigned char* const Kimg = smem + AB_OFF_K + hq * AB_IMG;
            const unsigned char* const Vimg = smem + AB_OFF_V + hq * AB_IMG;
            unsigned char* const Yimg = smem + AB_OFF_Y + hq * AB_IMG;
            const int qrow = 32 * qb + tl;
            const int sq = sqT[qrow];
            const bf16x8 qfr = ab_rfrag(Yimg, 32 * qb, lane);
            const int s_first = __builtin_amdgcn_readfirstlane(sq);                 // lane 0: the earliest sequence start of the block
            const int s_last = __builtin_amdgcn_readlane(sq, 31);                   // (non-decreasing over the rows)
            const int kb_lo = s_first >> 5;
            float m_run = -1e30f, l_run = 0.f;    // finite: a key block that is masked out entirely for this query leaves alpha = 1, p = 0
            f32x16 o = ab_zero16();
            for (int kb = kb_lo; kb <= qb; ++kb) {
              const f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab_rfrag(Kimg, kb * 32, lane), qfr, ab_zero16(), 0, 0, 0);
              // The running maximum is taken over ALL 32 keys of the block, masked ones included: any upper bound of the valid scores
              // keeps the exponentials <= 1, the log-sum-exp is exact for whatever bound was used, and the masked keys are real rows of
              // neighbouring sequences (scores of the same magnitude).  The mask -- sequence start <= key <= query, as a bit field over
              // the block's keys -- then only clears probabilities: two instructions per element instead of two compares and a select.
              float mloc = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
              for (int r = 3; r < 16; r += 2) mloc = fmaxf(fmaxf(mloc, sc[r]), r + 1 < 16 ? sc[r + 1] : sc[r]);
              mloc = ab_half_max(mloc);
              const float m_new = fmaxf(m_run, mloc);
              const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * AB_SCALE_LOG2E);
              const float mc = m_new * AB_SCALE_LOG2E;
              float pr[16];
              float lsum = 0.f;
              if (kb == qb || kb * 32 < s_last) {     // (uniform) the diagonal block, or one that some query's sequence starts inside / behind
                const int lo_c = sq - 32 * kb, hi_c = qrow - 32 * kb;          // valid key positions of the block: max(lo_c, 0) .. min(hi_c, 31)
                unsigned msk = (0xffffffffu >> (31 - (hi_c < 31 ? hi_c : 31))) & (lo_c > 0 ? 0xffffffffu << (lo_c < 32 ? lo_c : 31) : 0xffffffffu);
                msk = lo_c >= 32 ? 0u : msk;
                msk >>= 4 * half;              // register r <-> key position (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const int keep = __builtin_amdgcn_sbfe((int)msk, (r & 3) + 8 * (r >> 2), 1);     // v_bfe_i32: 0 or ~0
                  const float e = __builtin_amdgcn_exp2f(fmaf(sc[r], AB_SCALE_LOG2E, -mc));
                  pr[r] = __uint_as_float(__float_as_uint(e) & (unsigned)keep);
                  lsum += pr[r];
                }
              } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  pr[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], AB_SCALE_LOG2E, -mc));
                  lsum += pr[r];
                }
              }
              lsum = ab_half_sum(lsum);
              l_run = l_run * alpha + lsum;
              m_run = m_new;
#pragma unroll
              for (int r = 0; r < 8; ++r) o[r] *= alpha;   // dims < 16: registers 0..7
              const uint4 p0 = pack8(pr), p1 = pack8(pr + 8);
              o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab_tfrag(Vimg, kb * 32, lane), __builtin_bit_cast(bf16x8, p0), o, 0, 0, 0);
              o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab_tfrag(Vimg, kb * 32 + 16, lane), __builtin_bit_cast(bf16x8, p1), o, 0, 0, 0);
            }
            const float inv = __builtin_amdgcn_rcpf(l_run);
            const unsigned u0 = pack2bf(o[0] * inv, o[1] * inv), u1 = pack2bf(o[2] * inv, o[3] * inv);
            const unsigned u2 = pack2bf(o[4] * inv, o[5] * inv), u3 = pack2bf(o[6] * inv, o[7] * inv);
            *reinterpret_cast<bf16x8*>(Yimg + ab_img_off(qrow, half)) = ab_regroup(u0, u1, u2, u3);
            if (half == 0) lseT[qrow * 16 + g * 4 + hq] = m_run * AB_SCALE + __logf(l_run);
          }
          AB_T(3);
        }
      } else {
        // ======== xmid_acc[token][feature] += y_g Wproj[:, 64 g .. + 63]^T: rows 32 rb .. + 31, features 128 fh .. + 127 =========
        bf16x8 yf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) yf[ks] = ab_rfrag(smem + AB_OFF_Y + ks * AB_IMG, 32 * rb, lane);
        const unsigned char* const wrow = St + (128 * fh + tl) * 128;
        const int sw = (tl >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          bf16x8 wf[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wrow + j * 4096 + (((2 * ks + half) ^ sw) << 4));
#pragma unroll
          for (int j = 0; j < 4; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[ks], wf[j], acc2[j], 0, 0, 0);
        }
        AB_T(4);
        if (!dma_wave) {
          // k, v, y images of this head group -> HBM as whole 128-B row segments (addressing as the q copy above)
          {
            const int t = tid_o - 256, row = t >> 3, hd = (t >> 1) & 3, ch = t & 1;
            const unsigned char* const src = smem + hd * AB_IMG + ab_img_off(row, ch);
            const unsigned go = (unsigned)((row * 768 + hd * 16 + ch * 8) * 2), gy = (unsigned)((row * 256 + hd * 16 + ch * 8) * 2);
            char* const dq = reinterpret_cast<char*>(p.qkv + (long long)row0 * 768 + g * 64);
            char* const dy = reinterpret_cast<char*>(p.y + (long long)row0 * 256 + g * 64);
            // (all twelve LDS reads first, then the stores: one LDS round trip instead of four)
            uint4 vk[4], vv[4], vy[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              vk[it] = *reinterpret_cast<const uint4*>(src + AB_OFF_K + it * 1024);
              vv[it] = *reinterpret_cast<const uint4*>(src + AB_OFF_V + it * 1024);
              vy[it] = *reinterpret_cast<const uint4*>(src + AB_OFF_Y + it * 1024);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              if (row + 32 * it < nrows) {
                *reinterpret_cast<uint4*>(dq + it * (32 * 768 * 2) + go + 512) = vk[it];
                *reinterpret_cast<uint4*>(dq + it * (32 * 768 * 2) + go + 1024) = vv[it];
                *reinterpret_cast<uint4*>(dy + it * (32 * 256 * 2) + gy) = vy[it];
              }
            }
          }
          AB_T(5);
        }
      }
    }
    }

    // ================= group end: xmid = acc2 + bias (acc2 = x + the product: the residual was folded in at the start of the group):
    // one register = one 128-B line of an f32 row; log-sum-exp table -> HBM ==================
    {
      char* const out = reinterpret_cast<char*>(p.xmid + (long long)row0 * 256 + fh * 128);
      int lane_w = lane;
      asm volatile("" : "+v"(lane_w));   // (the row offsets are recomputed here, not kept alive from the start of the group)
      const int tl_o = lane_w & 31, rbase = 32 * rb + 4 * (lane_w >> 5);
      float bpj[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bpj[j] = reinterpret_cast<const float*>(smem + AB_OFF_BP)[fh * 128 + j * 32 + tl_o];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < nrows) {
          const unsigned ro = (unsigned)((row * 256 + tl_o) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<float*>(out + ro + j * 128) = acc2[j][r] + bpj[j];
        }
      }
    }
    if (!dma_wave) {
      int base[8];
      float lv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = (tid_o - 256) + 256 * it;
        base[it] = lseB[idx >> 4];
        lv[it] = lseT[idx];
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int h = ((tid_o - 256) + 256 * it) & 15;
        if (base[it] >= 0) p.lse[base[it] + h * p.Tl] = lv[it];
      }
    }
    AB_T(6);
  }
  AB_TDUMP();
}

// ---- groups of whole consecutive sequences with <= cap rows (greedy), built on the device: no host sync ------------------------
// node(0) = sequence 0, node(p + 1) = J(node(p)) with J(i) = the last j > i whose rows off[i] .. off[j] still fit; the chain is
// unrolled by pointer doubling (log2 B rounds).  grp[0] = number of groups, grp[1 + p] = first row of group p.
#define ABG_MAXB 4096
__global__ __launch_bounds__(1024) void attn_groups_kernel(const int* __restrict__ seq_off, int B, int Tl, int cap, int* __restrict__ grp) {
  __shared__ int offs[ABG_MAXB + 1], J[ABG_MAXB + 1], J2[ABG_MAXB + 1], node[ABG_MAXB + 1];
  const int tid = threadIdx.x;
  for (int i = tid; i <= B; i += 1024) offs[i] = seq_off != nullptr ? seq_off[i] : i * Tl;
  __syncthreads();
  for (int i = tid; i <= B; i += 1024) {
    int j = B;
    if (i < B) {
      int lo = i + 1, hi = B;   // the largest j in [i + 1, B] with offs[j] - offs[i] <= cap (j = i + 1 always fits: every sequence is <= cap rows)
      const int lim = offs[i] + cap;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (offs[mid] <= lim) lo = mid; else hi = mid - 1;
      }
      j = lo;
    }
    J[i] = j;
    node[i] = B;
  }
  __syncthreads();
  if (tid == 0) node[0] = 0;
  __syncthreads();
  for (int len = 1; len <= B; len <<= 1) {
    for (int q = tid; q < len && q + len <= B; q += 1024) node[q + len] = J[node[q]];
    for (int i = tid; i <= B; i += 1024) J2[i] = J[J[i]];
    __syncthreads();
    for (int i = tid; i <= B; i += 1024) J[i] = J2[i];
    __syncthreads();
  }
  for (int q = tid; q <= B; q += 1024) {
    const int nd = node[q];
    grp[1 + q] = offs[nd];
    // groups with zero rows (all-[PAD] sequences at the very end) do not exist: the count stops at the last row
    if (offs[nd] < offs[B] && (q == B || offs[node[q + 1]] >= offs[B])) grp[0] = q + 1;
  }
  if (tid == 0 && offs[B] == 0) grp[0] = 0;
}

// Opt-in (COATI_ATTN_BLOCK=1).  Measured in the step (B = 1024, ~ 50 000 packed rows per pass, profiles/r05_attn_block_*.txt):
// 3.7 - 4.0 ms for the 32 launches of a step against 3.75 ms for the 96 launches it replaces (qkv_fwd 1.57 + attn_fwd 1.08 + proj_fwd
// 1.10), and a persistent workgroup that holds all of a CU's LDS and registers leaves no room for the point encoder's side-stream
// kernels: 22.35 vs 21.82 ms per step on one box.  The phase trace says where a group's ~ 85 000 cycles go: attention 32 % (VALU
// bound: 16 exponentials + ~ 90 VALU per lane and (key block, query block) pair, 4 query blocks x 16 heads x ~ 2 pairs per group),
// the c_attn products 21 %, stage waits + barriers 26 %, LayerNorm + slab 9 %, c_proj 5 %, copies + write-out 9 %.
// On COLD caches (as in the step: tools/probes/ab_trace.py cold) a launch takes 116.8 us against 102.7 with the rows in the Infinity
// Cache: every workgroup reads its group's 128 KiB at the same moment, a 32-MB burst in which no CU computes (the three-launch path
// spreads the same reads over thousands of workgroups).  Prefetching the NEXT group's rows into L2 from the copy waves (one dword per
// line by global_load_lds into a sink) was built and measured: 114.9 us -- only every second group has a predecessor to prefetch under.
bool attn_block_fwd_supported(int B, int T, int C, int n_head) {
  static const bool on = getenv("COATI_ATTN_BLOCK") != nullptr && getenv("COATI_ATTN_BLOCK")[0] == '1';
  return on && C == 256 && n_head == 16 && T <= AB_R && B <= ABG_MAXB && B > 0;
}

int launch_attn_groups(const int* seq_off, int B, int T, int* grp, hipStream_t s) {
  COATI_CHECK_ARG(grp, "attn_groups: null table");
  COATI_CHECK_SHAPE(B > 0 && B <= ABG_MAXB && T > 0 && T <= AB_R, "attn_groups: unsupported shape B=%d T=%d", B, T);
  hipLaunchKernelGGL(attn_groups_kernel, dim3(1), dim3(1024), 0, s, seq_off, B, T, AB_R, grp);
  COATI_LAUNCH_CHECK("attn_groups");
  return COATI_OK;
}

int launch_attn_block_fwd(const AttnBlockArgs& a, hipStream_t s) {
  COATI_CHECK_ARG(a.x && a.xmid && a.ln_g && a.ln_b && a.mean && a.rstd && a.a1 && a.Wqkv && a.bqkv && a.Wproj && a.bproj && a.qkv && a.y &&
                      a.lse && a.cos_t && a.sin_t && a.grp,
                  "attn_block_fwd: null operand");
  COATI_CHECK_SHAPE(a.M > 0 && a.Tl > 0 && a.Tl <= AB_R, "attn_block_fwd: unsupported shape M=%d T=%d", a.M, a.Tl);
  static bool attr_set = false;
  static int n_cu = 256;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, AB_LDS_BYTES) != hipSuccess) {
      coati_set_error("attn_block_fwd: hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_block_fwd_kernel, dim3(n_cu), dim3(512), AB_LDS_BYTES, s, a);
  COATI_LAUNCH_CHECK("attn_block_fwd");
  return COATI_OK;
}

int launch_ab_probe_swap(unsigned* out, hipStream_t s) {
  hipLaunchKernelGGL(ab_probe_swap_kernel, dim3(1), dim3(64), 0, s, out);
  COATI_LAUNCH_CHECK("ab_probe_swap");
  return COATI_OK;
}
