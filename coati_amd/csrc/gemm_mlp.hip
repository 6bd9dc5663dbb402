// Chained two-GEMM kernel for the transformer MLP at C = 256 (reference basic_transformer.py:157-174, RotaryBlock.mlpf:
// x + W2 NewGELU(W1 ln_2(x) + b1) + b2) and for the matching input-gradient chain of the backward.
//
//   forward  (FWD):  a = LayerNorm(x)              (bf16, saved: the weight gradient of W1 reads it)
//                    pre = a W1^T + b1             (never leaves the chip)
//                    g = NewGELU(pre)              (bf16, saved: W2's weight gradient reads it)
//                    d = NewGELU'(pre)             (8-bit fixed point, saved: the backward multiplies by it)
//                    out = x + g W2^T + b2         (f32 residual stream)
//   backward (!FWD): dg = dY W1^T                  (W1 := fc2 weight transposed, [Hd, C])
//                    dh = dg * d                   (bf16, saved: both MLP weight gradients read it)
//                    out = dh W2^T                 (W2 := fc1 weight transposed, [C, Hd]; bf16: the LayerNorm backward reads it)
//
// Versus the two separate GEMM launches this removes one trip of the [M, 4C] intermediate through HBM per direction
// (forward: g is not re-read and x is read ONCE -- no residual re-read; backward: dh is not re-read).
//
// Both products are computed TRANSPOSED with v_mfma_f32_16x16x32_bf16 so that the first product's accumulator layout IS the
// second product's B-operand layout and the activation runs in registers:
//   GEMM1^T: pre^T[h, t] = sum_k W1[h, k] a[t, k]   A-operand = weight rows (LDS), B-operand = the wave's 16 token rows, resident
//            in registers (8 k-steps x bf16x8);  lane (t = lane & 15, q = lane >> 4) ends up with rows 4q .. 4q+3 of each
//            16-row output tile.
//   The two tiles of a 32-unit chunk are given the hidden units {8q' + r} and {8q' + 4 + r} (q' = 0..3, r = 0..3) as their
//   rows 4q' + r -- a free choice of which weight row a lane reads -- so lane (t, q) holds the 8 CONSECUTIVE hidden units
//   8q .. 8q+7 of token t: exactly the k-slots of the B operand of
//   GEMM2^T: out^T[c, t] += sum_h W2[c, h] g[t, h]   A-operand = W2 rows (LDS), one k-step of 32 per chunk, 16 column tiles.
// The same 8 values are 16 contiguous bytes of the saved [M, Hd] tensors.
// Weights stream HBM/L2 -> LDS with global_load_lds_dwordx4 in 32-hidden-unit chunks (W1 chunk [32][256] + W2 chunk
// [256][32] = 32 KiB) through a 4-slot ring; 16-byte pieces are XOR-swizzled on the global side so that every fragment
// ds_read_b128 is conflict-free (W1: position = chunk ^ tile-row; W2: position = q ^ ((-row >> 2) & 3)).
//
// History (round 2): version 1 of this kernel (one 10-wave workgroup per CU: load 160 rows -> 32 chunks -> store 160 rows)
// was correct at the first run but slower than the two launches it replaces (229 / 148 us against 198 / 145 us): every CU
// was in the same phase at the same time, so HBM idled during the chunk loops and the MFMAs during the row phases
// (ablations: skeleton without the two products 140 us, products alone 60 us, the 1-KiB-per-wave saved-tensor stores and
// derivative loads +50 us each).  Version 2 (below) overlaps the phases: two 5-wave consumer groups half a period apart
// share ONE weight stream issued by two producer waves.  It is correct (test_mlp_chain) and spill-free in the chunk loop,
// but measures 244 / 151 us against 191 / 131 us for the two launches (profiles/r02_mlp_chain_bench.txt), so the engine
// does NOT use it.  What bounds it: a wave owns 16 token rows, so every v_mfma_f32_16x16x32 (16 cycles) takes a fresh
// 1-KiB weight fragment from LDS -- 10 waves x 32 KiB per chunk stage = 320 KiB through a 128..256 B/clk LDS port is
// 1250..2500 cycles against 1280 cycles of MFMA issue per SIMD; the measured stage is ~4000 cycles.  The two-launch GEMMs
// reuse each LDS fragment over 64 rows.  The way forward is 32 token rows per wave (32x32x16 tiles, 128 accumulator +
// 64 operand registers: a 256-register, 8-wave design) -- not built.
#include <cstdlib>
#include "gemm_epi.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MLP_C 256
#define MLP_CH 32                              // hidden units per stage
#define MLP_W1_HALFS (MLP_CH * MLP_C)          // [32][256]
#define MLP_W2_HALFS (MLP_C * MLP_CH)          // [256][32]
#define MLP_STAGE_HALFS (MLP_W1_HALFS + MLP_W2_HALFS)
#define MLP_RING 4                             // ring slots (prefetch distance MLP_RING - 1 chunks)

// =====================================================================================================================
// The software-pipelined, wave-specialised kernel.  A persistent workgroup of 12 waves is split by ROLE:
//   * waves 10, 11 are PRODUCERS: they only stream the weight chunks into the 4-slot LDS ring, cyclically (chunk = stage mod
//     nst), for the whole kernel: 16 one-KiB DMAs per wave and stage, the only hand-placed vmcnt in the kernel;
//   * waves 0-4 and 5-9 are two CONSUMER groups; a group owns 80 rows (16 per wave) at a time and runs the fixed schedule
//     [issue the row loads | 3 stages of slack | LayerNorm | nst chunk stages | stores], the second group HALF A PERIOD
//     behind the first.  While one group waits for its rows or drains its stores the other one multiplies: the memory phases
//     hide behind the other group's compute, and both groups consume the SAME weight stream (any nst consecutive stages see
//     every chunk once, and the order of the hidden units does not matter for the sum).
// One s_barrier per stage for all 12 waves.  The consumers issue no DMA, so the compiler's own waitcnt placement is exact for
// them.  Forward: the fp32 row x is loaded straight into the GEMM2 accumulator registers (lane (t, q) <-> columns
// 16 ct + 4 q + r), the LayerNorm reads it there, and the accumulator is then initialised to x + b2 -- the residual is never
// read a second time, and the row costs no extra registers.  That layout fixes the k-order of the first product's B operand
// (slot (q, i) <-> channel 32 s + 16 (i >> 2) + 4 q + (i & 3)), so the fc1 weight is read from a copy whose columns are
// permuted the same way inside every group of 32 (mlp_permute_w1: one more bf16 shadow of the weight).
// =====================================================================================================================
#define MP_CW 10
#define MP_PW 2
#define MP_GW 5
#define MP_DP 3                                // stages between issuing a block's row loads and using them
#define MP_ROWS_WG (4 * MP_GW * 16)            // rows per workgroup: two consecutive blocks of 80 rows per group

// column permutation of the forward kernel's W1 copy: position 32 s + 8 q + i <- channel 32 s + 16 (i >> 2) + 4 q + (i & 3)
__global__ void mlp_permute_w1_kernel(const bf16_t* __restrict__ W, long long ldw, bf16_t* __restrict__ Wp, long long ldp, int Hd, int C) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Hd * C) return;
  const int h = (int)(idx / C), pos = (int)(idx % C);
  const int s = pos >> 5, qq = (pos >> 3) & 3, i = pos & 7;
  Wp[(long long)h * ldp + pos] = W[(long long)h * ldw + 32 * s + 16 * (i >> 2) + 4 * qq + (i & 3)];
}
int launch_mlp_permute_w1(const bf16_t* W, long long ldw, bf16_t* Wp, long long ldp, int Hd, int C, hipStream_t s) {
  COATI_CHECK_ARG(W && Wp, "mlp_permute_w1: null operand");
  COATI_CHECK_SHAPE(C % 32 == 0, "mlp_permute_w1: C must be a multiple of 32");
  hipLaunchKernelGGL(mlp_permute_w1_kernel, dim3(cdiv((long long)Hd * C, 256)), dim3(256), 0, s, W, ldw, Wp, ldp, Hd, C);
  COATI_LAUNCH_CHECK("mlp_permute_w1");
  return COATI_OK;
}

template <bool FWD>
__global__ __launch_bounds__(64 * (MP_CW + MP_PW), 1) void mlp_pipe_kernel(MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* const Ws = reinterpret_cast<bf16_t*>(smem);                                   // [MLP_RING][stage]
  float* const B1s = reinterpret_cast<float*>(smem + MLP_RING * MLP_STAGE_HALFS * 2);   // [Hd] (forward)
  float* const B2s = B1s + p.Hd;                                                        // [C]  (forward)
  float* const GBs = B2s + MLP_C;                                                       // [2][C] LayerNorm gamma | beta (forward)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nst = p.Hd / MLP_CH;
  const int PERIOD = MP_DP + 1 + nst;            // issue | slack | LayerNorm / reset | nst chunk stages (the stores ride on the last one)
  const int OFF1 = PERIOD / 2;
  const int T_total = OFF1 + 2 * PERIOD;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;

  if (wave >= MP_CW) {
    // ---------------------------------------------- producers -----------------------------------------------------------
    const int pw = wave - MP_CW;
    int off1[8], off2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 8 * pw + i;        // piece index 0..15 of each of the two weight chunks
      const int r1 = 2 * k + (lane >> 5), pos1 = lane & 31, f = ((r1 >> 1) & 12) | (r1 & 3);
      off1[i] = r1 * (int)p.ldw1 + ((pos1 ^ f) * 8);
      const int r2 = 16 * k + (lane >> 2), pos2 = lane & 3, hh = (4 - ((r2 >> 2) & 3)) & 3;
      off2[i] = r2 * (int)p.ldw2 + ((pos2 ^ hh) * 8);
    }
    auto issue = [&](int st) {
      const int h0 = (st % nst) * MLP_CH;
      bf16_t* S = Ws + (st % MLP_RING) * MLP_STAGE_HALFS;
      const bf16_t* b1p = p.W1 + (long long)h0 * p.ldw1;
      const bf16_t* b2p = p.W2 + h0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 8 * pw + i;
        __builtin_amdgcn_global_load_lds((gbl_void*)(b1p + off1[i]), (lds_void*)(S + k * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(b2p + off2[i]), (lds_void*)(S + MLP_W1_HALFS + k * 512), 16, 0, 0);
      }
    };
#pragma unroll
    for (int j = 0; j < MLP_RING - 1; ++j) issue(j);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    for (int st = 0; st < T_total; ++st) {
      issue(st + MLP_RING - 1);            // into the slot every consumer finished reading before the previous barrier
      // stage st + 1 has landed when at most the 2 x 16 DMAs of stages st + 2, st + 3 are still in flight: vmcnt(32)
      __builtin_amdgcn_s_waitcnt(0x8f70);  // vmcnt(32): bits [15:14] = 2, [3:0] = 0
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    return;
  }

  // ------------------------------------------------ consumers -------------------------------------------------------------
  const int t = lane & 15, q = lane >> 4;
  const int grp = wave / MP_GW, gw = wave - grp * MP_GW;
  if constexpr (FWD) {
    for (int i = tid; i < p.Hd; i += 64 * MP_CW) B1s[i] = p.b1 ? p.b1[i] : 0.f;
    if (tid < MLP_C) {
      B2s[tid] = p.b2 ? p.b2[tid] : 0.f;
      GBs[tid] = p.gamma[tid];
      GBs[MLP_C + tid] = p.beta[tid];
    }
  }
  f32x4 acc2[16];     // forward: holds the fp32 row x between its load and the LayerNorm, then x + b2 + the second product
  bf16x8 af[8];
#pragma unroll
  for (int ct = 0; ct < 16; ++ct) acc2[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) af[s] = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
  const int w1_row0 = (8 * (t >> 2) + (t & 3)) * (MLP_C * 2);
  const int w2_off = t * (MLP_CH * 2) + ((q ^ ((4 - (t >> 2)) & 3)) << 4);
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the bias vectors are in LDS
  __builtin_amdgcn_s_barrier();         // (matches the producers' prologue barrier)

  // The schedule as straight-line phases (one s_barrier = one stage; every wave of the workgroup executes T_total of them):
  //   group 0: [block 0][block 1][OFF1 idle stages]      group 1: [OFF1 idle stages][block 0][block 1]
  //   block  : issue the row loads | MP_DP - 1 slack stages | LayerNorm (fwd) / reset (bwd) | nst chunk stages (+ the stores)
  int st = 0;
  auto next_stage = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's reads of the ring slot are complete before it is refilled
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");        // the ring changes underneath: no LDS value is carried (or hoisted) across a stage
    ++st;
  };
  for (int i = 0; i < (grp ? OFF1 : 0); ++i) next_stage();
  for (int blk = 0; blk < 2; ++blk) {
    // ---- a new block of 80 rows: issue the row loads
    const int m0 = (int)blockIdx.x * MP_ROWS_WG + (2 * blk + grp) * (MP_GW * 16) + gw * 16;
    const int row = m0 + t;
    const bool rowok = row < p.M;
    const int rc = rowok ? row : p.M - 1;
    if constexpr (FWD) {
      const float* xp = p.x + (long long)rc * p.ldx + 4 * q;
#pragma unroll
      for (int ct = 0; ct < 16; ++ct) acc2[ct] = *reinterpret_cast<const f32x4*>(xp + 16 * ct);
    } else {
      const bf16_t* ap = p.a + (long long)rc * p.lda + 8 * q;
#pragma unroll
      for (int s = 0; s < 8; ++s) af[s] = *reinterpret_cast<const bf16x8*>(ap + 32 * s);
    }
    for (int i = 0; i < MP_DP; ++i) next_stage();   // the loads land underneath the other group's chunk stages
    uint2 aux_a = make_uint2(0, 0), aux_b = make_uint2(0, 0);   // backward: the saved derivative codes of the next two stages
    if constexpr (FWD) {
      // ---- LayerNorm on the row in the accumulator registers (4 lanes (t, q = 0..3) hold one row); two-pass statistics
      float sm = 0.f;
#pragma unroll
      for (int ct = 0; ct < 16; ++ct) sm += acc2[ct][0] + acc2[ct][1] + acc2[ct][2] + acc2[ct][3];
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm / (float)MLP_C;
      float q2 = 0.f;
#pragma unroll
      for (int ct = 0; ct < 16; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dd = acc2[ct][r] - mean;
          q2 += dd * dd;
        }
      q2 += __shfl_xor(q2, 16, 64);
      q2 += __shfl_xor(q2, 32, 64);
      const float rstd = 1.0f / sqrtf(q2 / (float)MLP_C + 1e-5f);
      if (q == 0 && rowok) {
        p.mean[row] = mean;
        p.rstd[row] = rstd;
      }
      // normalised row: k-step s of the first product takes column tiles 2s, 2s+1 (slot i <-> tile 2s + (i >> 2), r = i & 3)
      bf16_t* op = p.a + (long long)rc * p.lda;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        float o[8];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int ct = 2 * s + h2;
          const float4 g4 = *reinterpret_cast<const float4*>(GBs + 16 * ct + 4 * q);
          const float4 b4 = *reinterpret_cast<const float4*>(GBs + MLP_C + 16 * ct + 4 * q);
          o[4 * h2 + 0] = (acc2[ct][0] - mean) * rstd * g4.x + b4.x;
          o[4 * h2 + 1] = (acc2[ct][1] - mean) * rstd * g4.y + b4.y;
          o[4 * h2 + 2] = (acc2[ct][2] - mean) * rstd * g4.z + b4.z;
          o[4 * h2 + 3] = (acc2[ct][3] - mean) * rstd * g4.w + b4.w;
        }
        const uint4 u4 = pack8(o);
        af[s] = __builtin_bit_cast(bf16x8, u4);
        // the saved copy in natural column order: columns 32 s + 4 q .. + 3 (u4.x, u4.y) and 32 s + 16 + 4 q .. + 3 (u4.z, u4.w)
        if (rowok) {
          *reinterpret_cast<uint2*>(op + 32 * s + 4 * q) = make_uint2(u4.x, u4.y);
          *reinterpret_cast<uint2*>(op + 32 * s + 16 + 4 * q) = make_uint2(u4.z, u4.w);
        }
        __builtin_amdgcn_sched_barrier(0);   // one k-step at a time
      }
      // the accumulator starts from x + b2: the residual and the second bias cost no extra pass
#pragma unroll
      for (int ct = 0; ct < 16; ++ct) {
        const float4 b = *reinterpret_cast<const float4*>(B2s + 16 * ct + 4 * q);
        acc2[ct][0] += b.x; acc2[ct][1] += b.y; acc2[ct][2] += b.z; acc2[ct][3] += b.w;
      }
    } else {
#pragma unroll
      for (int ct = 0; ct < 16; ++ct) acc2[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      // the saved derivative codes of the first two chunk stages (chunk ids st + 1, st + 2)
      const unsigned char* ax = reinterpret_cast<const unsigned char*>(p.aux) + (long long)rc * p.ldh + 8 * q;
      aux_a = *reinterpret_cast<const uint2*>(ax + ((st + 1) % nst) * MLP_CH);
      aux_b = *reinterpret_cast<const uint2*>(ax + ((st + 2) % nst) * MLP_CH);
    }
    next_stage();
    for (int c = 0; c < nst; ++c) {
      // ---- one chunk stage: hidden units h0 .. h0 + 31 (whatever the ring holds at this stage)
      const int h0 = (st % nst) * MLP_CH;
      const bf16_t* cur = Ws + (st % MLP_RING) * MLP_STAGE_HALFS;
      f32x4 a0, a1;
      if constexpr (FWD) {
        a0 = *reinterpret_cast<const f32x4*>(B1s + h0 + 8 * q);
        a1 = *reinterpret_cast<const f32x4*>(B1s + h0 + 8 * q + 4);
      } else {
        a0 = f32x4{0.f, 0.f, 0.f, 0.f};
        a1 = a0;
      }
      {
        const unsigned char* w1 = reinterpret_cast<const unsigned char*>(cur) + w1_row0;
        bf16x8 wf0[2], wf1[2];
        // 16-byte piece (4 s + q) ^ t of the row = (sw ^ ((s & 3) << 6)) + ((s >> 2) << 8): one register + one xor per k-step
        unsigned ln = threadIdx.x;
        asm volatile("" : "+v"(ln));   // (recomputed per stage: hoisted out of the chunk loop these offsets get spilled)
        const unsigned sw = ((((ln >> 4) ^ ln) & 3u) << 4) | (((ln >> 2) & 3u) << 6);
        wf0[0] = *reinterpret_cast<const bf16x8*>(w1 + sw);
        wf1[0] = *reinterpret_cast<const bf16x8*>(w1 + 4 * MLP_C * 2 + sw);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          if (s + 1 < 8) {
            const unsigned o1 = (sw ^ (unsigned)(((s + 1) & 3) << 6)) + (unsigned)(((s + 1) >> 2) << 8);
            wf0[(s + 1) & 1] = *reinterpret_cast<const bf16x8*>(w1 + o1);
            wf1[(s + 1) & 1] = *reinterpret_cast<const bf16x8*>(w1 + 4 * MLP_C * 2 + o1);
          }
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[s & 1], af[s], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[s & 1], af[s], a1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      float o[8];
      if constexpr (FWD) {
        float d[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          coati_v2f hh, dd;
          gelu_and_grad_f2(coati_v2f{v[e], v[e + 1]}, hh, dd);
          o[e] = hh.x; o[e + 1] = hh.y;
          d[e] = dd.x; d[e + 1] = dd.y;
        }
        if (rowok) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.d) + (long long)row * p.ldh + h0 + 8 * q) = packq8(d);
      } else {
        float x[8];
        unpackq8(aux_a, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e] * x[e];
        aux_a = aux_b;
        aux_b = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(p.aux) + (long long)rc * p.ldh + ((st + 2) % nst) * MLP_CH + 8 * q);
      }
      const uint4 hv = pack8(o);
      if (rowok) *reinterpret_cast<uint4*>(p.h + (long long)row * p.ldh + h0 + 8 * q) = hv;
      const bf16x8 gf = __builtin_bit_cast(bf16x8, hv);
      {
        const unsigned char* w2 = reinterpret_cast<const unsigned char*>(cur + MLP_W1_HALFS) + w2_off;
        bf16x8 wf[3];
        wf[0] = *reinterpret_cast<const bf16x8*>(w2);
        wf[1] = *reinterpret_cast<const bf16x8*>(w2 + 16 * MLP_CH * 2);
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
          if (ct + 2 < 16) wf[(ct + 2) % 3] = *reinterpret_cast<const bf16x8*>(w2 + (ct + 2) * 16 * MLP_CH * 2);
          acc2[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct % 3], gf, acc2[ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (c == nst - 1 && rowok) {
        // ---- the block is complete: store its rows (lane (t, q) holds columns 16 ct + 4 q + 0..3 of row t)
        if constexpr (FWD) {
          float* outp = reinterpret_cast<float*>(p.out) + (long long)row * p.ldo + 4 * q;
#pragma unroll
          for (int ct = 0; ct < 16; ++ct) *reinterpret_cast<f32x4*>(outp + 16 * ct) = acc2[ct];
        } else {
          bf16_t* outp = reinterpret_cast<bf16_t*>(p.out) + (long long)row * p.ldo + 4 * q;
#pragma unroll
          for (int ct = 0; ct < 16; ++ct)
            *reinterpret_cast<uint2*>(outp + 16 * ct) = make_uint2(pack2bf(acc2[ct][0], acc2[ct][1]), pack2bf(acc2[ct][2], acc2[ct][3]));
        }
      }
      next_stage();
    }
  }
  for (int i = 0; i < (grp ? 0 : OFF1); ++i) next_stage();
}

bool mlp_chain_supported(const MlpArgs& a) {
  // C == 256 (one accumulator row = 16 column tiles); hidden width a whole number of 32-unit chunks that the ring's
  // schedule covers (PERIOD = MP_DP + 1 + Hd / 32 stages); rows 16-byte aligned
  return a.C == MLP_C && a.Hd % MLP_CH == 0 && a.Hd >= MLP_CH && a.Hd <= 4096 && a.M > 0 && a.lda % 8 == 0 && a.ldh % 8 == 0 &&
         a.ldx % 4 == 0 && a.ldo % 4 == 0;
}

template <bool FWD>
static int launch_mlp_pipe(const MlpArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = mlp_pipe_kernel<FWD>;
  const size_t lds = (size_t)MLP_RING * MLP_STAGE_HALFS * 2 + (FWD ? (size_t)(a.Hd + 3 * MLP_C) * 4 : 0);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((size_t)MLP_RING * MLP_STAGE_HALFS * 2 + (size_t)(4096 + 3 * MLP_C) * 4));
    if (e != hipSuccess) {
      coati_set_error("mlp_pipe: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(a.M, MP_ROWS_WG)), dim3(64 * (MP_CW + MP_PW)), lds, s, a);
  COATI_LAUNCH_CHECK("mlp_pipe");
  return COATI_OK;
}

// forward: W1 is the COLUMN-PERMUTED copy of the fc1 weight (launch_mlp_permute_w1); d receives NewGELU' as 8-bit codes
int launch_mlp_fwd(const MlpArgs& a, hipStream_t s) {
  COATI_CHECK_ARG(a.x && a.gamma && a.beta && a.mean && a.rstd && a.a && a.W1 && a.W2 && a.h && a.d && a.out, "mlp_fwd: null operand");
  COATI_CHECK_SHAPE(mlp_chain_supported(a), "mlp_fwd: unsupported shape C=%d Hd=%d", a.C, a.Hd);
  COATI_CHECK_SHAPE(a.ldx % 4 == 0 && a.lda % 8 == 0 && a.ldh % 8 == 0 && a.ldo % 4 == 0 && a.ldw1 % 8 == 0 && a.ldw2 % 8 == 0, "mlp_fwd: alignment");
  return launch_mlp_pipe<true>(a, s);
}
// backward: aux = the saved 8-bit NewGELU' codes
int launch_mlp_bwd(const MlpArgs& a, hipStream_t s) {
  COATI_CHECK_ARG(a.a && a.W1 && a.W2 && a.h && a.aux && a.out, "mlp_bwd: null operand");
  COATI_CHECK_SHAPE(mlp_chain_supported(a), "mlp_bwd: unsupported shape C=%d Hd=%d", a.C, a.Hd);
  COATI_CHECK_SHAPE(a.lda % 8 == 0 && a.ldh % 8 == 0 && a.ldo % 8 == 0 && a.ldw1 % 8 == 0 && a.ldw2 % 8 == 0, "mlp_bwd: alignment");
  return launch_mlp_pipe<false>(a, s);
}
