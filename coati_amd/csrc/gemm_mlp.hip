// Chained two-GEMM kernel for the transformer MLP at C = 256 (reference basic_transformer.py:157-174, RotaryBlock.mlpf:
// x + W2 NewGELU(W1 ln_2(x) + b1) + b2) and for the matching input-gradient chain of the backward.
//
//   forward  (FWD):  a = LayerNorm(x)              (bf16, saved: the weight gradient of W1 reads it)
//                    pre = a W1^T + b1             (never leaves the chip)
//                    g = NewGELU(pre), d = NewGELU'(pre)   (bf16, saved: W2's weight gradient reads g, the backward multiplies by d)
//                    out = x + g W2^T + b2         (f32 residual stream)
//   backward (!FWD): dg = dY W1^T                  (W1 := fc2 weight transposed, [Hd, C])
//                    dh = dg * d                   (bf16, saved: both MLP weight gradients read it)
//                    out = dh W2^T                 (W2 := fc1 weight transposed, [C, Hd]; bf16: the LayerNorm backward reads it)
//
// Versus the two separate GEMM launches this removes one trip of the [M, 4C] intermediate through HBM per direction
// (forward: g is not re-read, x is read once instead of twice; backward: dh is not re-read): 3 KB of the 9.7 KB per token
// and layer the forward MLP moved, 2 KB of the 7.2 KB of the two input-gradient products.
//
// Shape of the kernel.  One workgroup of 10 waves owns 160 rows (81,920 rows = 512 workgroups = 2 per CU); a wave owns 16
// rows for the whole kernel.  Both products are computed TRANSPOSED with v_mfma_f32_16x16x32_bf16 so that the first
// product's accumulator layout IS the second product's B-operand layout and the activation runs in registers:
//   GEMM1^T: pre^T[h, t] = sum_k W1[h, k] a[t, k]   A-operand = weight rows (LDS), B-operand = the wave's 16 token rows, resident
//            in registers for the whole kernel (8 k-steps x bf16x8);  lane (t = lane & 15, q = lane >> 4) ends up with
//            rows 4q .. 4q+3 of each 16-row output tile.
//   The two tiles of a 32-unit chunk are given the hidden units {8q' + r} and {8q' + 4 + r} (q' = 0..3, r = 0..3) as their
//   rows 4q' + r -- a free choice of which weight row a lane reads -- so lane (t, q) holds the 8 CONSECUTIVE hidden units
//   8q .. 8q+7 of token t: exactly the k-slots of the B operand of
//   GEMM2^T: out^T[c, t] += sum_h W2[c, h] g[t, h]   A-operand = W2 rows (LDS), one k-step of 32 per chunk, 16 column tiles.
// The same 8 values are 16 contiguous bytes of the saved [M, Hd] tensors.
// Weights stream HBM/L2 -> LDS with global_load_lds_dwordx4 in 32-hidden-unit chunks (W1 chunk [32][256] + W2 chunk
// [256][32] = 32 KiB) through a 4-slot ring: the DMAs of chunk j + 3 are issued in chunk j (a chunk is ~1 us of work, an L2
// round trip under load is longer: with a prefetch distance of one chunk the kernel spent 2/3 of its time waiting at the
// barrier).  One barrier per chunk; vmcnt completes in order, so "the next chunk's weights have landed" is
// s_waitcnt vmcnt(everything this wave issued after them) -- every wave issues the same number of memory operations per
// chunk (4 DMA slots, the unused ones skipped by count).  16-byte pieces are XOR-swizzled on the global side so that every
// fragment ds_read_b128 is conflict-free (W1: position = chunk ^ tile-row; W2: position = q ^ ((-row >> 2) & 3)).
// The saved tensors of chunk j are stored at the start of chunk j + 1 (behind the barrier, so that they drain underneath it).
#include <cstdlib>
#include "gemm_epi.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MLP_C 256
#define MLP_CH 32                              // hidden units per stage
#define MLP_W 10                               // waves per workgroup
#define MLP_ROWS (16 * MLP_W)
#define MLP_W1_HALFS (MLP_CH * MLP_C)          // [32][256]
#define MLP_W2_HALFS (MLP_C * MLP_CH)          // [256][32]
#define MLP_STAGE_HALFS (MLP_W1_HALFS + MLP_W2_HALFS)
#define MLP_RING 4                             // ring slots (prefetch distance MLP_RING - 1 chunks)
#define MLP_XRING 3                            // backward: ring of the saved-derivative pieces (prefetch distance 2)

template <typename F>
__device__ __forceinline__ void mlp_call_restrict(F&& f, int j, const bf16_t* __restrict__ cur, bf16_t* __restrict__ nxt) {
  f(j, cur, nxt);
}

// ABL: timing ablations (COATI_MLP_ABL, tools/mlp_bench.py): 1 no activation maths, 2 no saved-tensor stores, 4 no GEMM2,
// 8 no GEMM1, 16 no weight DMA, 32 no per-chunk wait + barrier.  0 in the product.
template <bool FWD, int ABL = 0>
__global__ __launch_bounds__(64 * MLP_W, 1) void mlp_chain_kernel(MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* const Ws = reinterpret_cast<bf16_t*>(smem);                                   // [MLP_RING][stage]
  float* const B1s = reinterpret_cast<float*>(smem + MLP_RING * MLP_STAGE_HALFS * 2);   // [Hd] (forward)
  // backward: the saved NewGELU' pieces of this wave, one KiB per chunk in lane order, 3-slot ring (prefetch distance 2)
  unsigned char* const Xs = smem + MLP_RING * MLP_STAGE_HALFS * 2;                      // [MLP_XRING][MLP_W][1 KiB] (backward)
  float* const B2s = B1s + p.Hd;                                                        // [C]  (forward)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform in an SGPR: scalar branches, scalar addresses
  const int t = lane & 15, q = lane >> 4;
  const int m0 = (blockIdx.x * MLP_W + wave) * 16;
  const int row = m0 + t, rc = row < p.M ? row : p.M - 1;
  const bool rowok = row < p.M;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;

  // ---- weight chunk DMA: 32 one-KiB wave instructions per chunk (16 for W1: two 512-B rows each; 16 for W2: sixteen 64-B
  // rows each).  Waves 0..7 issue four each (W1 pieces 2w, 2w+1; W2 pieces 2w, 2w+1); waves 8, 9 none.  Per-lane element
  // offsets are loop-invariant; the chunk base is wave-uniform.
  const bool dma_wave = wave < 8;
  int off1[2], off2[2], lds1[2], lds2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = (2 * wave + i) & 15;
    const int r1 = 2 * k + (lane >> 5), pos1 = lane & 31, f = ((r1 >> 1) & 12) | (r1 & 3);
    off1[i] = r1 * (int)p.ldw1 + ((pos1 ^ f) * 8);
    lds1[i] = k * 512;
    const int r2 = 16 * k + (lane >> 2), pos2 = lane & 3, hh = (4 - ((r2 >> 2) & 3)) & 3;
    off2[i] = r2 * (int)p.ldw2 + ((pos2 ^ hh) * 8);
    lds2[i] = MLP_W1_HALFS + k * 512;
  }
  auto load_stage = [&](int h0, bf16_t* S) {
    if (dma_wave && !(ABL & 16)) {
      const bf16_t* b1p = p.W1 + (long long)h0 * p.ldw1;
      const bf16_t* b2p = p.W2 + h0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_void*)(b1p + off1[i]), (lds_void*)(S + lds1[i]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(b2p + off2[i]), (lds_void*)(S + lds2[i]), 16, 0, 0);
      }
    }
  };
  const int nst = p.Hd / MLP_CH;
#pragma unroll
  for (int j = 0; j < MLP_RING - 1; ++j) load_stage((j < nst ? j : nst - 1) * MLP_CH, Ws + j * MLP_STAGE_HALFS);

  // ---- resident B operand of GEMM1: the wave's 16 rows, lane (t, q) holds k = 32 s + 8 q .. + 7 for s = 0..7
  bf16x8 af[8];
  if constexpr (FWD) {
    for (int i = tid; i < p.Hd; i += 64 * MLP_W) B1s[i] = p.b1 ? p.b1[i] : 0.f;
    if (tid < MLP_C) B2s[tid] = p.b2 ? p.b2[tid] : 0.f;
    // LayerNorm fused into the slab load: the 4 lanes (t, q = 0..3) hold one f32 row of 256; two-pass statistics
    const float* xp = p.x + (long long)rc * p.ldx + 8 * q;
    float xf[8][8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float4 x0 = *reinterpret_cast<const float4*>(xp + 32 * s), x1 = *reinterpret_cast<const float4*>(xp + 32 * s + 4);
      xf[s][0] = x0.x; xf[s][1] = x0.y; xf[s][2] = x0.z; xf[s][3] = x0.w;
      xf[s][4] = x1.x; xf[s][5] = x1.y; xf[s][6] = x1.z; xf[s][7] = x1.w;
    }
    float sm = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) sm += xf[s][i];
    sm += __shfl_xor(sm, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    const float mean = sm / (float)MLP_C;
    float q2 = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float dd = xf[s][i] - mean;
        q2 += dd * dd;
      }
    q2 += __shfl_xor(q2, 16, 64);
    q2 += __shfl_xor(q2, 32, 64);
    const float rstd = 1.0f / sqrtf(q2 / (float)MLP_C + 1e-5f);
    if (q == 0 && rowok) {
      p.mean[row] = mean;
      p.rstd[row] = rstd;
    }
    bf16_t* op = p.a + (long long)rc * p.lda + 8 * q;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float* gp = p.gamma + 32 * s + 8 * q;
      const float* bp = p.beta + 32 * s + 8 * q;
      const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
      const float4 c0 = *reinterpret_cast<const float4*>(bp), c1 = *reinterpret_cast<const float4*>(bp + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (xf[s][i] - mean) * rstd * g[i] + bt[i];
      const uint4 u = pack8(o);
      af[s] = __builtin_bit_cast(bf16x8, u);
      if (rowok) *reinterpret_cast<uint4*>(op + 32 * s) = u;
    }
  } else {
    const bf16_t* ap = p.a + (long long)rc * p.lda + 8 * q;
#pragma unroll
    for (int s = 0; s < 8; ++s) af[s] = *reinterpret_cast<const bf16x8*>(ap + 32 * s);
  }

  // ---- accumulator of GEMM2^T: 16 column tiles, lane (t, q) holds columns 16 ct + 4 q + r of row t
  f32x4 acc2[16];
#pragma unroll
  for (int ct = 0; ct < 16; ++ct) acc2[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS byte offsets of this lane's fragment reads
  const int w1_row0 = (8 * (t >> 2) + (t & 3)) * (MLP_C * 2);            // tile 0 row of lane t; tile 1 is 4 rows further
  const int w2_off = t * (MLP_CH * 2) + ((q ^ ((4 - (t >> 2)) & 3)) << 4);
  // deferred stores of the previous chunk's saved tensors
  uint4 sv_h = make_uint4(0, 0, 0, 0), sv_d = make_uint4(0, 0, 0, 0);
  const bf16_t* const auxrow = FWD ? nullptr : p.aux + (long long)rc * p.ldh + 8 * q;
  auto load_aux = [&](int jc) {   // chunk jc's 16 B of this lane -> ring slot jc % MLP_XRING (clamped past the end)
    const int jj = jc < nst ? jc : nst - 1;
    __builtin_amdgcn_global_load_lds((gbl_void*)(auxrow + jj * MLP_CH), (lds_void*)(Xs + ((jc % MLP_XRING) * MLP_W + wave) * 1024), 16, 0, 0);
  };
  if constexpr (!FWD) {
    load_aux(0);
    load_aux(1);
  }
  // rows past M: this wave issues no stores, so its in-order operation count per chunk is lower (wave-uniform)
  const bool tail_wave = m0 >= p.M;
  bf16_t* const hrow = p.h + (long long)rc * p.ldh + 8 * q;
  bf16_t* const drow = FWD ? p.d + (long long)rc * p.ldh + 8 * q : nullptr;

  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): chunks 0 .. MLP_RING-2 and the resident rows
  __syncthreads();

  auto stage = [&](int j, const bf16_t* cur, bf16_t* nxt) {
    const int h0 = j * MLP_CH;
    // stores of chunk j-1 (their values were produced before the barrier that ended it: they drain underneath this chunk)
    if (j > 0 && rowok && !(ABL & 2)) {
      *reinterpret_cast<uint4*>(hrow + h0 - MLP_CH) = sv_h;
      if constexpr (FWD) *reinterpret_cast<uint4*>(drow + h0 - MLP_CH) = sv_d;
    }
    {   // chunk j + MLP_RING - 1 (clamped past the end: an unused re-read keeps the number of DMAs per chunk uniform)
      const int jn = j + MLP_RING - 1;
      load_stage((jn < nst ? jn : nst - 1) * MLP_CH, nxt);
    }
    if constexpr (!FWD) load_aux(j + 2);
    // ---- GEMM1^T: two 16-row tiles over k = 256
    f32x4 a0, a1;
    if constexpr (FWD) {
      a0 = *reinterpret_cast<const f32x4*>(B1s + h0 + 8 * q);
      a1 = *reinterpret_cast<const f32x4*>(B1s + h0 + 8 * q + 4);
    } else {
      a0 = f32x4{0.f, 0.f, 0.f, 0.f};
      a1 = a0;
    }
    if constexpr (!(ABL & 8)) {
      const unsigned char* w1 = reinterpret_cast<const unsigned char*>(cur) + w1_row0;
      bf16x8 wf0[2], wf1[2];
      wf0[0] = *reinterpret_cast<const bf16x8*>(w1 + (((0 + q) ^ t) << 4));
      wf1[0] = *reinterpret_cast<const bf16x8*>(w1 + 4 * MLP_C * 2 + (((0 + q) ^ t) << 4));
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) {
          wf0[(s + 1) & 1] = *reinterpret_cast<const bf16x8*>(w1 + (((4 * (s + 1) + q) ^ t) << 4));
          wf1[(s + 1) & 1] = *reinterpret_cast<const bf16x8*>(w1 + 4 * MLP_C * 2 + (((4 * (s + 1) + q) ^ t) << 4));
        }
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[s & 1], af[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[s & 1], af[s], a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads one step ahead of the MFMAs (the scheduler sinks them)
      }
    }
    // ---- activation in registers: lane (t, q) holds hidden units h0 + 8 q .. + 7 of row t
    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    float o[8];
    if constexpr (ABL & 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[e];
      sv_d = pack8(v);
    } else if constexpr (FWD) {
      float d[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        coati_v2f hh, dd;
        gelu_and_grad_f2(coati_v2f{v[e], v[e + 1]}, hh, dd);
        o[e] = hh.x; o[e + 1] = hh.y;
        d[e] = dd.x; d[e + 1] = dd.y;
      }
      sv_d = pack8(d);
    } else {
      float x[8];
      unpack8(*reinterpret_cast<const uint4*>(Xs + ((j % MLP_XRING) * MLP_W + wave) * 1024 + lane * 16), x);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[e] * x[e];
    }
    sv_h = pack8(o);
    const bf16x8 gf = __builtin_bit_cast(bf16x8, sv_h);
    // ---- GEMM2^T: 16 column tiles, one k-step of 32
    if constexpr (ABL & 4) {
      acc2[j & 15][0] += o[0] + o[5];
    } else {
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
    // Chunk j + 1's weights (and, backward, its saved-derivative piece) must have landed.  vmcnt completes in issue order and
    // every chunk issues [stores of chunk j-1] [4 weight DMAs (waves 0..7)] [backward: 1 piece DMA], so the operations this
    // wave issued AFTER the ones it needs are known: forward 2 x (2 + 4) = 12 after W(j+1); backward 1 + 4 + 1 = 6 after
    // piece(j+1) (W(j+1) is older still).  The stores of the last two chunks stay in flight across the barrier.  A wave with
    // rows past M may have issued fewer stores: it drains everything.
    if constexpr (!(ABL & 32)) {
      if (tail_wave) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
      } else if constexpr (FWD) {
        if (dma_wave) __builtin_amdgcn_s_waitcnt(0x0f7c);   // vmcnt(12)
      } else {
        if (dma_wave) __builtin_amdgcn_s_waitcnt(0x0f76);   // vmcnt(6)
        else __builtin_amdgcn_s_waitcnt(0x0f72);            // vmcnt(2): store, piece(j+2)
      }
      __builtin_amdgcn_s_barrier();   // a plain barrier: __syncthreads() also fences, i.e. drains every DMA and store in flight
    }
  };
  for (int j = 0; j < nst; ++j)
    mlp_call_restrict(stage, j, Ws + (j % MLP_RING) * MLP_STAGE_HALFS, Ws + ((j + MLP_RING - 1) % MLP_RING) * MLP_STAGE_HALFS);
  __builtin_amdgcn_s_waitcnt(0x0f70);   // the clamped tail DMAs must not land in the region the epilogue reuses
  __syncthreads();

  if (rowok) {
    *reinterpret_cast<uint4*>(hrow + (nst - 1) * MLP_CH) = sv_h;
    if constexpr (FWD) *reinterpret_cast<uint4*>(drow + (nst - 1) * MLP_CH) = sv_d;
  }
  // ---- output: lane (t, q) holds out[row t, 16 ct + 4 q + 0..3]
  if constexpr (FWD) {
    if (rowok) {
      float* outp = reinterpret_cast<float*>(p.out) + (long long)row * p.ldo + 4 * q;
      const float* resp = p.x + (long long)row * p.ldx + 4 * q;
#pragma unroll
      for (int ct = 0; ct < 16; ++ct) {
        const float4 r = *reinterpret_cast<const float4*>(resp + 16 * ct);
        const float4 b = *reinterpret_cast<const float4*>(B2s + 16 * ct + 4 * q);
        float4 o4;
        o4.x = acc2[ct][0] + r.x + b.x; o4.y = acc2[ct][1] + r.y + b.y; o4.z = acc2[ct][2] + r.z + b.z; o4.w = acc2[ct][3] + r.w + b.w;
        *reinterpret_cast<float4*>(outp + 16 * ct) = o4;
      }
    }
  } else {
    // bf16 rows through a wave-private LDS transpose (the weight buffers are idle: every wave passed the last barrier):
    // 16 rows x 512 B per wave -> each lane then stores 2 x 16 B pieces of a row, 64 lanes = 4 rows x 128 B x 2
    unsigned char* Ts = smem + wave * (16 * 528);      // rows padded to 528 B
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) {
      const uint2 u = make_uint2(pack2bf(acc2[ct][0], acc2[ct][1]), pack2bf(acc2[ct][2], acc2[ct][3]));
      *reinterpret_cast<uint2*>(Ts + t * 528 + (16 * ct + 4 * q) * 2) = u;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int task = lane + 64 * i, r = task >> 5, cg = task & 31;     // 16 rows x 32 sixteen-byte pieces
      if (m0 + r < p.M) *reinterpret_cast<uint4*>(outp + (long long)(m0 + r) * p.ldo + cg * 8) = *reinterpret_cast<const uint4*>(Ts + r * 528 + cg * 16);
    }
  }
}

bool mlp_chain_supported(const MlpArgs& a) {
  static const bool off = getenv("COATI_NO_MLP_CHAIN") != nullptr;      // A/B switch
  return !off && a.C == MLP_C && a.Hd % MLP_CH == 0 && a.Hd >= MLP_CH && a.Hd <= 4096 && a.M > 0;
}

template <bool FWD, int ABL>
static int launch_mlp_abl(const MlpArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = mlp_chain_kernel<FWD, ABL>;
  // weight ring + (forward) the two bias vectors; the backward's output transpose (10 x 16 rows x 528 B) reuses the ring
  // after the last barrier
  constexpr size_t kTranspose = (size_t)MLP_W * 16 * 528;
  static_assert(kTranspose <= (size_t)MLP_RING * MLP_STAGE_HALFS * 2, "the backward's output transpose lives in the ring");
  const size_t lds = (size_t)MLP_RING * MLP_STAGE_HALFS * 2 + (FWD ? (size_t)(a.Hd + MLP_C) * 4 : (size_t)MLP_XRING * MLP_W * 1024);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((size_t)MLP_RING * MLP_STAGE_HALFS * 2 + (FWD ? (size_t)(4096 + MLP_C) * 4 : (size_t)MLP_XRING * MLP_W * 1024)));
    if (e != hipSuccess) {
      coati_set_error("mlp_chain: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(a.M, MLP_ROWS)), dim3(64 * MLP_W), lds, s, a);
  COATI_LAUNCH_CHECK("mlp_chain");
  return COATI_OK;
}

template <bool FWD>
static int launch_mlp_t(const MlpArgs& a, hipStream_t s) {
  static const int abl = getenv("COATI_MLP_ABL") ? atoi(getenv("COATI_MLP_ABL")) : 0;
  switch (abl) {
    case 1: return launch_mlp_abl<FWD, 1>(a, s);
    case 2: return launch_mlp_abl<FWD, 2>(a, s);
    case 4: return launch_mlp_abl<FWD, 4>(a, s);
    case 8: return launch_mlp_abl<FWD, 8>(a, s);
    case 16: return launch_mlp_abl<FWD, 16>(a, s);
    case 32: return launch_mlp_abl<FWD, 32>(a, s);
    case 48: return launch_mlp_abl<FWD, 48>(a, s);
    case 15: return launch_mlp_abl<FWD, 15>(a, s);
    default: return launch_mlp_abl<FWD, 0>(a, s);
  }
}

int launch_mlp_fwd(const MlpArgs& a, hipStream_t s) {
  COATI_CHECK_ARG(a.x && a.gamma && a.beta && a.mean && a.rstd && a.a && a.W1 && a.W2 && a.h && a.d && a.out, "mlp_fwd: null operand");
  COATI_CHECK_SHAPE(mlp_chain_supported(a), "mlp_fwd: unsupported shape C=%d Hd=%d", a.C, a.Hd);
  COATI_CHECK_SHAPE(a.ldx % 4 == 0 && a.lda % 8 == 0 && a.ldh % 8 == 0 && a.ldo % 4 == 0 && a.ldw1 % 8 == 0 && a.ldw2 % 8 == 0, "mlp_fwd: alignment");
  return launch_mlp_t<true>(a, s);
}
int launch_mlp_bwd(const MlpArgs& a, hipStream_t s) {
  COATI_CHECK_ARG(a.a && a.W1 && a.W2 && a.h && a.aux && a.out, "mlp_bwd: null operand");
  COATI_CHECK_SHAPE(mlp_chain_supported(a), "mlp_bwd: unsupported shape C=%d Hd=%d", a.C, a.Hd);
  COATI_CHECK_SHAPE(a.lda % 8 == 0 && a.ldh % 8 == 0 && a.ldo % 8 == 0 && a.ldw1 % 8 == 0 && a.ldw2 % 8 == 0, "mlp_bwd: alignment");
  return launch_mlp_t<false>(a, s);
}
