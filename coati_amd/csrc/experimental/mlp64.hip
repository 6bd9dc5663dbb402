// The MLP half of a transformer block as ONE launch (round 5): out = x + c_proj(NewGELU(c_fc(ln_2(x)))) (basic_transformer.py:157-174,
// forward), d = 256, hidden 1024, at packed-batch sizes.  The hidden activation g never comes BACK from memory: it is written once
// (bf16, with the 8-bit NewGELU' codes: the backward reads both) and feeds the second product from the registers it was computed in.
//
// Both products are issued TRANSPOSED on v_mfma_f32_32x32x16_bf16 (weight rows = the MFMA's row operand from LDS, tokens = its column
// operand from registers), so that a lane ends every product with values of ONE token row:
//   FC1   D1^T[hidden][token] = W1_tile (64 hidden x 256 k)  x  ln_2(x)^T        -- the resident slab as in gemm_t32.hip
//   FC2   D2^T[out][token]   += W2_tile (256 out x 64 hidden) x  g_tile^T        -- g_tile straight out of FC1's accumulators
// The link between the two is a choice of WHICH hidden unit sits at which MFMA row of FC1 (free: it only moves the LDS address of the
// operand read): with unit phi(rho) = 16 (rho >> 4) + 8 ((rho >> 2) & 1) + 4 ((rho >> 3) & 1) + (rho & 3) at row rho, the lane (token,
// half) ends a 32-unit block with units 16 s + 8 half .. + 7 in registers 8 s .. 8 s + 7 (s = 0, 1) -- exactly the 8 consecutive k
// values the lane owes FC2's k step s as column operand, and one 16-B store of g (8-B of codes) per k step.  No cross-lane traffic,
// no LDS round trip, no permuted weight copy.
//
// Workgroup = 4 waves, ONE per SIMD, each with 64 token rows (two 32-token blocks: every weight fragment read feeds two MFMAs, LDS
// reads run at half the matrix core's demand): 128 VGPRs of slab, FC2's 2 x 8 accumulator blocks = 256 accumulation registers (AGPR
// form: this file is built without -amdgpu-mfma-vgpr-form), FC1's 2 x 16.  With a single wave per SIMD nothing hides a latency for
// another wave: the LDS fragment reads of the whole tile run as ONE software-pipelined stream (inline assembly, four fragment
// registers in rotation, three reads ahead of the MFMAs, across the FC1 -> NewGELU -> FC2 phase changes), and the weight tiles of
// step j + 1 (32 KiB of W1 + 32 KiB of W2 by global_load_lds) are on their way during the whole of step j.
#include <cstdlib>
#include <type_traits>
#include "../gemm_epi.h"

#define M64_HID 1024
#define M64_C 256
#define M64_TILE 64
#define M64_STAGE 65536
#define M64_OFF_W2 32768
#define M64_BIAS_OFF (2 * M64_STAGE)   // floats: b1[1024] | b2[256] | gamma[256] | beta[256]
#define M64_LDS (M64_BIAS_OFF + (1024 + 256 + 512) * 4)
#ifndef M64_ABLATE
#define M64_ABLATE 0   // probe builds (timing only, results wrong): 1 = no g / codes stores, 2 = no NewGELU arithmetic, 4 = no weight DMA behind the first tile, 8 = no final write-out
#endif

#ifdef COATI_M64_TRACE
// Probe build (-DCOATI_M64_TRACE, tools/probes/m64_trace.py): shader-clock totals per phase of the 4 waves of the first 16 workgroups:
// [slab load + LayerNorm, tile barrier, FC1, NewGELU + stores, FC2, wait for the next tile's DMA, final write-out, whole kernel]
__device__ unsigned long long m64_trace_buf[16 * 4 * 8];
extern "C" int coati_m64_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(m64_trace_buf), sizeof(m64_trace_buf)) == hipSuccess ? 0 : -3;
}
#define M64_T(i) do { unsigned long long m64_now; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m64_now) :: "memory"); m64_acc[i] += m64_now - m64_last; m64_last = m64_now; } while (0)
#else
#define M64_T(i) do { } while (0)
#endif

typedef unsigned m64_v4u __attribute__((ext_vector_type(4)));
typedef unsigned m64_v2u __attribute__((ext_vector_type(2)));

// ---- the tile's stream of 64 LDS fragment reads: n = 16 ph + i, phase ph = 0: FC1 block 0 (i = k step), 1: FC2 of block 0's units (i = 2 ob + s),
// 2: FC1 block 1, 3: FC2 of block 1's units.  Inline assembly (left to the compiler every read sinks to its use: gemm_t32.hip), the
// fragment register is n & 3; the waits count the reads that may stay in flight behind read n (LDS operations return in order, so any
// other LDS operation in between only makes a wait stricter)
template <int N>
__device__ __forceinline__ void m64_issue(bf16x8 (&wf)[4], unsigned p1, unsigned p2) {
  if constexpr (N < 64) {
    constexpr int ph = N >> 4, i = N & 15, slot = N & 3;
    if constexpr ((ph & 1) == 0) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(wf[slot]) : "v"(p1 ^ ((unsigned)i << 5)), "n"((ph >> 1) * 16384));
    } else {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(wf[slot]) : "v"(p2 ^ ((unsigned)(2 * (ph >> 1) + (i & 1)) << 5)), "n"((i >> 1) * 4096));
    }
  }
}
template <int N>
__device__ __forceinline__ void m64_wait(bf16x8 (&wf)[4]) {
  constexpr int slot = N & 3;
  if constexpr (N <= 60) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wf[slot]));
  else if constexpr (N == 61) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wf[slot]));
  else if constexpr (N == 62) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wf[slot]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[slot]));
}
// FC1's accumulators must sit in ordinary VGPRs: FC2's 2 x 8 blocks take all 256 accumulation registers, and left to the compiler
// (AGPR form for every MFMA of the file) 288 accumulators overflow them -- it then spills slab fragments and DMA addresses to scratch,
// and with one wave per SIMD and DMAs in flight every reload's s_waitcnt vmcnt(0) costs microseconds (268 us per launch, measured).
// So these MFMAs are inline assembly with "v" operands; the first k step takes the constant 0 as its addend (no initialisation, no
// dependency), and the NewGELU code waits out the MFMA -> VALU hazard explicitly (the compiler cannot see it behind the asm).
template <int B, int KS>
__device__ __forceinline__ void m64_fc1_step(bf16x8 (&wf)[4], unsigned p1, unsigned p2, const bf16x8 (&af)[2][16], f32x16 (&acc1)[2]) {
  constexpr int n = 32 * B + KS;
  m64_issue<n + 3>(wf, p1, p2);
  m64_wait<n>(wf);
  if constexpr (KS == 0) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc1[0]) : "v"(wf[n & 3]), "v"(af[0][KS]));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc1[1]) : "v"(wf[n & 3]), "v"(af[1][KS]));
  } else {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1[0]) : "v"(wf[n & 3]), "v"(af[0][KS]));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1[1]) : "v"(wf[n & 3]), "v"(af[1][KS]));
  }
}
template <int B, int I>
__device__ __forceinline__ void m64_fc2_step(bf16x8 (&wf)[4], unsigned p1, unsigned p2, const bf16x8 (&hf)[2][2], f32x16 (&acc2)[2][8]) {
  constexpr int n = 32 * B + 16 + I, ob = I >> 1, s = I & 1;
  m64_issue<n + 3>(wf, p1, p2);
  m64_wait<n>(wf);
  acc2[0][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n & 3], hf[0][s], acc2[0][ob], 0, 0, 0);
  acc2[1][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n & 3], hf[1][s], acc2[1][ob], 0, 0, 0);
}
// one 32-unit block of the tile: FC1 (K = 256), bias + NewGELU + derivative on the accumulators, g / codes out, FC2 on the fresh g
template <int B>
__device__ __forceinline__ void m64_sub_tile(bf16x8 (&wf)[4], unsigned p1, unsigned p2, const bf16x8 (&af)[2][16], f32x16 (&acc2)[2][8],
                                             const float* B1, int j, int half, const bool (&rowok)[2], bf16_t* grow0, bf16_t* grow1,
                                             unsigned char* crow0, unsigned char* crow1) {
  f32x16 acc1[2];
  m64_fc1_step<B, 0>(wf, p1, p2, af, acc1); m64_fc1_step<B, 1>(wf, p1, p2, af, acc1); m64_fc1_step<B, 2>(wf, p1, p2, af, acc1); m64_fc1_step<B, 3>(wf, p1, p2, af, acc1);
  m64_fc1_step<B, 4>(wf, p1, p2, af, acc1); m64_fc1_step<B, 5>(wf, p1, p2, af, acc1); m64_fc1_step<B, 6>(wf, p1, p2, af, acc1); m64_fc1_step<B, 7>(wf, p1, p2, af, acc1);
  m64_fc1_step<B, 8>(wf, p1, p2, af, acc1); m64_fc1_step<B, 9>(wf, p1, p2, af, acc1); m64_fc1_step<B, 10>(wf, p1, p2, af, acc1); m64_fc1_step<B, 11>(wf, p1, p2, af, acc1);
  m64_fc1_step<B, 12>(wf, p1, p2, af, acc1); m64_fc1_step<B, 13>(wf, p1, p2, af, acc1); m64_fc1_step<B, 14>(wf, p1, p2, af, acc1); m64_fc1_step<B, 15>(wf, p1, p2, af, acc1);
  // (MFMA result -> VALU read: 8 passes + the write-back; the compiler's hazard recogniser does not look into inline assembly)
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc1[0]), "+v"(acc1[1]));
  // registers 8 s .. 8 s + 7 = units 64 j + 32 B + 16 s + 8 half .. + 7 of the lane's token
  bf16x8 hf[2][2];
  const float* bq = B1 + j * M64_TILE + 32 * B + 8 * half;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float4 c0 = *reinterpret_cast<const float4*>(bq + 16 * s), c1 = *reinterpret_cast<const float4*>(bq + 16 * s + 4);
    const float bs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float h8[8], d8[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
#if M64_ABLATE & 2
        h8[e] = acc1[t][8 * s + e] + bs[e]; h8[e + 1] = acc1[t][8 * s + e + 1] + bs[e + 1]; d8[e] = h8[e]; d8[e + 1] = h8[e + 1];
#else
        coati_v2f hh, dd;
        gelu_and_grad_f2(coati_v2f{acc1[t][8 * s + e] + bs[e], acc1[t][8 * s + e + 1] + bs[e + 1]}, hh, dd);
        h8[e] = hh.x; h8[e + 1] = hh.y; d8[e] = dd.x; d8[e + 1] = dd.y;
#endif
      }
      const uint4 u = pack8(h8);
      hf[t][s] = __builtin_bit_cast(bf16x8, u);
      const uint2 q = packq8(d8);
      const int col = j * M64_TILE + 32 * B + 16 * s;
      if (rowok[t] && !(M64_ABLATE & 1)) {
        *reinterpret_cast<uint4*>((t ? grow1 : grow0) + col) = u;
        __builtin_nontemporal_store(m64_v2u{q.x, q.y}, reinterpret_cast<m64_v2u*>((t ? crow1 : crow0) + col));   // (read by the backward only)
      }
    }
  }
  m64_fc2_step<B, 0>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 1>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 2>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 3>(wf, p1, p2, hf, acc2);
  m64_fc2_step<B, 4>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 5>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 6>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 7>(wf, p1, p2, hf, acc2);
  m64_fc2_step<B, 8>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 9>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 10>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 11>(wf, p1, p2, hf, acc2);
  m64_fc2_step<B, 12>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 13>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 14>(wf, p1, p2, hf, acc2); m64_fc2_step<B, 15>(wf, p1, p2, hf, acc2);
}

__global__ __launch_bounds__(256, 1) void mlp64_fwd_kernel(Mlp64Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gbl_void;
  float* const B1 = reinterpret_cast<float*>(smem + M64_BIAS_OFF);
  float* const B2 = B1 + M64_HID;
  float* const GB = B2 + M64_C;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, half = lane >> 5;
  const int row_base = (blockIdx.x * 4 + wave) * 64;
#ifdef COATI_M64_TRACE
  unsigned long long m64_last, m64_first, m64_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m64_last) :: "memory");
  m64_first = m64_last;
#endif
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_u8*)smem);

  // ---- weight tiles of hidden units 64 j .. + 63: 32 pieces of W1 (two 512-B rows each; chunk c of row f at slot c ^ (f & 31)) and 32 pieces
  // of W2 (eight 128-B rows each; chunk c of row o at slot c ^ ((o >> 1) & 7)); piece i by wave i & 3: 16 DMA instructions per wave and tile
  // (the lane's offsets are recomputed per tile from a laundered lane id: hoisted out of the tile loop they are 16 address pairs = 32 VGPRs)
  auto load_tile = [&](int j, unsigned char* S) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = wave + 4 * k;
      const int f = 2 * i + (lane >> 5), sl = lane & 31;
      const bf16_t* src = p.W1 + (long long)(j * M64_TILE + f) * M64_C + ((sl ^ (f & 31)) << 3);
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(S + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int q = wave + 4 * k;
      const int o = 8 * q + (lane >> 3), c = lane & 7;
      const bf16_t* src = p.W2 + (long long)o * M64_HID + j * M64_TILE + ((c ^ ((o >> 1) & 7)) << 3);
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(S + M64_OFF_W2 + q * 1024), 16, 0, 0);
    }
  };
  load_tile(0, smem);
  for (int c = tid; c < M64_HID; c += 256) B1[c] = p.b1[c];
  B2[tid] = p.b2[tid];
  GB[tid] = p.gamma[tid];
  GB[256 + tid] = p.beta[tid];
  __syncthreads();   // (drains the first tile's DMA as well: it is needed right behind the slab load anyway)

  // ---- resident slab: fragment ks of token block t = k 16 ks + 8 half .. + 7 of row row_base + 32 t + tl, LayerNorm evaluated in the load
  // (basic_transformer.py:165-173: two-pass statistics over the row's 256 values, 128 in this lane, 128 in lane ^ 32)
  bf16x8 af[2][16];
  bool rowok[2];
  int rowi[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = row_base + 32 * t + tl;
    rowok[t] = row < p.M;
    rowi[t] = rowok[t] ? row : p.M - 1;
    const float* xp = p.x + (long long)rowi[t] * p.ldx + half * 8;
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
    const float mean = half_xchg_sum(sm) * (1.0f / M64_C);
    float vs = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = xf[ks][i] - mean; vs = fmaf(d, d, vs); }
    const float rstd = rsqrtf(half_xchg_sum(vs) * (1.0f / M64_C) + 1e-5f);
    if (half == 0 && rowok[t]) { p.mean[row] = mean; p.rstd[row] = rstd; }
    bf16_t* op = p.a2 + (long long)rowi[t] * M64_C + half * 8;
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
      af[t][ks] = __builtin_bit_cast(bf16x8, u);
      // (non-temporal: only the weight gradient reads this copy, a whole forward pass later)
      if (rowok[t]) __builtin_nontemporal_store(m64_v4u{u.x, u.y, u.z, u.w}, reinterpret_cast<m64_v4u*>(op + ks * 16));
    }
  }
  // wave-uniform: how many of the wave's two token blocks hold a valid row (their stores are issued at all): the vmcnt bookkeeping below
  const int live = __builtin_amdgcn_readfirstlane((int)(row_base < p.M) + (int)(row_base + 32 < p.M));

  // ---- FC2 accumulators: token block t, out-feature block ob; register 4 rg + jj = out feature 32 ob + 8 rg + 4 half + jj of the lane's token
  f32x16 acc2[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][ob][r] = 0.f;

  // LDS addresses of the fragment reads (stage 0; stage 1 = + M64_STAGE).  FC1, block b, k step ks: row 32 b + phi, chunk (2 ks + half) ^ phi
  //   = (P1 ^ (ks << 5)) + b * 16384.   FC2, out block ob, k step s of the tile: row 32 ob + tl, chunk (2 s + half) ^ ((tl >> 1) & 7)
  //   = (P2 ^ (s << 5)) + ob * 4096
  const int phi = ((tl >> 4) << 4) | (((tl >> 2) & 1) << 3) | (((tl >> 3) & 1) << 2) | (tl & 3);
  const unsigned P1 = lds0 + (unsigned)(phi * 512 + (((phi >> 1) << 5) | ((half ^ (phi & 1)) << 4)));
  const int sw = (tl >> 1) & 7;
  const unsigned P2 = lds0 + M64_OFF_W2 + (unsigned)(tl * 128 + (((sw >> 1) << 5) | ((half ^ (sw & 1)) << 4)));

  bf16_t* const grow0 = p.g + (long long)rowi[0] * M64_HID + 8 * half;
  bf16_t* const grow1 = p.g + (long long)rowi[1] * M64_HID + 8 * half;
  unsigned char* const crow0 = p.codes + (long long)rowi[0] * M64_HID + 8 * half;
  unsigned char* const crow1 = p.codes + (long long)rowi[1] * M64_HID + 8 * half;

  M64_T(0);
#pragma unroll 1
  for (int j = 0; j < M64_HID / M64_TILE; ++j) {
    // tile j has landed (every wave waited for its own pieces) and every wave is done with tile j - 1, whose stage the next DMAs overwrite
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    M64_T(1);
    if (j + 1 < M64_HID / M64_TILE && !(M64_ABLATE & 4)) load_tile(j + 1, smem + ((j + 1) & 1) * M64_STAGE);
    const unsigned st = (unsigned)((j & 1) * M64_STAGE);
    const unsigned p1 = P1 + st, p2 = P2 + st;

    // ---- one stream of 64 fragment reads per tile (m64_issue / m64_wait above): register n & 3; read n + 3 is issued in front of the
    // MFMAs of read n, across the FC1 -> NewGELU -> FC2 phase changes
    bf16x8 wf[4];
    m64_issue<0>(wf, p1, p2);
    m64_issue<1>(wf, p1, p2);
    m64_issue<2>(wf, p1, p2);
    m64_sub_tile<0>(wf, p1, p2, af, acc2, B1, j, half, rowok, grow0, grow1, crow0, crow1);
    m64_sub_tile<1>(wf, p1, p2, af, acc2, B1, j, half, rowok, grow0, grow1, crow0, crow1);

    // this wave's pieces of tile j + 1 have landed.  vmcnt counts in order: behind them sit this tile's stores (8 per live token block),
    // which are not waited for
    asm volatile("" ::: "memory");
    M64_T(2);
    if (M64_ABLATE & 5) __builtin_amdgcn_s_waitcnt(0x0f70);
    else if (j + 1 < M64_HID / M64_TILE) {
      if (live == 2) __builtin_amdgcn_s_waitcnt(0x4f70);        // vmcnt(16)
      else if (live == 1) __builtin_amdgcn_s_waitcnt(0x0f78);   // vmcnt(8)
      else __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    M64_T(3);
  }

  // ---- out = x + b2 + FC2: register 4 rg + jj of block ob = out feature 32 ob + 8 rg + 4 half + jj: one float4 per (ob, rg)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float* xr = p.x + (long long)rowi[t] * p.ldx + 4 * half;
    float* orow = p.out + (long long)rowi[t] * p.ldo + 4 * half;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) {
      float4 xv[4];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) xv[rg] = *reinterpret_cast<const float4*>(xr + 32 * ob + 8 * rg);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 bb = *reinterpret_cast<const float4*>(B2 + 32 * ob + 8 * rg + 4 * half);
        const float4 o = make_float4(acc2[t][ob][4 * rg] + bb.x + xv[rg].x, acc2[t][ob][4 * rg + 1] + bb.y + xv[rg].y,
                                     acc2[t][ob][4 * rg + 2] + bb.z + xv[rg].z, acc2[t][ob][4 * rg + 3] + bb.w + xv[rg].w);
        if (rowok[t] && (!(M64_ABLATE & 8) || o.x == 1234.5f)) *reinterpret_cast<float4*>(orow + 32 * ob + 8 * rg) = o;
      }
    }
  }
#ifdef COATI_M64_TRACE
  M64_T(4);
  m64_acc[7] = m64_last - m64_first;
  if (blockIdx.x < 16 && lane == 0)
    for (int i = 0; i < 8; ++i) m64_trace_buf[(blockIdx.x * 4 + wave) * 8 + i] = m64_acc[i];
#endif
}

bool mlp64_fwd_supported(int M, int C, int hidden) {
  static const bool on = getenv("COATI_MLP_FUSED") != nullptr;   // opt-in until it has been measured in the step
  return on && C == M64_C && hidden == M64_HID && M >= 24577 && M <= 65536;
}

int launch_mlp64_fwd(const Mlp64Args& a, hipStream_t s) {
  COATI_CHECK_ARG(a.x && a.gamma && a.beta && a.a2 && a.mean && a.rstd && a.W1 && a.b1 && a.W2 && a.b2 && a.g && a.codes && a.out, "mlp64_fwd: null operand");
  COATI_CHECK_SHAPE(a.M > 0 && a.ldx % 4 == 0 && a.ldo % 4 == 0, "mlp64_fwd: unsupported shape (M=%d)", a.M);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp64_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, M64_LDS) != hipSuccess) {
      coati_set_error("mlp64_fwd: hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(mlp64_fwd_kernel, dim3(cdiv(a.M, 256)), dim3(256), M64_LDS, s, a);
  COATI_LAUNCH_CHECK("mlp64_fwd");
  return COATI_OK;
}
