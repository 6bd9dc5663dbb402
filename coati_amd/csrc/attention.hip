// Causal rotary self-attention for head size 16 (grande / "closed": d=256, 16 heads) and 32 (the COATI2-size
// transformer: d=512, 16 heads) -- reference basic_transformer.py:83-100, 126-154 -- forward and backward; every
// function is a template on HS.  One 64-lane wave per (batch row, head); a workgroup = 4 waves = 4 adjacent heads of
// one batch row, so the workgroup consumes whole 128-B lines of the [B*T, 3C] qkv matrix.
//
// Head size 16 = exactly one K step of v_mfma_f32_32x32x16_bf16, so the kernels are softmax / LDS / latency
// bound, not MFMA bound.  The operands a kernel sweeps over (T <= 256 tokens x 16 dims: K,V in the forward and dQ
// kernels, Q,dO in the dK/dV kernel) sit in LDS as ROW-MAJOR [T][16] bf16 images; the operand that is fixed for a
// 32-row block is read from global memory straight into an MFMA fragment, so a workgroup needs only 2 images per head:
//   * q, k arrive already rotated (the QKV GEMM applies RoPE in its epilogue); dq, dk are rotated back on store;
//   * scores are computed TRANSPOSED (S^T = K Q^T) so each lane owns one query column and the softmax
//     row statistics are lane-local (+ one cross-half shuffle);
//   * P^T leaves the MFMA accumulator in exactly the layout the next MFMA wants as its B operand, provided
//     the A operand (V^T, K^T, ...) is read with the same key permutation.  That permuted, transposed A
//     fragment comes straight out of the row-major image with two ds_read_b64_tr_b16 (the gfx950 LDS
//     transpose read: a 16-lane group fetches a 4-row x 16-col block, lane i receives column i);
//   * the [T,T] score matrix is never materialised; the backward recomputes P from the saved log-sum-exp
//     in two kernels (dQ; dK+dV) -- no atomics, deterministic;
//   * results leave a workgroup as whole 128-B row segments through a shared LDS tile (tile_put / tile_store).
// Layout: qkv [B*T, 3C] bf16 (q | k | v, head h at columns h*HS..h*HS+HS-1), y / dy [B*T, C], lse, D [B, nh, T].
#include <cstdlib>
#include "kernels.h"

#include "attn_img.h"

// A/B fragment of a row-major [*,HS] image for reduction step ks (16 dims each):
// lane (r = lane&31, half = lane>>5) -> row blk*32+r, dims ks*16 + half*8..+7
template <int HS>
__device__ __forceinline__ bf16x8 rfrag(const bf16_t* rm, int blk, int ks, int lane) {
  // (blk * 32 rows do not change the swizzle: the offset inside the block is a function of the lane alone)
  const int tl = lane & 31;
  return *reinterpret_cast<const bf16x8*>(rm + blk * 32 * HS + (tl * HS + img_chunk<HS>(tl, ks * 2 + (lane >> 5)) * 8));
}
// The same fragment straight from global memory (rows >= T read as zero): for the operand that is needed for ONE
// 32-row block only, so it never occupies LDS.
__device__ __forceinline__ bf16x8 gfrag(const bf16_t* src, long long stride, int blk, int ks, int T, int lane) {
  const int row = blk * 32 + (lane & 31);
  const int rc = row < T ? row : T - 1;
  uint4 u = *reinterpret_cast<const uint4*>(src + (long long)rc * stride + ks * 16 + (lane >> 5) * 8);
  if (row >= T) u = make_uint4(0, 0, 0, 0);
  return __builtin_bit_cast(bf16x8, u);
}
// A fragment X^T[d][row] with the accumulator's row permutation, read from the row-major image with the LDS
// transpose read: lane (d = lane&31, h = lane>>5), slot j <-> row base + 4h + (j&3) + 8*(j>>2).  HS = 16: lanes with
// d >= 16 feed don't-care rows; HS = 32: the second 16-lane group of each half reads dims 16..31.
template <int HS>
__device__ __forceinline__ bf16x8 tfrag(const bf16_t* rm, int base, int lane) {
  typedef __attribute__((address_space(3))) v4s16a lds_v4;
  // this lane's 8-B piece: dims 4 (lane & 3) .. + 3 (+ 16 for the second 16-lane group at HS = 32) = chunk c, half (lane & 1);
  // rows t and t + 8 of the two reads may swizzle differently (HS = 16: they do)
  // (base is a multiple of 16 rows: it does not change the swizzle, so both offsets are functions of the lane alone and every
  // call shares the same two address registers)
  const int tl = 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int c = ((lane & 3) >> 1) + (HS == 32 ? 2 * ((lane >> 4) & 1) : 0), h8 = 4 * (lane & 1);
  const bf16_t* p = rm + base * HS + (tl * HS + img_chunk<HS>(tl, c) * 8 + h8);
  const bf16_t* p2 = rm + base * HS + ((tl + 8) * HS + img_chunk<HS>(tl + 8, c) * 8 + h8);
  const v4s16a lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p);
  const v4s16a hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p2);
  // lanes with d >= 16 feed don't-care rows of the A operand: MFMA output rows are independent and rows 16..31 of the
  // result (accumulator registers 8..15) are never read, so no masking is needed.
  const v8s16a r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 pfrag(const float* p) {  // 8 accumulator values -> bf16 B fragment
  const uint4 u = pack8(p);
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
__device__ __forceinline__ int arow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- output path: results leave the workgroup as whole 128-B row segments --------------------------------------
// An accumulator block holds, per lane, 2 x 4 dims of ONE row; storing that directly is 8-B granules scattered over 32
// rows per instruction (address-coalescer / partial-line bound: it cost ~40 of the 78 us of the dK/dV kernel).  Each
// wave instead drops its head's 32 x 16 block into a workgroup-shared LDS tile [32 rows][4 heads x 16 dims] (row pitch
// 144 B), and after a barrier the 256 threads store the tile as 16-B chunks, 8 lanes per 128-B row.
template <int HS> __host__ __device__ __forceinline__ constexpr int ot_pitch() { return 4 * HS * 2 + 16; }   // bytes per tile row (144 / 272)
template <int HS> __host__ __device__ __forceinline__ constexpr int ot_bytes() { return 32 * ot_pitch<HS>(); }
// v[0 .. HS/2): the lane's live accumulator registers of one row; register group g (4 regs) = dims 8g + 4*half .. +3
template <int HS>
__device__ __forceinline__ void tile_put(unsigned char* tile, int wave, int lane, const float* v) {
  unsigned char* row = tile + (lane & 31) * ot_pitch<HS>() + wave * HS * 2 + (lane >> 5) * 8;
#pragma unroll
  for (int g = 0; g < HS / 8; ++g)
    *reinterpret_cast<uint2*>(row + 16 * g) = make_uint2(pack2bf(v[4 * g], v[4 * g + 1]), pack2bf(v[4 * g + 2], v[4 * g + 3]));
}
// gradient block (lane = token row, register group g = dims 8g + 4*half + j) -> tile, with the inverse rotation of the
// RoPE pairs (d, d + HS/2) = register groups (g, g + HS/16)
template <int HS>
__device__ __forceinline__ void tile_put_grad(unsigned char* tile, int wave, int lane, const f32x16& g, bool rope_inv,
                                              const float* cos_t, const float* sin_t, int t) {
  const int half = lane >> 5;
  float v[HS / 2];
#pragma unroll
  for (int j = 0; j < HS / 2; ++j) v[j] = g[j];
  if (rope_inv) {
#pragma unroll
    for (int gq = 0; gq < HS / 16; ++gq) {   // pair (group gq, group gq + HS/16): dims 8gq + 4half + j and + HS/2
      const float4 cs = *reinterpret_cast<const float4*>(cos_t + t * HS + 8 * gq + 4 * half);
      const float4 sn = *reinterpret_cast<const float4*>(sin_t + t * HS + 8 * gq + 4 * half);
      const float csv[4] = {cs.x, cs.y, cs.z, cs.w}, snv[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ga = v[4 * gq + j], gc = v[4 * (gq + HS / 16) + j];
        v[4 * gq + j] = ga * csv[j] + gc * snv[j];
        v[4 * (gq + HS / 16) + j] = gc * csv[j] - ga * snv[j];
      }
    }
  }
  tile_put<HS>(tile, wave, lane, v);
}
// the same with the rotation rows already in registers (fetched early, so their global round trip hides behind the sweep)
template <int HS>
struct RopeRow { float4 cs[HS / 16], sn[HS / 16]; };
template <int HS>
__device__ __forceinline__ RopeRow<HS> rope_row_load(const float* cos_t, const float* sin_t, int t, int lane) {
  RopeRow<HS> r;
#pragma unroll
  for (int gq = 0; gq < HS / 16; ++gq) {
    r.cs[gq] = *reinterpret_cast<const float4*>(cos_t + t * HS + 8 * gq + 4 * (lane >> 5));
    r.sn[gq] = *reinterpret_cast<const float4*>(sin_t + t * HS + 8 * gq + 4 * (lane >> 5));
  }
  return r;
}
template <int HS>
__device__ __forceinline__ void tile_put_grad_rot(unsigned char* tile, int wave, int lane, const f32x16& g, const RopeRow<HS>& rr) {
  float v[HS / 2];
#pragma unroll
  for (int j = 0; j < HS / 2; ++j) v[j] = g[j];
#pragma unroll
  for (int gq = 0; gq < HS / 16; ++gq) {
    const float csv[4] = {rr.cs[gq].x, rr.cs[gq].y, rr.cs[gq].z, rr.cs[gq].w}, snv[4] = {rr.sn[gq].x, rr.sn[gq].y, rr.sn[gq].z, rr.sn[gq].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ga = v[4 * gq + j], gc = v[4 * (gq + HS / 16) + j];
      v[4 * gq + j] = ga * csv[j] + gc * snv[j];
      v[4 * (gq + HS / 16) + j] = gc * csv[j] - ga * snv[j];
    }
  }
  tile_put<HS>(tile, wave, lane, v);
}
// cooperative store of one tile: task -> (row, 16-B chunk); dst points at (row 0, first head of the quad)
template <int HS>
__device__ __forceinline__ void tile_store(const unsigned char* tile, bf16_t* dst, long long stride, int row0, int T,
                                           int heads_here, int tid) {
  constexpr int CPR = HS / 2, CPH = HS / 8;
#pragma unroll
  for (int task = tid; task < 32 * CPR; task += 256) {
    const int row = task / CPR, ch = task - row * CPR;
    if (row0 + row < T && ch / CPH < heads_here)
      *reinterpret_cast<uint4*>(dst + (long long)(row0 + row) * stride + ch * 8) =
          *reinterpret_cast<const uint4*>(tile + row * ot_pitch<HS>() + ch * 16);
  }
}

// ---------------------------------------------------------------------------------------------------
// S^T block (keys x queries) = sum over the HS/16 reduction steps
template <int HS>
__device__ __forceinline__ f32x16 score_block(const bf16_t* Xs, int blk, const bf16x8 (&f)[HS / 16], int lane) {
  f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag<HS>(Xs, blk, 0, lane), f[0], zero16(), 0, 0, 0);
  if constexpr (HS == 32) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfrag<HS>(Xs, blk, 1, lane), f[1], s, 0, 0, 0);
  return s;
}

// NB > 0: the sequence fits NB 32-row blocks and every loop bound is a compile-time constant (the staging loads of a
// workgroup are then issued back to back instead of one load -> store round trip per iteration); NB = 0: any T <= 256.
template <int HS, int NB>
__device__ __forceinline__ void attn_fwd_body(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ y, float* __restrict__ lse,
                                              int Tl, int n_head, int b, int hq, int T, long long row0) {
  constexpr int NK = HS / 16, LIVE = HS / 2;
  constexpr float SCALE = att_scale<HS>(), SCALE_LOG2E = att_scale_log2e<HS>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = hq * 4 + wave;
  const int heads_here = (n_head - hq * 4) < 4 ? (n_head - hq * 4) : 4;
  const int C = n_head * HS, Tp = NB ? 32 * NB : ((T + 31) & ~31);
  const size_t pw = (size_t)2 * Tp * HS * 2 + ATT_PW_PAD;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem + (size_t)wave * pw);
  bf16_t* Vs = Ks + Tp * HS;
  const long long stride = 3LL * C;
  const bf16_t* base = qkv + row0 * stride + hq * 4 * HS;
  const bool active = hh < n_head;   // inactive waves of a partial quad only take part in the barriers
  const bf16_t* qsrc = qkv + row0 * stride + (active ? hh : 0) * HS;
  bf16x8 qnext[NK];   // the Q fragment of the next query block: fetched one block ahead, so only the first one is waited for
  if constexpr (NB > 0) {
    // every load of the prologue is issued before the first LDS write: one memory round trip for K, V and the first Q fragment
    Stage4Regs<HS, NB ? 32 * NB : 32> rk, rv;
    stage4_load<HS, NB ? 32 * NB : 32>(rk, base + C, stride, T, heads_here, threadIdx.x);
    stage4_load<HS, NB ? 32 * NB : 32>(rv, base + 2 * C, stride, T, heads_here, threadIdx.x);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) qnext[ks] = gfrag(qsrc, stride, 0, ks, T, lane);
    stage4_store<HS, NB ? 32 * NB : 32>(rk, T, smem, pw, 0, heads_here, threadIdx.x);
    stage4_store<HS, NB ? 32 * NB : 32>(rv, T, smem, pw, 1, heads_here, threadIdx.x);
  } else {
    stage4<HS>(base + C, stride, T, Tp, smem, pw, 0, heads_here, threadIdx.x);
    stage4<HS>(base + 2 * C, stride, T, Tp, smem, pw, 1, heads_here, threadIdx.x);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) qnext[ks] = gfrag(qsrc, stride, 0, ks, T, lane);
  }
  __syncthreads();
  unsigned char* const otile = smem + 4 * pw;
  bf16_t* const ydst = y + row0 * C + hq * 4 * HS;

  const int nblk = Tp >> 5, half = lane >> 5;
#pragma unroll
  for (int qb = 0; qb < nblk; ++qb) {
    const int q = qb * 32 + (lane & 31);
    if (active) {
      bf16x8 qf[NK];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        qf[ks] = qnext[ks];
        if (qb + 1 < nblk) qnext[ks] = gfrag(qsrc, stride, qb + 1, ks, T, lane);
      }
      float m_run = -INFINITY, l_run = 0.f;
      f32x16 o = zero16();
      for (int kb = 0; kb <= qb; ++kb) {
        f32x16 s = score_block<HS>(Ks, kb, qf, lane);
        // running max / sum are kept on the RAW scores (the scale is positive); exp(x*scale) = 2^(x*scale*log2e)
        float p[16];
        float mloc = -INFINITY;
        if (kb == qb) {   // only the diagonal block needs the causal mask
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + arow(r, lane);
            p[r] = (key <= q) ? s[r] : -INFINITY;
            mloc = fmaxf(mloc, p[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            p[r] = s[r];
            mloc = fmaxf(mloc, p[r]);
          }
        }
        mloc = half_xchg_max(mloc);
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * SCALE_LOG2E);
        const float mc = m_new * SCALE_LOG2E;
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[r] = __builtin_amdgcn_exp2f(fmaf(p[r], SCALE_LOG2E, -mc));
          lsum += p[r];
        }
        lsum = half_xchg_sum(lsum);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < LIVE; ++r) o[r] *= alpha;   // only d < HS is live (regs 0..HS/2-1)
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Vs, kb * 32, lane), pfrag(p), o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Vs, kb * 32 + 16, lane), pfrag(p + 8), o, 0, 0, 0);
      }
      const float inv = 1.0f / l_run;
      float v[LIVE];
#pragma unroll
      for (int r = 0; r < LIVE; ++r) v[r] = o[r] * inv;
      tile_put<HS>(otile, wave, lane, v);
      if (q < T && half == 0) lse[((long long)b * n_head + hh) * Tl + q] = m_run * SCALE + __logf(l_run);
    }
    __syncthreads();
    tile_store<HS>(otile, ydst, C, qb * 32, T, heads_here, threadIdx.x);
    __syncthreads();
  }
}

template <int HS, int NB>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ y,
                                                       float* __restrict__ lse, int Tl, int n_head, int quads,
                                                       const int* __restrict__ seq_off) {
  const int b = blockIdx.x / quads, hq = blockIdx.x - b * quads;
  ATT_SEQ(0);
  attn_fwd_body<HS, NB>(qkv, y, lse, Tl, n_head, b, hq, T, row0);
}
// (Round 4 occupancy probe -- extra dynamic LDS per workgroup, profiles/r04_attn_occupancy.txt: 5 workgroups per CU 31.8 us per launch,
// 3: 40.2, 2: 40.5, 1: 64.7 -- the kernel is close to its VALU / LDS throughput at 5, a persistent form that prefetches the next
// sequence's operands would buy little.)
// packed rows, T <= 32 * NBMAX <= 128: ONE launch; every workgroup runs the body compiled for its own sequence's block count
// (separate launches per block count, the first form, serialised three under-filled grids: no gain over the padded batch)
template <int HS, int NBMAX>
__global__ __launch_bounds__(256) void attn_fwd_varlen_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ y,
                                                              float* __restrict__ lse, int Tl, int n_head, int quads,
                                                              const int* __restrict__ seq_off) {
  const int b = blockIdx.x / quads, hq = blockIdx.x - b * quads;
  ATT_SEQ(0);
  const int nb = (T + 31) >> 5;   // uniform over the workgroup
  if (nb == 1) attn_fwd_body<HS, 1>(qkv, y, lse, Tl, n_head, b, hq, T, row0);
  else if (NBMAX >= 2 && nb == 2) attn_fwd_body<HS, (NBMAX >= 2 ? 2 : 1)>(qkv, y, lse, Tl, n_head, b, hq, T, row0);
  else if (NBMAX >= 3 && nb == 3) attn_fwd_body<HS, (NBMAX >= 3 ? 3 : 1)>(qkv, y, lse, Tl, n_head, b, hq, T, row0);
  else if (NBMAX >= 4 && nb == 4) attn_fwd_body<HS, (NBMAX >= 4 ? 4 : 1)>(qkv, y, lse, Tl, n_head, b, hq, T, row0);
}

template <int HS, int NB>
static int launch_attn_fwd_t(const bf16_t* qkv, bf16_t* y, float* lse, int B, int T, int n_head, hipStream_t s, const int* seq_off) {
  const int Tp = NB ? 32 * NB : ((T + 31) & ~31);
  const size_t lds = (size_t)4 * (2 * Tp * HS * 2 + ATT_PW_PAD) + ot_bytes<HS>();
  if (seq_off != nullptr && NB > 0) {   // packed rows: one launch, per-workgroup block count
    static bool attr_v = false;
    if (!attr_v) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_varlen_kernel<HS, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        coati_set_error("attn_fwd(varlen): hipFuncSetAttribute failed");
        return COATI_EHIP;
      }
      attr_v = true;
    }
    const int quads = cdiv(n_head, 4);
    hipLaunchKernelGGL((attn_fwd_varlen_kernel<HS, NB>), dim3(B * quads), dim3(256), lds, s, qkv, y, lse, T, n_head, quads, seq_off);
    COATI_LAUNCH_CHECK("attn_fwd_varlen");
    return COATI_OK;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<HS, NB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      coati_set_error("attn_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int quads = cdiv(n_head, 4);
  hipLaunchKernelGGL((attn_fwd_kernel<HS, NB>), dim3(B * quads), dim3(256), lds, s, qkv, y, lse, T, n_head, quads, seq_off);
  COATI_LAUNCH_CHECK("attn_fwd");
  return COATI_OK;
}

int launch_attn_fwd(const bf16_t* qkv, bf16_t* y, float* lse, int B, int T, int n_head, int head_size, hipStream_t s, const int* seq_off, const int* seq_ord) {
  COATI_CHECK_ARG(qkv && y && lse, "attn_fwd: null operand");
  COATI_CHECK_SHAPE(B > 0 && T > 0 && T <= 256 && n_head > 0 && (head_size == 16 || head_size == 32),
                    "attn_fwd: unsupported shape B=%d T=%d nh=%d hs=%d", B, T, n_head, head_size);
  // round 6: head size 16, T <= 128 on 16-row causal granularity (attention16.hip); COATI_ATTN_BLOCK32=1 keeps the 32-row kernels for A/B runs
  static const bool block32 = getenv("COATI_ATTN_BLOCK32") != nullptr;
  if (head_size == 16 && T <= 128 && !block32) return launch_attn16_fwd(qkv, y, lse, B, T, n_head, s, seq_off, seq_ord);
  const int nb = (T + 31) / 32;
#define FWD_CASE(H, N) if (head_size == H && nb == N) return launch_attn_fwd_t<H, N>(qkv, y, lse, B, T, n_head, s, seq_off);
  FWD_CASE(16, 1) FWD_CASE(16, 2) FWD_CASE(16, 3) FWD_CASE(16, 4)
  FWD_CASE(32, 1) FWD_CASE(32, 2) FWD_CASE(32, 3) FWD_CASE(32, 4)
#undef FWD_CASE
  return head_size == 16 ? launch_attn_fwd_t<16, 0>(qkv, y, lse, B, T, n_head, s, seq_off) : launch_attn_fwd_t<32, 0>(qkv, y, lse, B, T, n_head, s, seq_off);
}

// ---------------------------------------------------------------------------------------------------
// backward.  dS = P * (dP - D) * scale with D[q] = sum_d dO[q,d] O[q,d].
//   kernel 1 (per query block): dQ^T[d][q] = sum_keys K^T[d][key] dS^T[key][q]          (also writes D)
//   kernel 2 (per key block)  : dK^T[d][key] = sum_q Q^T[d][q] dS[q][key],  dV^T[d][key] = sum_q dO^T[d][q] P[q][key]
// ---------------------------------------------------------------------------------------------------
template <int HS>
__global__ __launch_bounds__(256, HS == 16 ? 4 : 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ y,
                                                          const bf16_t* __restrict__ dy, const float* __restrict__ lse,
                                                          float* __restrict__ Dout, bf16_t* __restrict__ dqkv,
                                                          const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                          int Tl, int n_head, int quads, const int* __restrict__ seq_off) {
  constexpr int NK = HS / 16, LIVE = HS / 2;
  constexpr float SCALE = att_scale<HS>(), SCALE_LOG2E = att_scale_log2e<HS>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / quads, hq = blockIdx.x - b * quads;
  ATT_SEQ(0);
  const int hh = hq * 4 + wave;
  const int heads_here = (n_head - hq * 4) < 4 ? (n_head - hq * 4) : 4;
  const int C = n_head * HS, Tp = (T + 31) & ~31;
  const size_t pw = (size_t)2 * Tp * HS * 2 + ATT_PW_PAD;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem + (size_t)wave * pw);
  bf16_t* Vs = Ks + Tp * HS;
  const long long stride = 3LL * C;
  const bf16_t* qbase = qkv + row0 * stride + hq * 4 * HS;
  stage4<HS>(qbase + C, stride, T, Tp, smem, pw, 0, heads_here, threadIdx.x);
  stage4<HS>(qbase + 2 * C, stride, T, Tp, smem, pw, 1, heads_here, threadIdx.x);
  __syncthreads();
  const bool active = hh < n_head;
  const int hc = active ? hh : 0;
  unsigned char* const otile = smem + 4 * pw;
  const bf16_t* ybase = y + row0 * C + hc * HS;
  const bf16_t* qsrc = qkv + row0 * stride + hc * HS;
  const bf16_t* gsrc = dy + row0 * C + hc * HS;
  const long long sbase = ((long long)b * n_head + hc) * Tl;

  const int nblk = Tp >> 5, half = lane >> 5;
  bf16_t* const dbase = dqkv + row0 * stride + hq * 4 * HS;
  for (int qb = 0; qb < nblk; ++qb) {
    const int q = qb * 32 + (lane & 31);
    if (active) {
      bf16x8 qf[NK], gf[NK];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        qf[ks] = gfrag(qsrc, stride, qb, ks, T, lane);
        gf[ks] = gfrag(gsrc, (long long)C, qb, ks, T, lane);
      }
      // D[q] = sum_d dO[q,d] O[q,d]: this lane holds dims ks*16 + half*8..+7 of dO[q] in gf; the partner half-wave adds
      // the rest
      float lq = INFINITY, dq_ = 0.f;
      if (q < T) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          float o8[8], g8[8];
          unpack8(*reinterpret_cast<const uint4*>(ybase + (long long)q * C + ks * 16 + half * 8), o8);
          unpack8(__builtin_bit_cast(uint4, gf[ks]), g8);
#pragma unroll
          for (int i = 0; i < 8; ++i) dq_ += o8[i] * g8[i];
        }
        lq = lse[sbase + q] * LOG2E;
      }
      dq_ += __shfl_xor(dq_, 32, 64);
      if (q < T && half == 0) Dout[sbase + q] = dq_;
      f32x16 acc = zero16();
      for (int kb = 0; kb <= qb; ++kb) {
        const f32x16 s = score_block<HS>(Ks, kb, qf, lane);
        const f32x16 dp = score_block<HS>(Vs, kb, gf, lane);
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(s[r], SCALE_LOG2E, -lq));
          if (kb == qb && kb * 32 + arow(r, lane) > q) p = 0.f;
          ds[r] = p * (dp[r] - dq_);   // the softmax scale is applied once to the finished dQ block
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Ks, kb * 32, lane), pfrag(ds), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Ks, kb * 32 + 16, lane), pfrag(ds + 8), acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < LIVE; ++r) acc[r] *= SCALE;
      tile_put_grad<HS>(otile, wave, lane, acc, true, cos_t, sin_t, q < T ? q : 0);
    }
    __syncthreads();
    tile_store<HS>(otile, dbase, stride, qb * 32, T, heads_here, threadIdx.x);
    __syncthreads();
  }
}

template <int HS>
__global__ __launch_bounds__(256, HS == 16 ? 3 : 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dy,
                                                           const float* __restrict__ lse, const float* __restrict__ Din,
                                                           bf16_t* __restrict__ dqkv, const float* __restrict__ cos_t,
                                                           const float* __restrict__ sin_t, int Tl, int n_head, int quads,
                                                           const int* __restrict__ seq_off) {
  constexpr int NK = HS / 16, LIVE = HS / 2;
  constexpr float SCALE = att_scale<HS>(), SCALE_LOG2E = att_scale_log2e<HS>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / quads, hq = blockIdx.x - b * quads;
  ATT_SEQ(0);
  const int hh = hq * 4 + wave;
  const int heads_here = (n_head - hq * 4) < 4 ? (n_head - hq * 4) : 4;
  const int C = n_head * HS, Tp = (T + 31) & ~31;
  const size_t pw = (size_t)2 * Tp * HS * 2 + (size_t)2 * Tp * 4 + ATT_PW_PAD;
  unsigned char* my = smem + (size_t)wave * pw;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(my);
  bf16_t* Gs = Qs + Tp * HS;
  float* Ls = reinterpret_cast<float*>(Gs + Tp * HS);
  float* Ds = Ls + Tp;
  const long long stride = 3LL * C;
  const bf16_t* qbase = qkv + row0 * stride + hq * 4 * HS;
  stage4<HS>(qbase, stride, T, Tp, smem, pw, 0, heads_here, threadIdx.x);
  stage4<HS>(dy + row0 * C + hq * 4 * HS, (long long)C, T, Tp, smem, pw, 1, heads_here, threadIdx.x);
  __syncthreads();
  const bool active = hh < n_head;
  const int hc = active ? hh : 0;
  unsigned char* const otile = smem + 4 * pw;   // two tiles: dK, dV
  const bf16_t* ksrc = qkv + row0 * stride + C + hc * HS;
  for (int t = lane; t < Tp; t += 64) {
    const long long o = ((long long)b * n_head + hc) * Tl + t;
    Ls[t] = (t < T) ? lse[o] : INFINITY;
    Ds[t] = (t < T) ? Din[o] : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();

  const int nblk = Tp >> 5, half = lane >> 5;
  bf16_t* const dbase = dqkv + row0 * stride + hq * 4 * HS;
  for (int kb = 0; kb < nblk; ++kb) {
    const int key = kb * 32 + (lane & 31);
    if (active) {
      bf16x8 kf[NK], vf[NK];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        kf[ks] = gfrag(ksrc, stride, kb, ks, T, lane);
        vf[ks] = gfrag(ksrc + C, stride, kb, ks, T, lane);
      }
      f32x16 dk = zero16(), dv = zero16();
      for (int qb = kb; qb < nblk; ++qb) {
        const f32x16 s = score_block<HS>(Qs, qb, kf, lane);
        const f32x16 dp = score_block<HS>(Gs, qb, vf, lane);
        float p[16], ds[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int q0 = qb * 32 + 8 * g4 + 4 * half;
          const float4 l4 = *reinterpret_cast<const float4*>(Ls + q0);
          const float4 d4 = *reinterpret_cast<const float4*>(Ds + q0);
          const float lv[4] = {l4.x * LOG2E, l4.y * LOG2E, l4.z * LOG2E, l4.w * LOG2E}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = g4 * 4 + j, q = q0 + j;
            p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], SCALE_LOG2E, -lv[j]));
            if (qb == kb && key > q) p[r] = 0.f;
            ds[r] = p[r] * (dp[r] - dvv[j]);   // scale applied once to the finished dK block
          }
        }
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Gs, qb * 32, lane), pfrag(p), dv, 0, 0, 0);
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Gs, qb * 32 + 16, lane), pfrag(p + 8), dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Qs, qb * 32, lane), pfrag(ds), dk, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Qs, qb * 32 + 16, lane), pfrag(ds + 8), dk, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < LIVE; ++r) dk[r] *= SCALE;
      tile_put_grad<HS>(otile, wave, lane, dk, true, cos_t, sin_t, key < T ? key : 0);
      tile_put_grad<HS>(otile + ot_bytes<HS>(), wave, lane, dv, false, cos_t, sin_t, 0);
    }
    __syncthreads();
    tile_store<HS>(otile, dbase + C, stride, kb * 32, T, heads_here, threadIdx.x);
    tile_store<HS>(otile + ot_bytes<HS>(), dbase + 2 * C, stride, kb * 32, T, heads_here, threadIdx.x);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// Fused backward for T <= 32 * NB (NB <= 4): ONE sweep over the (key block, query block) pairs produces dK, dV and dQ.
// S, dP and the exponentials are evaluated once per pair in the dK/dV orientation (lane = key); dS is then dropped as
// bf16 into a wave-private [32 keys][32 queries] LDS tile and read back TRANSPOSED (ds_read_b64_tr_b16) as the B operand
// of dQ^T += K^T dS^T.  dQ of all NB query blocks stays in registers until the end.  All four operands are staged once.
// ---------------------------------------------------------------------------------------------------
#define DST_PITCH 36   // halfs per row of the dS tile (72 B); the 4 wave-private tiles alias the two output tiles
__device__ __forceinline__ bf16x8 dst_frag(const bf16_t* tile, int base, int lane) {   // B fragment: keys base.., column q
  typedef __attribute__((address_space(3))) v4s16a lds_v4;
  const bf16_t* p = tile + (base + 4 * (lane >> 5) + ((lane & 15) >> 2)) * DST_PITCH + 4 * (lane & 3) + 16 * ((lane >> 4) & 1);
  const v4s16a lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p);
  const v4s16a hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p + 8 * DST_PITCH));
  const v8s16a r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <int HS, int NB>
__device__ __forceinline__ void attn_bwd_fused_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ y,
                                                    const bf16_t* __restrict__ dy, const float* __restrict__ lse,
                                                    bf16_t* __restrict__ dqkv, const float* __restrict__ cos_t,
                                                    const float* __restrict__ sin_t, int Tl, int n_head, int b, int hq, int T, long long row0) {
  constexpr int NK = HS / 16, LIVE = HS / 2, Tp = 32 * NB;
  constexpr float SCALE = att_scale<HS>(), SCALE_LOG2E = att_scale_log2e<HS>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = hq * 4 + wave;
  const int heads_here = (n_head - hq * 4) < 4 ? (n_head - hq * 4) : 4;
  const int C = n_head * HS;
  constexpr size_t pw = (size_t)3 * Tp * HS * 2 + (size_t)2 * Tp * 4 + ATT_PW_PAD;
  unsigned char* my = smem + (size_t)wave * pw;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(my);
  bf16_t* Ks = Qs + Tp * HS;
  bf16_t* Gs = Ks + Tp * HS;
  float* Ls = reinterpret_cast<float*>(Gs + Tp * HS);
  float* Ds = Ls + Tp;
  const long long stride = 3LL * C;
  const bf16_t* qbase = qkv + row0 * stride + hq * 4 * HS;
  const bool active = hh < n_head;
  const int hc = active ? hh : 0;
  const bf16_t* vsrc = qkv + row0 * stride + 2 * C + hc * HS;
  // Every global load of the prologue is issued before the first LDS write: the four staged operands (Q, K, dO, O), this
  // head's log-sum-exp rows and the V fragment of key block 0 cost ONE memory round trip instead of one per staging task
  // (SQ counters of the looped version: 75 % of the wave cycles parked in s_waitcnt).
  Stage4Regs<HS, Tp> rq, rk, rg, ro;
  stage4_load<HS, Tp>(rq, qbase, stride, T, heads_here, threadIdx.x);
  stage4_load<HS, Tp>(rk, qbase + C, stride, T, heads_here, threadIdx.x);
  stage4_load<HS, Tp>(rg, dy + row0 * C + hq * 4 * HS, C, T, heads_here, threadIdx.x);
  stage4_load<HS, Tp>(ro, y + row0 * C + hq * 4 * HS, C, T, heads_here, threadIdx.x);
  float lrow[(Tp + 63) / 64];
#pragma unroll
  for (int i = 0; i < (Tp + 63) / 64; ++i) {
    const int t = lane + 64 * i;
    lrow[i] = lse[((long long)b * n_head + hc) * Tl + (t < T ? t : T - 1)];
  }
  bf16x8 vnext[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) vnext[ks] = gfrag(vsrc, stride, 0, ks, T, lane);
  stage4_store<HS, Tp>(rq, T, smem, pw, 0, heads_here, threadIdx.x);
  stage4_store<HS, Tp>(rk, T, smem, pw, 1, heads_here, threadIdx.x);
  stage4_store<HS, Tp>(rg, T, smem, pw, 2, heads_here, threadIdx.x);
  {  // D[t] = sum_d dO[t,d] O[t,d]: the thread that staged a 16-B chunk of dO holds the same chunk of O; the HS/8 chunk lanes
     // of a head add up through shuffles (masked rows / heads contribute nothing: their dO chunk is zeroed below)
    constexpr int CPR = HS / 2, CPH = HS / 8, NT = (Tp * CPR + 255) / 256;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int task = threadIdx.x + 256 * i, t = task / CPR, c = task - t * CPR, w = c / CPH;
      const bool ok = t < T && w < heads_here;
      float g8[8], o8[8];
      unpack8(rg.v[i], g8);
      unpack8(ro.v[i], o8);
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d += g8[e] * o8[e];
      d = ok ? d : 0.f;
      d += __shfl_xor(d, 1, 64);
      if constexpr (CPH == 4) d += __shfl_xor(d, 2, 64);
      if ((Tp * CPR % 256 == 0 || task < Tp * CPR) && (c & (CPH - 1)) == 0)
        reinterpret_cast<float*>(smem + (size_t)w * pw + (size_t)3 * Tp * HS * 2)[Tp + t] = d;
    }
  }
  // log-sum-exp of this head's rows, pre-multiplied by log2(e) (rows >= T: +inf -> p = 0)
#pragma unroll
  for (int i = 0; i < (Tp + 63) / 64; ++i) {
    const int t = lane + 64 * i;
    if (t < Tp) Ls[t] = (t < T) ? lrow[i] * LOG2E : INFINITY;
  }
  __syncthreads();
  unsigned char* const otile = smem + 4 * pw;   // two tiles (dK | dV, then dQ); during a sweep they hold the dS tiles
  static_assert(4 * 32 * DST_PITCH * 2 <= 2 * ot_bytes<HS>(), "dS tiles must fit the output tiles");
  bf16_t* const dsT = reinterpret_cast<bf16_t*>(otile) + wave * 32 * DST_PITCH;

  const int half = lane >> 5;
  const bool tail16 = T - (NB - 1) * 32 <= 16;
  bf16_t* const dbase = dqkv + row0 * stride + hq * 4 * HS;
  f32x16 dq[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) dq[i] = zero16();
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    const int key = kb * 32 + (lane & 31);
    if (active) {
      // issued now, used after the sweep over the query blocks: the rotation rows of this key block and the V fragment of the next
      const RopeRow<HS> rot = rope_row_load<HS>(cos_t, sin_t, key < T ? key : 0, lane);
      bf16x8 kf[NK], vf[NK];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        kf[ks] = rfrag<HS>(Ks, kb, ks, lane);
        vf[ks] = vnext[ks];
        if (kb + 1 < NB) vnext[ks] = gfrag(vsrc, stride, kb + 1, ks, T, lane);
      }
      f32x16 dk = zero16(), dv = zero16();
#pragma unroll
      for (int qb = kb; qb < NB; ++qb) {
        const f32x16 s = score_block<HS>(Qs, qb, kf, lane);
        const f32x16 dp = score_block<HS>(Gs, qb, vf, lane);
        float p[16], ds[16];
        // the last query block may hold <= 16 rows (T = 80: rows 64..79): register groups 2, 3 (queries 16..31 of the
        // block) are then padding for every lane -- no exponentials, no tile writes, no second MFMA for them
        const bool short_q = (qb == NB - 1) && tail16;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          if (g4 >= 2 && short_q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) p[g4 * 4 + j] = ds[g4 * 4 + j] = 0.f;
            continue;
          }
          const int q0 = qb * 32 + 8 * g4 + 4 * half;
          const float4 l4 = *reinterpret_cast<const float4*>(Ls + q0);
          const float4 d4 = *reinterpret_cast<const float4*>(Ds + q0);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = g4 * 4 + j, q = q0 + j;
            p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], SCALE_LOG2E, -lv[j]));
            if (qb == kb && key > q) p[r] = 0.f;
            ds[r] = p[r] * (dp[r] - dvv[j]);   // the softmax scale is applied once to the finished blocks
          }
          // dS tile: row = key (this lane), 4 consecutive queries 8*g4 + 4*half .. +3
          *reinterpret_cast<uint2*>(dsT + (lane & 31) * DST_PITCH + 8 * g4 + 4 * half) =
              make_uint2(pack2bf(ds[4 * g4], ds[4 * g4 + 1]), pack2bf(ds[4 * g4 + 2], ds[4 * g4 + 3]));
        }
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Gs, qb * 32, lane), pfrag(p), dv, 0, 0, 0);
        if (!short_q) dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Gs, qb * 32 + 16, lane), pfrag(p + 8), dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Qs, qb * 32, lane), pfrag(ds), dk, 0, 0, 0);
        if (!short_q) dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Qs, qb * 32 + 16, lane), pfrag(ds + 8), dk, 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0xc07f);   // the tile writes above are visible to the whole wave
        __builtin_amdgcn_wave_barrier();
        // (K^T fragments re-read per pair: 8 registers fewer across the sweep than keeping them)
        dq[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Ks, kb * 32, lane), dst_frag(dsT, 0, lane), dq[qb], 0, 0, 0);
        dq[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfrag<HS>(Ks, kb * 32 + 16, lane), dst_frag(dsT, 16, lane), dq[qb], 0, 0, 0);
        __builtin_amdgcn_wave_barrier();      // the next pair's tile writes stay behind these reads
      }
#pragma unroll
      for (int r = 0; r < LIVE; ++r) dk[r] *= SCALE;
      __syncthreads();   // every wave is done with its dS tile
      tile_put_grad_rot<HS>(otile, wave, lane, dk, rot);
      tile_put_grad<HS>(otile + ot_bytes<HS>(), wave, lane, dv, false, cos_t, sin_t, 0);
    } else {
      __syncthreads();
    }
    __syncthreads();
    tile_store<HS>(otile, dbase + C, stride, kb * 32, T, heads_here, threadIdx.x);
    tile_store<HS>(otile + ot_bytes<HS>(), dbase + 2 * C, stride, kb * 32, T, heads_here, threadIdx.x);
    __syncthreads();
  }
  RopeRow<HS> qrot[NB];   // the rotation rows of all query blocks: one round trip
#pragma unroll
  for (int qb = 0; qb < NB; ++qb) {
    const int q = qb * 32 + (lane & 31);
    qrot[qb] = rope_row_load<HS>(cos_t, sin_t, q < T ? q : 0, lane);
  }
#pragma unroll
  for (int qb = 0; qb < NB; ++qb) {
    if (active) {
#pragma unroll
      for (int r = 0; r < LIVE; ++r) dq[qb][r] *= SCALE;
      tile_put_grad_rot<HS>(otile, wave, lane, dq[qb], qrot[qb]);
    }
    __syncthreads();
    tile_store<HS>(otile, dbase, stride, qb * 32, T, heads_here, threadIdx.x);
    __syncthreads();
  }
}

template <int HS, int NB>
__global__ __launch_bounds__(256, (HS == 16 && NB <= 3) ? 3 : 2) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ y,
                                                             const bf16_t* __restrict__ dy, const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dqkv, const float* __restrict__ cos_t,
                                                             const float* __restrict__ sin_t, int Tl, int n_head, int quads,
                                                             const int* __restrict__ seq_off) {
  const int b = blockIdx.x / quads, hq = blockIdx.x - b * quads;
  ATT_SEQ(0);
  attn_bwd_fused_body<HS, NB>(qkv, y, dy, lse, dqkv, cos_t, sin_t, Tl, n_head, b, hq, T, row0);
}
// packed rows: per-workgroup block count (see attn_fwd_varlen_kernel), in TWO launches: sequences of NBMIN .. NBMAX blocks per
// launch (a workgroup outside the range leaves at once).  The backward's LDS and registers grow with the block count (three
// staged operands + dQ of every block), so one launch for all lengths runs the short sequences at the long ones' occupancy:
// measured 2.58 ms per step against 2.32 for separate launches (the forward, two operands, is the other way round).
template <int HS, int NBMAX, int NBMIN>
__global__ __launch_bounds__(256, (HS == 16 && NBMAX <= 2) ? 4 : ((HS == 16 && NBMAX <= 3) ? 3 : 2)) void attn_bwd_fused_varlen_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ y,
                                                             const bf16_t* __restrict__ dy, const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dqkv, const float* __restrict__ cos_t,
                                                             const float* __restrict__ sin_t, int Tl, int n_head, int quads,
                                                             const int* __restrict__ seq_off) {
  const int b = blockIdx.x / quads, hq = blockIdx.x - b * quads;
  ATT_SEQ(0);
  const int nb = (T + 31) >> 5;
  if (nb < NBMIN || nb > NBMAX) return;
  if (NBMIN <= 1 && nb == 1) attn_bwd_fused_body<HS, 1>(qkv, y, dy, lse, dqkv, cos_t, sin_t, Tl, n_head, b, hq, T, row0);
  else if (NBMIN <= 2 && NBMAX >= 2 && nb == 2) attn_bwd_fused_body<HS, (NBMAX >= 2 ? 2 : 1)>(qkv, y, dy, lse, dqkv, cos_t, sin_t, Tl, n_head, b, hq, T, row0);
  else if (NBMIN <= 3 && NBMAX >= 3 && nb == 3) attn_bwd_fused_body<HS, (NBMAX >= 3 ? 3 : 1)>(qkv, y, dy, lse, dqkv, cos_t, sin_t, Tl, n_head, b, hq, T, row0);
  else if (NBMAX >= 4 && nb == 4) attn_bwd_fused_body<HS, (NBMAX >= 4 ? 4 : 1)>(qkv, y, dy, lse, dqkv, cos_t, sin_t, Tl, n_head, b, hq, T, row0);
}

template <int HS, int NBMAX, int NBMIN>
static int launch_attn_bwd_fused_varlen_t(const bf16_t* qkv, const bf16_t* y, const bf16_t* dy, const float* lse, bf16_t* dqkv,
                                          const float* cos_t, const float* sin_t, int B, int T, int n_head, hipStream_t s, const int* seq_off) {
  constexpr int Tp = 32 * NBMAX;
  const size_t lds = (size_t)4 * ((size_t)3 * Tp * HS * 2 + (size_t)2 * Tp * 4 + ATT_PW_PAD) + 2 * ot_bytes<HS>();
  static bool attr_v = false;
  if (!attr_v) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_varlen_kernel<HS, NBMAX, NBMIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      coati_set_error("attn_bwd(fused, varlen): hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_v = true;
  }
  const int quads = cdiv(n_head, 4);
  hipLaunchKernelGGL((attn_bwd_fused_varlen_kernel<HS, NBMAX, NBMIN>), dim3(B * quads), dim3(256), lds, s, qkv, y, dy, lse, dqkv, cos_t, sin_t, T, n_head, quads, seq_off);
  COATI_LAUNCH_CHECK("attn_bwd_fused_varlen");
  return COATI_OK;
}

template <int HS, int NB>
static int launch_attn_bwd_fused_t(const bf16_t* qkv, const bf16_t* y, const bf16_t* dy, const float* lse, bf16_t* dqkv,
                                   const float* cos_t, const float* sin_t, int B, int T, int n_head, hipStream_t s, const int* seq_off) {
  constexpr int Tp = 32 * NB;
  const size_t lds = (size_t)4 * ((size_t)3 * Tp * HS * 2 + (size_t)2 * Tp * 4 + ATT_PW_PAD) + 2 * ot_bytes<HS>();
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<HS, NB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      coati_set_error("attn_bwd(fused): hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int quads = cdiv(n_head, 4);
  hipLaunchKernelGGL((attn_bwd_fused_kernel<HS, NB>), dim3(B * quads), dim3(256), lds, s, qkv, y, dy, lse, dqkv, cos_t, sin_t, T,
                     n_head, quads, seq_off);
  COATI_LAUNCH_CHECK("attn_bwd_fused");
  return COATI_OK;
}

template <int HS>
static int launch_attn_bwd_t(const bf16_t* qkv, const bf16_t* y, const bf16_t* dy, const float* lse, float* dscratch,
                             bf16_t* dqkv, const float* cos_t, const float* sin_t, int B, int T, int n_head, hipStream_t s, const int* seq_off) {
  const int Tp = (T + 31) & ~31;
  const size_t lds_dq = (size_t)4 * (2 * Tp * HS * 2 + ATT_PW_PAD) + ot_bytes<HS>();
  const size_t lds_dkv = (size_t)4 * (2 * Tp * HS * 2 + 2 * Tp * 4 + ATT_PW_PAD) + 2 * ot_bytes<HS>();
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<HS>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<HS>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      coati_set_error("attn_bwd: hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int quads = cdiv(n_head, 4);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<HS>, dim3(B * quads), dim3(256), lds_dq, s, qkv, y, dy, lse, dscratch, dqkv, cos_t,
                     sin_t, T, n_head, quads, seq_off);
  COATI_LAUNCH_CHECK("attn_bwd_dq");
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<HS>, dim3(B * quads), dim3(256), lds_dkv, s, qkv, dy, lse, dscratch, dqkv, cos_t,
                     sin_t, T, n_head, quads, seq_off);
  COATI_LAUNCH_CHECK("attn_bwd_dkv");
  return COATI_OK;
}

int launch_attn_bwd(const bf16_t* qkv, const bf16_t* y, const bf16_t* dy, const float* lse, float* dscratch,
                    bf16_t* dqkv, const float* cos_t, const float* sin_t, int B, int T, int n_head, int head_size, hipStream_t s,
                    const int* seq_off, const int* seq_ord) {
  COATI_CHECK_ARG(qkv && y && dy && lse && dscratch && dqkv && cos_t && sin_t, "attn_bwd: null operand");
  COATI_CHECK_SHAPE(B > 0 && T > 0 && T <= 256 && n_head > 0 && (head_size == 16 || head_size == 32),
                    "attn_bwd: unsupported shape B=%d T=%d nh=%d hs=%d", B, T, n_head, head_size);
  static const bool block32 = getenv("COATI_ATTN_BLOCK32") != nullptr;
  if (head_size == 16 && T <= 128 && !block32) return launch_attn16_bwd(qkv, y, dy, lse, dqkv, cos_t, sin_t, B, T, n_head, s, seq_off, seq_ord);
  // T <= 128: the single-sweep kernel (grande: 120 vs 141 us); longer sequences: the two kernels below
  if (T <= 128) {
    const int nb = (T + 31) / 32;
    if (seq_off != nullptr) {   // packed rows: sequences of 1-2 blocks in one launch, of 3-4 blocks in another (ONE launch over 1-3 blocks, round 4: 2.44 vs 2.19 ms per step -- the short sequences lose the 4-workgroups-per-CU register budget)
#define VL(H, HI, LO) return launch_attn_bwd_fused_varlen_t<H, HI, LO>(qkv, y, dy, lse, dqkv, cos_t, sin_t, B, T, n_head, s, seq_off)
#define VL2(H, HI, LO) COATI_TRY((launch_attn_bwd_fused_varlen_t<H, HI, LO>(qkv, y, dy, lse, dqkv, cos_t, sin_t, B, T, n_head, s, seq_off)))
      if (head_size == 16) {
        if (nb == 1) VL(16, 1, 1);
        if (nb == 2) VL(16, 2, 1);
        VL2(16, 2, 1);
        if (nb == 3) VL(16, 3, 3);
        VL(16, 4, 3);
      } else {
        if (nb == 1) VL(32, 1, 1);
        if (nb == 2) VL(32, 2, 1);
        VL2(32, 2, 1);
        if (nb == 3) VL(32, 3, 3);
        VL(32, 4, 3);
      }
#undef VL
#undef VL2
    }
#define FUSED_CASE(H, N) if (head_size == H && nb == N) return launch_attn_bwd_fused_t<H, N>(qkv, y, dy, lse, dqkv, cos_t, sin_t, B, T, n_head, s, seq_off);
    FUSED_CASE(16, 1) FUSED_CASE(16, 2) FUSED_CASE(16, 3) FUSED_CASE(16, 4)
    FUSED_CASE(32, 1) FUSED_CASE(32, 2) FUSED_CASE(32, 3) FUSED_CASE(32, 4)
#undef FUSED_CASE
  }
  return head_size == 16 ? launch_attn_bwd_t<16>(qkv, y, dy, lse, dscratch, dqkv, cos_t, sin_t, B, T, n_head, s, seq_off)
                         : launch_attn_bwd_t<32>(qkv, y, dy, lse, dscratch, dqkv, cos_t, sin_t, B, T, n_head, s, seq_off);
}
