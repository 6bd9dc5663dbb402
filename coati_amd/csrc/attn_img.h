// Shared by attention.hip (32-row blocks: head size 32, T > 128) and attention16.hip (head size 16 on 16-row blocks): the row-major
// [T][HS] bf16 operand images of one (sequence, 4 heads) workgroup in LDS -- bank swizzle, cooperative staging with every global
// load issued before the first LDS write -- and the sequence addressing of the padded / packed row layouts.
#pragma once
#include "kernels.h"

#define LOG2E 1.4426950408889634f
// 1/sqrt(HS) and its product with log2(e): softmax exponentials go straight to v_exp_f32 (2^x)
template <int HS> __device__ __forceinline__ constexpr float att_scale() { return HS == 16 ? 0.25f : 0.17677669529663687f; }
template <int HS> __device__ __forceinline__ constexpr float att_scale_log2e() { return HS == 16 ? 0.36067376022224085f : 0.25503486588225905f; }
typedef short v4s16a __attribute__((ext_vector_type(4)));
typedef short v8s16a __attribute__((ext_vector_type(8)));

// LDS bank swizzle of the row-major [T][HS] images (round 2; SQ_LDS_BANK_CONFLICT was 41 % / 50 % of SQ_LDS_IDX_ACTIVE in the
// backward / forward kernel).  A row is 32 B (HS = 16) or 64 B (HS = 32), so 8 (4) consecutive rows span the 64 banks and
// rows 8 (4) apart start on the same bank: the 16 lanes of one ds_read_b128 cycle (16 different rows, same 16-B chunk) hit
// every bank group twice (four times).  The 16-B chunk c of row t is therefore stored at chunk position
// c ^ ((t / rows_per_span) & (chunks_per_row - 1)): rows that used to collide now sit in different chunks.  Applied by the
// staging writes and by both fragment reads (the transpose read takes per-lane addresses).
template <int HS>
__device__ __forceinline__ int img_chunk(int t, int c) {
  constexpr int CPH = HS / 8, SPAN = 256 / (HS * 2);   // chunks per row; rows per 256-B bank span
  return c ^ ((t / SPAN) & (CPH - 1));
}
// The per-head image sets of a workgroup are ATT_PW_PAD bytes further apart than their size: with a multiple of 256 B
// between them, the 8 lanes of one ds_write_b128 cycle of the staging (same row, 4 heads x 2 chunks) hit the same banks 4 times.
#define ATT_PW_PAD 64

// Cooperative staging of one operand for the 4 heads of a workgroup: rows t < T, 4 * HS contiguous bf16 per row (128 /
// 256 B); HS/2 consecutive threads fetch one row -> fully coalesced 16-B loads.  Chunk c of a row lands in head
// c / (HS/8)'s row-major image at dims (c % (HS/8)) * 8.  Rows [T, Tp) are zero-filled.  q and k arrive already rotated
// (the QKV GEMM applies RoPE in its epilogue).
template <int HS>
__device__ __forceinline__ void stage4(const bf16_t* src, long long stride, int T, int Tp, unsigned char* smem,
                                       size_t per_wave_bytes, int image, int heads_here, int tid) {
  constexpr int CPR = HS / 2, CPH = HS / 8;   // 16-B chunks per row / per head
  for (int task = tid; task < Tp * CPR; task += 256) {
    const int t = task / CPR, c = task - t * CPR, w = c / CPH;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t < T && w < heads_here) v = *reinterpret_cast<const uint4*>(src + (long long)t * stride + c * 8);
    bf16_t* img = reinterpret_cast<bf16_t*>(smem + (size_t)w * per_wave_bytes) + (size_t)image * Tp * HS;
    *reinterpret_cast<uint4*>(img + t * HS + img_chunk<HS>(t, c - w * CPH) * 8) = v;
  }
}

// The same with every load of the operand issued before the first LDS write (compile-time trip count: NT tasks per
// thread): out-of-range rows / heads read a clamped address and are masked afterwards, so there is no branch between the
// loads and one global round trip covers the whole operand (the looped form above costs one round trip per task when the
// trip count is a runtime value).
template <int HS, int TP>
struct Stage4Regs { uint4 v[(TP * (HS / 2) + 255) / 256]; };
template <int HS, int TP>
__device__ __forceinline__ void stage4_load(Stage4Regs<HS, TP>& r, const bf16_t* src, long long stride, int T, int heads_here, int tid) {
  constexpr int CPR = HS / 2, CPH = HS / 8, NT = (TP * CPR + 255) / 256;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int task = tid + 256 * i, t = task / CPR, c = task - t * CPR;
    const int tc = t < T ? t : T - 1, cc = (c / CPH) < heads_here ? c : 0;
    r.v[i] = *reinterpret_cast<const uint4*>(src + (long long)tc * stride + cc * 8);
  }
}
template <int HS, int TP>
__device__ __forceinline__ void stage4_store(const Stage4Regs<HS, TP>& r, int T, unsigned char* smem, size_t per_wave_bytes, int image,
                                             int heads_here, int tid) {
  constexpr int CPR = HS / 2, CPH = HS / 8, NT = (TP * CPR + 255) / 256;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int task = tid + 256 * i, t = task / CPR, c = task - t * CPR, w = c / CPH;
    if (TP * CPR % 256 != 0 && task >= TP * CPR) break;
    const unsigned keep = (t < T && w < heads_here) ? 0xffffffffu : 0u;   // mask, not select: the load stays unpredicated
    const uint4 v = make_uint4(r.v[i].x & keep, r.v[i].y & keep, r.v[i].z & keep, r.v[i].w & keep);
    bf16_t* img = reinterpret_cast<bf16_t*>(smem + (size_t)w * per_wave_bytes) + (size_t)image * TP * HS;
    *reinterpret_cast<uint4*>(img + t * HS + img_chunk<HS>(t, c - w * CPH) * 8) = v;
  }
}


// Sequence b of a kernel: rows [row0, row0 + T) of the token-major matrices.  Padded layout: row0 = b * Tl, T = Tl.  Packed rows
// (seq_off != null, embed.hip launch_seq_pack): row0 = seq_off[b], T = seq_off[b + 1] - seq_off[b] <= Tl; lse / D keep the
// padded pitch Tl.  The kernels with compile-time block counts then run through the *_varlen_kernel dispatchers below: a
// workgroup executes the body compiled for its own sequence's number of 32-row blocks.
#define ATT_SEQ(NB_)                                                                   \
  int T = Tl;                                                                          \
  long long row0 = (long long)b * Tl;                                                  \
  if (seq_off != nullptr) {                                                            \
    const int o_ = seq_off[b];                                                         \
    T = seq_off[b + 1] - o_;                                                           \
    row0 = o_;                                                                         \
    if (T <= 0) return;                                                                \
  }

