// Fused GEMM epilogues on "8 consecutive columns of one output row" (shared by gemm.hip and gemm_rb.hip).
#pragma once
#include "kernels.h"

// C/D fragment of v_mfma_f32_32x32x16: lane holds col = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- epilogue on 8 consecutive columns of one row ----------------------------------------------------
// register-resident pre-loaded extra operand of one task (tiled kernel): see `staged` below for the meaning per epilogue
struct EpiPre {
  float4 f0, f1;
  uint4 h;
  bool have;
};

// CE_LANES: lanes that share one output row of the tile (EPI_CE_PARTIAL reduces across them): 16 for the 128-column
// tiles of the tiled kernel, 8 for the 64-column tiles of the row-block kernel
// ROPE_HS: 0 = head size read from p.rope_hs at run time; 16 = compiled for head size 16 (row-block kernel: no second DPP
// permutation, no selects)
template <int EPI, int ROPE_PARTNER = 1, int CE_LANES = 16, int ROPE_HS = 0>
__device__ __forceinline__ void epilogue8(const GemmArgs& p, int row, int col0, float (&v)[8], bool rowok,
                                          int tile_n, int tiles_n, const void* staged = nullptr, const EpiPre* pre = nullptr) {
  // staged: optional operand the caller pre-staged (in LDS) so that the epilogue issues no global load for it:
  //   EPI_QKV_ROPE       -> float[16] = [8 cos | 8 sin] of this row's token position (instead of the global tables)
  //   EPI_DGELU / DSILU  -> uint4 = the 8 saved pre-activations aux_in[row, col0..col0+7]
  //   EPI_MUL_AUX        -> uint2 = the 8 saved 8-bit NewGELU' codes aux_in[row, col0..col0+7]
  //   EPI_EDGE_DPRE      -> float[8 + ...]: staged[e] = w1c of column col0+e, staged[64 + e] = b1 of column col0+e
  //   EPI_RES_F32        -> float4[2] = aux_in[row, col0..col0+7];   EPI_ACC_F32 -> float4[2] = C[row, col0..col0+7]
  const float* rope_row = (EPI == EPI_QKV_ROPE) ? reinterpret_cast<const float*>(staged) : nullptr;
  const int N = p.N;
  if (p.bias != nullptr) {
    if (col0 + 8 <= N) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col0), b1 = *reinterpret_cast<const float4*>(p.bias + col0 + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    } else {
      for (int e = 0; e < 8; ++e)
        if (col0 + e < N) v[e] += p.bias[col0 + e];
    }
  }
  if (EPI == EPI_CE_PARTIAL) {
    // every lane of the CE_LANES-lane group that shares this row takes part in the shuffles
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (col0 + e < N) mx = fmaxf(mx, v[e]);
#pragma unroll
    for (int o = 1; o < CE_LANES; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sm = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (col0 + e < N) sm += __expf(v[e] - mx);
#pragma unroll
    for (int o = 1; o < CE_LANES; o <<= 1) sm += __shfl_xor(sm, o, 64);
    if (rowok && (threadIdx.x & (CE_LANES - 1)) == 0) p.partial[(long long)row * tiles_n + tile_n] = make_float2(mx, sm);
    return;
  }
  if (!rowok && EPI != EPI_QKV_ROPE) return;
  const int nst = p.n_store > N ? p.n_store : N;
  if (col0 >= nst) return;
  // element offsets fit 32 bits (launch_gemm_nt refuses M * ld >= 2^32): one 32-bit multiply per operand instead of a 64-bit
  // one (three quarter-rate instructions) -- the epilogues are issue-bound (tools/probes/rb_trace.py)
  const long long off = (long long)((unsigned)row * (unsigned)p.ldc + (unsigned)col0);
  const long long aoff = (long long)((unsigned)row * (unsigned)p.ld_aux + (unsigned)col0);
  const bool full = (col0 + 8 <= N);

  if (EPI == EPI_F32 || EPI == EPI_RES_F32 || EPI == EPI_ACC_F32) {
    float* C = reinterpret_cast<float*>(p.C);
    if (full) {
      float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
      if (EPI == EPI_RES_F32) {
        const float* R = reinterpret_cast<const float*>(p.aux_in);
        float4 r0, r1;
        if (pre && pre->have) { r0 = pre->f0; r1 = pre->f1; }
        else { r0 = *reinterpret_cast<const float4*>(R + aoff); r1 = *reinterpret_cast<const float4*>(R + aoff + 4); }
        o0.x += r0.x; o0.y += r0.y; o0.z += r0.z; o0.w += r0.w;
        o1.x += r1.x; o1.y += r1.y; o1.z += r1.z; o1.w += r1.w;
      }
      if (EPI == EPI_ACC_F32) {
        float4 r0, r1;
        if (pre && pre->have) { r0 = pre->f0; r1 = pre->f1; }
        else { r0 = *reinterpret_cast<const float4*>(C + off); r1 = *reinterpret_cast<const float4*>(C + off + 4); }
        o0.x += r0.x; o0.y += r0.y; o0.z += r0.z; o0.w += r0.w;
        o1.x += r1.x; o1.y += r1.y; o1.z += r1.z; o1.w += r1.w;
      }
      *reinterpret_cast<float4*>(C + off) = o0;
      *reinterpret_cast<float4*>(C + off + 4) = o1;
    } else {
      for (int e = 0; e < 8; ++e) {
        if (col0 + e >= nst) break;
        float o = (col0 + e < N) ? v[e] : 0.f;
        if (col0 + e < N) {
          if (EPI == EPI_RES_F32) o += reinterpret_cast<const float*>(p.aux_in)[aoff + e];
          if (EPI == EPI_ACC_F32) o += C[off + e];
        }
        C[off + e] = o;
      }
    }
    return;
  }

  // bf16 outputs
  float o[8];
  if (EPI == EPI_BF16) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e];
  } else if (EPI == EPI_QKV_ROPE) {
    // This lane holds 8 consecutive dims of one head; the RoPE partner (d +- hs/2) sits in lane ^ 1 (head size 16) or
    // lane ^ 2 (head size 32): a DPP quad permutation (one VALU op), not a ds_bpermute through the LDS pipe.
    // RotaryEmbedding.rotary_embed (basic_transformer.py:83-100): y_i = x_i c_i - x_{i+h} s_i ; y_{i+h} = x_{i+h} c_i + x_i s_i
    const bool hs32 = (ROPE_HS == 16) ? false : p.rope_hs == 32;
    const bool hi_half = (col0 & (hs32 ? 16 : 8)) != 0;
    const bool rot = col0 < 2 * p.rope_C;
    if (ROPE_HS == 16 && !rot) {
      // v block: nothing to rotate (wave-uniform for the row-block kernel: its 64-column tiles never straddle 2C)
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[e];
    } else {
      const int t = (rope_row && ROPE_HS == 16) ? 0 : (p.rope_pos ? *p.rope_pos : (p.rope_row_t ? p.rope_row_t[row < p.M ? row : p.M - 1] : row % p.rope_T));
      const int tab = hs32 ? t * 32 + (col0 & 8) : t * 16;   // tables are [n_seq, hs] with entries i and i + hs/2 equal
      const float sgn = hi_half ? 1.0f : -1.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int vi = __builtin_bit_cast(int, v[e]);
        float other;
        if (ROPE_PARTNER != 1) {
          other = __shfl_xor(v[e], hs32 ? 2 * ROPE_PARTNER : ROPE_PARTNER, 64);
        } else if (ROPE_HS == 16) {
          other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0xB1, 0xF, 0xF, true));   // lane ^ 1
        } else {
          const float o1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0xB1, 0xF, 0xF, true));   // lane ^ 1
          const float o2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x4E, 0xF, 0xF, true));   // lane ^ 2
          other = hs32 ? o2 : o1;
        }
        const float c = rope_row ? rope_row[e] : p.rope_cos[tab + e];
        const float s_ = rope_row ? rope_row[8 + e] : p.rope_sin[tab + e];
        const float r = fmaf(sgn * other, s_, v[e] * c);
        o[e] = (ROPE_HS == 16 || rot) ? r : v[e];
      }
    }
  } else if (EPI == EPI_GELU_GRAD) {
    unsigned char* X = reinterpret_cast<unsigned char*>(p.aux_out);   // NewGELU' as 8-bit fixed point (common.h, packq8)
    float d[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      coati_v2f hh, dd;
      gelu_and_grad_f2(coati_v2f{v[e], v[e + 1]}, hh, dd);
      o[e] = hh.x; o[e + 1] = hh.y;
      d[e] = dd.x; d[e + 1] = dd.y;
    }
    if (full) {
      *reinterpret_cast<uint2*>(X + aoff) = packq8(d);
    } else {
      for (int e = 0; e < 8 && col0 + e < N; ++e) X[aoff + e] = q8_one(d[e]);
    }
  } else if (EPI == EPI_GELU || EPI == EPI_SILU) {
    bf16_t* X = reinterpret_cast<bf16_t*>(p.aux_out);
    if (full) {
      *reinterpret_cast<uint4*>(X + aoff) = pack8(v);
    } else {
      for (int e = 0; e < 8 && col0 + e < N; ++e) X[aoff + e] = f2bf(v[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (EPI == EPI_GELU) ? gelu_f(v[e]) : silu_f(v[e]);
  } else if (EPI == EPI_MUL_AUX) {
    const unsigned char* X = reinterpret_cast<const unsigned char*>(p.aux_in);   // 8-bit fixed point written by EPI_GELU_GRAD
    float x[8];
    if (pre && pre->have) {
      unpackq8(make_uint2(pre->h.x, pre->h.y), x);
    } else if (staged) {
      unpackq8(*reinterpret_cast<const uint2*>(staged), x);
    } else if (full) {
      unpackq8(*reinterpret_cast<const uint2*>(X + aoff), x);
    } else {
      for (int e = 0; e < 8; ++e) x[e] = (col0 + e < N) ? dq8_one(X[aoff + e]) : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e] * x[e];
  } else if (EPI == EPI_DGELU || EPI == EPI_DSILU) {
    const bf16_t* X = reinterpret_cast<const bf16_t*>(p.aux_in);
    float x[8];
    if (pre && pre->have) {
      unpack8(pre->h, x);
    } else if (staged) {
      unpack8(*reinterpret_cast<const uint4*>(staged), x);
    } else if (full) {
      unpack8(*reinterpret_cast<const uint4*>(X + aoff), x);
    } else {
      for (int e = 0; e < 8; ++e) x[e] = (col0 + e < N) ? bf2f(X[aoff + e]) : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e] * ((EPI == EPI_DGELU) ? dgelu_f(x[e]) : dsilu_f(x[e]));
  } else if (EPI == EPI_CE_BWD) {
    const long long tgt = p.target[row];
    const float cnt = p.scal[1];
    const float inv = (tgt >= 0 && cnt > 0.f) ? 1.0f / cnt : 0.f;
    const float l = p.lse[row];
    // (softmax(v) - onehot) / count in few VALU slots (they add to the tile time): the target's position relative to this lane's 8
    // columns as ONE 32-bit value (anything outside 0..7 never matches).  exp(v - lse) keeps the subtraction FIRST: folded into
    // exp2(fma(v, log2e, -lse log2e)) the argument carries the rounding of two numbers of size ~ 30, i.e. 1e-6 of absolute error
    // per row, which is the size of (p_target - 1) on well-fitted tokens -- the mid-curve gradient norms of the 20-step reference
    // curve moved by 30 % (tests/test_gpu_grande.py) when this was tried.
    const long long rel = tgt - (long long)col0;
    const int hit = (rel >= 0 && rel < 8) ? (int)rel : -1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pr = __expf(v[e] - l);
      o[e] = (pr - (hit == e ? 1.0f : 0.0f)) * inv;
    }
  } else if (EPI == EPI_EDGE_DPRE) {
    const int A = p.natom, H = p.H;
    int bj, bk;
    if (p.e_bj) {   // compacted edge list
      const int rr = rowok ? row : 0;
      bj = p.e_bj[rr];
      bk = p.e_bk[rr];
    } else {        // dense grid
      bj = row / A;
      bk = (row / (A * A)) * A + (row - bj * A);
    }
    const bf16_t* Pa = p.P + (long long)bj * p.ldp + col0;
    const bf16_t* Pb = p.P + (long long)bk * p.ldp + H + col0;
    const float d2 = p.d2[rowok ? row : 0];
    float pa[8], pb[8];
    if (full) {
      unpack8(*reinterpret_cast<const uint4*>(Pa), pa);
      unpack8(*reinterpret_cast<const uint4*>(Pb), pb);
    } else {
      for (int e = 0; e < 8; ++e) {
        pa[e] = (col0 + e < N) ? bf2f(Pa[e]) : 0.f;
        pb[e] = (col0 + e < N) ? bf2f(Pb[e]) : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col0 + e;
      float pre = 0.f;
      if (c < N) {
        const float* st = reinterpret_cast<const float*>(staged);
        const float wc = st ? st[e] : p.w1c[(long long)c * p.w1c_stride];
        const float bc = st ? st[64 + e] : p.b1[c];
        pre = pa[e] + pb[e] + d2 * wc + bc;
      }
      o[e] = v[e] * dsilu_f(pre);
    }
  }
  if (!rowok) return;
  if (p.q8_out != nullptr) {
    // fp8 mode (gemm_mx8.hip): the consumer of this bf16 tensor is another MXFP8 product -- emit its quantised copy here instead
    // of a separate pass over the tensor.  One scale block = 32 columns = the 4 consecutive lanes that hold them (N % 32 == 0,
    // all rows of a quad alike: the launcher checks); same arithmetic as quant_mx8_kernel.
    float am = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(o[e]));
    am = fmaxf(am, __shfl_xor(am, 1, 64));
    am = fmaxf(am, __shfl_xor(am, 2, 64));
    int se = (int)((__float_as_uint(am) >> 23) & 0xff) - 127 - 8;
    se = se < -127 ? -127 : (se > 127 ? 127 : se);
    const float inv = __uint_as_float((unsigned)(127 - se) << 23);
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = __builtin_amdgcn_fmed3f(o[e] * inv, -448.f, 448.f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(qv[0], qv[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(qv[2], qv[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(qv[4], qv[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(qv[6], qv[7], hi, true);
    *reinterpret_cast<uint2*>(p.q8_out + (long long)row * p.ld_q8 + col0) = make_uint2(lo, hi);
    if ((col0 & 31) == 0) p.q8_scales[(long long)row * (N >> 5) + (col0 >> 5)] = (unsigned char)(se + 127);
  }
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
  if (col0 + 8 <= nst) {
    if (!full) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (col0 + e >= N) o[e] = 0.f;
    }
    *reinterpret_cast<uint4*>(C + off) = pack8(o);
  } else {
    for (int e = 0; e < 8 && col0 + e < nst; ++e) C[off + e] = (col0 + e < N) ? f2bf(o[e]) : (bf16_t)0;
  }
}

