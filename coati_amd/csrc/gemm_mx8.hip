// MXFP8 GEMM for gfx950 (BASELINE.json configs[4]: "COATI2 d=512 ... fp8 MFMA GEMMs"): C[M,N] = A[M,K] * W[N,K]^T with both
// operands in OCP e4m3 and one E8M0 scale per 32 consecutive k (the OCP Microscaling block format), multiplied by
// v_mfma_scale_f32_32x32x64_f8f6f4 -- the block scales are applied by the matrix core (hardware dequantisation), 64 k per
// instruction at twice the bf16 MFMA rate -- with fp32 accumulation and the SAME fused epilogues as the bf16 kernels
// (gemm_epi.h: bias, RoPE, NewGELU + derivative, x saved derivative, f32 residual).
//
// The pinned maths of configs[4] is the d = 512 transformer block (reference coati/models/simple_coati2/transformer_only.py:43,
// = basic_transformer.py:157-174 at another width); its four Linear layers' forward and input-gradient products are what
// runs here.  There is no fp8 code in the reference: parity is stated against the bf16 path and the oracle (tests/test_gpu_fp8.py).
//
// Layout: operands row-major bytes [rows, K] (K % 128 == 0) + scales [rows, K / 32] bytes (E8M0: value 2^(s - 127)); a 128-k
// tile of a row = 128 B of data + ONE 32-bit word of four scales.  Tiling as the bf16 tiled kernel (gemm.hip): 128 x 128 outputs
// per workgroup, 8 waves x (32 x 64), two LDS buffers + two register stage sets, BK = 128 bytes of k per tile (2 MFMA steps).
// Fragment of the 32x32x64 form, determined with tools/probes/mx_probe.hip (profiles/r03_mx_probe.txt): lane (r = lane & 31,
// h = lane >> 5) holds 32 bytes of row r -- bytes 0..15 = k 16 h .. 16 h + 15 of the step's FIRST scale block, bytes 16..31 =
// k 32 + 16 h .. of the SECOND -- and the scale register of lane (r, h) is the scale of block h (applied to both half-waves'
// bytes of that block).
#include <cstdlib>
#include "gemm_epi.h"

#define MX_BM 128
#define MX_BN 128
#define MX_BKB 128                 // bytes (= fp8 elements) of k per tile
#define MX_PITCHB 144              // bytes per LDS row: 16 consecutive rows hit 16 distinct 16-B slots of the 256-B bank row
#define MX_TILE_BYTES (128 * MX_PITCHB)
#define MX_CPITCH 132
#define MX_LDS_BYTES (4 * MX_TILE_BYTES)   // 73,728 B (>= 128 * 132 * 4 for the staged accumulators)

typedef int mx_i32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int mx_xcd_swizzle(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

struct MxStage { uint4 v[2]; };   // 128 rows x 8 chunks of 16 B over 512 threads
__device__ __forceinline__ void mx_stage_load(MxStage& s, const unsigned char* base, long long ld, int row0, int rows, int k0, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 512 * i, row = c >> 3, kc = (c & 7) * 16;
    const int g = row0 + row, gc = g < rows ? g : rows - 1;
    s.v[i] = *reinterpret_cast<const uint4*>(base + (long long)gc * ld + k0 + kc);
  }
}
__device__ __forceinline__ void mx_stage_store(const MxStage& s, unsigned char* S, int row0, int rows, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 512 * i, row = c >> 3, kc = (c & 7) * 16;
    const unsigned keep = (row0 + row) < rows ? 0xffffffffu : 0u;   // fp8 0x00 = +0
    *reinterpret_cast<uint4*>(S + row * MX_PITCHB + kc) = make_uint4(s.v[i].x & keep, s.v[i].y & keep, s.v[i].z & keep, s.v[i].w & keep);
  }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_mx8_kernel(GemmArgs p, const unsigned* __restrict__ sa_w, const unsigned* __restrict__ sb_w) {
  // p.A / p.B: e4m3 bytes, p.lda / p.ldb in BYTES; sa_w / sb_w: the scale bytes viewed as one 32-bit word per (row, 128-k tile)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + MX_BN - 1) / MX_BN;
  const int wg = mx_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_m = wg / tiles_n, tile_n = wg - tile_m * tiles_n;
  const int m0 = tile_m * MX_BM, n0 = tile_n * MX_BN;
  const unsigned char* A = reinterpret_cast<const unsigned char*>(p.A);
  const unsigned char* B = reinterpret_cast<const unsigned char*>(p.B);
  const int nk = p.K / MX_BKB;

  f32x16 acc[2];
  GemmArgs q = p;
  q.bias = nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    const float b = (p.bias != nullptr) ? p.bias[col < p.N ? col : p.N - 1] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = b;
  }
  // scale words of this lane's rows: A row m0 + 32 wm + r, weight rows n0 + 64 wn + 32 j + r (clamped: the outputs of rows /
  // columns past the end are never stored, and their data is zero)
  const int ra = m0 + wm * 32 + (lane & 31), rac = ra < p.M ? ra : p.M - 1;
  const unsigned* sap = sa_w + (long long)rac * nk;
  const unsigned* sbp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rb = n0 + wn * 64 + j * 32 + (lane & 31), rbc = rb < p.N ? rb : p.N - 1;
    sbp[j] = sb_w + (long long)rbc * nk;
  }
  const int hsh = (lane >> 5) * 8;   // bit offset of this half-wave's scale byte inside a k step's 16 bits

  auto mma_ktile = [&](const unsigned char* As, const unsigned char* Bs, unsigned wa, const unsigned (&wb)[2]) __attribute__((always_inline)) {
    const int r = lane & 31, kb = (lane >> 5) * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned char* ap = As + (wm * 32 + r) * MX_PITCHB + ks * 64 + kb;
      const uint4 a0 = *reinterpret_cast<const uint4*>(ap), a1 = *reinterpret_cast<const uint4*>(ap + 32);
      const mx_i32x8 af = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
      const int sa = (int)((wa >> (16 * ks + hsh)) & 0xffu);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned char* bp = Bs + (wn * 64 + j * 32 + r) * MX_PITCHB + ks * 64 + kb;
        const uint4 b0 = *reinterpret_cast<const uint4*>(bp), b1 = *reinterpret_cast<const uint4*>(bp + 32);
        const mx_i32x8 bf = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
        const int sb = (int)((wb[j] >> (16 * ks + hsh)) & 0xffu);
        // (cbsz, blgp) = (0, 0): both operands e4m3; opsel 0: the scale is byte 0 of the scale register
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af, bf, acc[j], 0, 0, 0, sa, 0, sb);
      }
    }
  };

  auto kofs = [&](int kt) { return (kt < nk ? kt : nk - 1); };
  MxStage sa0, sa1, sb0, sb1;
  mx_stage_load(sa0, A, p.lda, m0, p.M, 0, tid);
  mx_stage_load(sb0, B, p.ldb, n0, p.N, 0, tid);
  mx_stage_load(sa1, A, p.lda, m0, p.M, kofs(1) * MX_BKB, tid);
  mx_stage_load(sb1, B, p.ldb, n0, p.N, kofs(1) * MX_BKB, tid);
  unsigned wa_c = sap[0], wb_c[2] = {sbp[0][0], sbp[1][0]};
  mx_stage_store(sa0, smem, m0, p.M, tid);
  mx_stage_store(sb0, smem + 2 * MX_TILE_BYTES, n0, p.N, tid);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    mx_stage_load(sa0, A, p.lda, m0, p.M, kofs(kt + 2) * MX_BKB, tid);
    mx_stage_load(sb0, B, p.ldb, n0, p.N, kofs(kt + 2) * MX_BKB, tid);
    unsigned wa_n = sap[kofs(kt + 1)], wb_n[2] = {sbp[0][kofs(kt + 1)], sbp[1][kofs(kt + 1)]};
    mma_ktile(smem, smem + 2 * MX_TILE_BYTES, wa_c, wb_c);
    if (kt + 1 < nk) {
      mx_stage_store(sa1, smem + MX_TILE_BYTES, m0, p.M, tid);
      mx_stage_store(sb1, smem + 3 * MX_TILE_BYTES, n0, p.N, tid);
    }
    __syncthreads();
    if (kt + 1 >= nk) break;
    mx_stage_load(sa1, A, p.lda, m0, p.M, kofs(kt + 3) * MX_BKB, tid);
    mx_stage_load(sb1, B, p.ldb, n0, p.N, kofs(kt + 3) * MX_BKB, tid);
    wa_c = sap[kofs(kt + 2)]; wb_c[0] = sbp[0][kofs(kt + 2)]; wb_c[1] = sbp[1][kofs(kt + 2)];
    mma_ktile(smem + MX_TILE_BYTES, smem + 3 * MX_TILE_BYTES, wa_n, wb_n);
    if (kt + 2 < nk) {
      mx_stage_store(sa0, smem, m0, p.M, tid);
      mx_stage_store(sb0, smem + 2 * MX_TILE_BYTES, n0, p.N, tid);
    }
    __syncthreads();
  }

  // accumulators -> LDS (fp32) -> row-contiguous epilogue (as gemm_nt_kernel)
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) Cs[(wm * 32 + frag_row(r, lane)) * MX_CPITCH + wn * 64 + j * 32 + (lane & 31)] = acc[j][r];
  __syncthreads();
  constexpr int TASKS = 4;
  constexpr bool PRE_F32 = (EPI == EPI_RES_F32 || EPI == EPI_ACC_F32);
  EpiPre pre[TASKS];
#pragma unroll
  for (int i = 0; i < TASKS; ++i) {
    const int task = tid + 512 * i, r = task >> 4, cg = task & 15;
    const int row = m0 + r, col0 = n0 + cg * 8;
    pre[i].have = (PRE_F32 || EPI == EPI_MUL_AUX) && row < p.M && col0 + 8 <= p.N;
    if (pre[i].have) {
      if constexpr (PRE_F32) {
        const float* src = (EPI == EPI_RES_F32) ? reinterpret_cast<const float*>(p.aux_in) + (long long)row * p.ld_aux + col0
                                                : reinterpret_cast<const float*>(p.C) + (long long)row * p.ldc + col0;
        pre[i].f0 = *reinterpret_cast<const float4*>(src);
        pre[i].f1 = *reinterpret_cast<const float4*>(src + 4);
      }
      if constexpr (EPI == EPI_MUL_AUX) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(p.aux_in) + (long long)row * p.ld_aux + col0);
        pre[i].h = make_uint4(u.x, u.y, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TASKS; ++i) {
    const int task = tid + 512 * i, r = task >> 4, cg = task & 15;
    float v[8];
    const float4 c0 = *reinterpret_cast<const float4*>(Cs + r * MX_CPITCH + cg * 8);
    const float4 c1 = *reinterpret_cast<const float4*>(Cs + r * MX_CPITCH + cg * 8 + 4);
    v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
    epilogue8<EPI>(q, m0 + r, n0 + cg * 8, v, (m0 + r) < p.M, tile_n, tiles_n, nullptr, &pre[i]);
  }
}

template <int EPI>
static int launch_mx8_t(const GemmArgs& a, const unsigned char* sa, const unsigned char* sb, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_mx8_kernel<EPI>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS_BYTES) != hipSuccess) {
      coati_set_error("gemm_mx8: hipFuncSetAttribute failed");
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int tiles = cdiv(a.M, MX_BM) * cdiv(a.N, MX_BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), MX_LDS_BYTES, s, a, reinterpret_cast<const unsigned*>(sa), reinterpret_cast<const unsigned*>(sb));
  COATI_LAUNCH_CHECK("gemm_mx8");
  return COATI_OK;
}

// a.A / a.B: e4m3 bytes with a.lda / a.ldb in bytes; sa / sb: E8M0 scales [rows, K / 32] (row pitch exactly K / 32 bytes)
int launch_gemm_mx8(const GemmArgs& a, const unsigned char* sa, const unsigned char* sb, int epi, hipStream_t s) {
  COATI_CHECK_ARG(a.A && a.B && a.C && sa && sb, "gemm_mx8: null operand");
  COATI_CHECK_SHAPE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % MX_BKB == 0, "gemm_mx8: K=%d must be a positive multiple of %d", a.K, MX_BKB);
  COATI_CHECK_SHAPE(a.lda % 16 == 0 && a.ldb % 16 == 0, "gemm_mx8: lda / ldb alignment");
  const bool out_f32 = (epi == EPI_F32 || epi == EPI_RES_F32 || epi == EPI_ACC_F32);
  COATI_CHECK_SHAPE(a.ldc % (out_f32 ? 4 : 8) == 0, "gemm_mx8: ldc=%lld alignment", a.ldc);
  COATI_CHECK_SHAPE(((long long)a.M + 128) * a.ldc < (1LL << 32) && ((long long)a.M + 128) * a.ld_aux < (1LL << 32), "gemm_mx8: 32-bit element offsets");
  if (epi == EPI_RES_F32) COATI_CHECK_ARG(a.aux_in && a.ld_aux % 4 == 0, "gemm_mx8: residual missing/misaligned");
  if (epi == EPI_GELU_GRAD) COATI_CHECK_ARG(a.aux_out && a.ld_aux % 8 == 0, "gemm_mx8: aux_out missing");
  if (epi == EPI_MUL_AUX) COATI_CHECK_ARG(a.aux_in && a.ld_aux % 8 == 0, "gemm_mx8: aux_in missing");
  if (epi == EPI_QKV_ROPE) COATI_CHECK_ARG(a.rope_cos && a.rope_sin && a.rope_T > 0 && a.rope_C > 0 && a.N % 16 == 0 && (a.rope_hs == 16 || a.rope_hs == 32), "gemm_mx8: rope operands");
  if (a.q8_out) COATI_CHECK_ARG(a.q8_scales && a.N % 32 == 0 && a.ld_q8 % 8 == 0 && !out_f32 && epi != EPI_QKV_ROPE, "gemm_mx8: fused MXFP8 output needs N %% 32 == 0 and a bf16 epilogue");
  switch (epi) {
    case EPI_BF16: return launch_mx8_t<EPI_BF16>(a, sa, sb, s);
    case EPI_F32: return launch_mx8_t<EPI_F32>(a, sa, sb, s);
    case EPI_RES_F32: return launch_mx8_t<EPI_RES_F32>(a, sa, sb, s);
    case EPI_GELU_GRAD: return launch_mx8_t<EPI_GELU_GRAD>(a, sa, sb, s);
    case EPI_MUL_AUX: return launch_mx8_t<EPI_MUL_AUX>(a, sa, sb, s);
    case EPI_QKV_ROPE: return launch_mx8_t<EPI_QKV_ROPE>(a, sa, sb, s);
    default:
      coati_set_error("gemm_mx8: unsupported epilogue %d", epi);
      return COATI_EARG;
  }
}

// ---- quantisation to MXFP8 (OCP MX: block of 32 along k, shared E8M0 exponent = floor(log2(amax)) - 8, elements e4m3 with
// saturation).  One thread per 8 consecutive elements (16 B of bf16 in, 8 B out), the 4 threads of a block reduce |max|.
template <typename T>
__global__ __launch_bounds__(256) void quant_mx8_kernel(const T* __restrict__ x, long long ldx, unsigned char* __restrict__ q, long long ldq,
                                                        unsigned char* __restrict__ sc, int M, int K) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = K / 8;
  const long long row = g / per_row;
  const int c8 = (int)(g - row * per_row);
  const bool ok = row < M;
  float v[8];
  if (ok) {
    if constexpr (sizeof(T) == 2) {
      unpack8(*reinterpret_cast<const uint4*>(x + row * ldx + c8 * 8), v);
    } else {
      const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c8 * 8), b = *reinterpret_cast<const float4*>(x + row * ldx + c8 * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) am = fmaxf(am, fabsf(v[i]));
  am = fmaxf(am, __shfl_xor(am, 1, 64));
  am = fmaxf(am, __shfl_xor(am, 2, 64));
  // shared exponent: 2^e <= amax < 2^(e+1);  scale = 2^(e - 8)  (e4m3: largest binade 2^8, max 448);  E8M0 byte = e - 8 + 127
  int e = ((__float_as_uint(am) >> 23) & 0xff) - 127;          // floor(log2(am)) for normal numbers; -127 for 0 / denormals
  int se = e - 8;
  if (se < -127) se = -127;
  if (se > 127) se = 127;
  const float inv = __uint_as_float((unsigned)(127 - se) << 23);   // 2^-se  (se = -127 -> 2^254: finite)
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i] * inv, -448.f, 448.f);   // saturate (|x| / scale < 512; e4m3 max = 448)
  unsigned lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  if (ok) {
    *reinterpret_cast<uint2*>(q + row * ldq + c8 * 8) = make_uint2(lo, hi);
    if ((c8 & 3) == 0) sc[row * (K / 32) + (c8 >> 2)] = (unsigned char)(se + 127);
  }
}

int launch_quant_mx8(const void* x, int x_f32, long long ldx, unsigned char* q, long long ldq, unsigned char* scales, int M, int K, hipStream_t s) {
  COATI_CHECK_ARG(x && q && scales, "quant_mx8: null operand");
  COATI_CHECK_SHAPE(M > 0 && K > 0 && K % 32 == 0 && ldx % 8 == 0 && ldq % 8 == 0, "quant_mx8: K=%d must be a multiple of 32 (ldx=%lld ldq=%lld)", K, ldx, ldq);
  const long long n = (long long)M * (K / 8);
  if (x_f32) hipLaunchKernelGGL(quant_mx8_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, s, reinterpret_cast<const float*>(x), ldx, q, ldq, scales, M, K);
  else hipLaunchKernelGGL(quant_mx8_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(x), ldx, q, ldq, scales, M, K);
  COATI_LAUNCH_CHECK("quant_mx8");
  return COATI_OK;
}
