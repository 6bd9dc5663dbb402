// Shared device/host helpers for the gfx950 (CDNA4) kernels of libcoati_hip.so.
// Written for MI355X only: wave = 64 lanes, MFMA 32x32x16 bf16, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define COATI_OK 0
#define COATI_EARG (-1)
#define COATI_ESHAPE (-2)
#define COATI_EHIP (-3)

void coati_set_error(const char* fmt, ...);

#define COATI_CHECK_ARG(cond, ...)        \
  do {                                    \
    if (!(cond)) {                        \
      coati_set_error(__VA_ARGS__);       \
      return COATI_EARG;                  \
    }                                     \
  } while (0)

#define COATI_CHECK_SHAPE(cond, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      coati_set_error(__VA_ARGS__);       \
      return COATI_ESHAPE;                \
    }                                     \
  } while (0)

#define COATI_LAUNCH_CHECK(name)                                                   \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      coati_set_error("%s: HIP launch failed: %s", name, hipGetErrorString(_e));  \
      return COATI_EHIP;                                                           \
    }                                                                              \
  } while (0)

#define COATI_TRY(expr)            \
  do {                             \
    int _rc = (expr);              \
    if (_rc != COATI_OK) return _rc; \
  } while (0)

// ---- bf16 <-> f32: hardware round-to-nearest-even conversion (v_cvt_pk_bf16_f32 on gfx950), the same rounding
// as torch's .bfloat16() -------------------------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// 8 consecutive bf16 <-> 8 floats
__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
  v[0] = bflo(u.x); v[1] = bfhi(u.x); v[2] = bflo(u.y); v[3] = bfhi(u.y);
  v[4] = bflo(u.z); v[5] = bfhi(u.z); v[6] = bflo(u.w); v[7] = bfhi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 u;
  u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
  u.z = pack2bf(v[4], v[5]); u.w = pack2bf(v[6], v[7]);
  return u;
}

// ---- 8-bit fixed point for the saved NewGELU' (the backward's multiplier): NewGELU' lies in [-0.129, 1.129]; code q
// <-> value q / 200 - 0.13 (q = 26 is exactly 0, q = 226 exactly 1), so |error| <= 0.0025 -- one bf16 rounding at 1.0 is
// 0.0039.  Halves the bytes of that tensor (written by the forward MLP, read by the FC2 input-gradient product).
#define COATI_DQ_SCALE 200.0f
#define COATI_DQ_OFF 26.0f
__device__ __forceinline__ uint2 packq8(const float* d) {
  unsigned lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lo = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d[i], COATI_DQ_SCALE, COATI_DQ_OFF), i, lo);         // round-to-nearest, saturating
    hi = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d[4 + i], COATI_DQ_SCALE, COATI_DQ_OFF), i, hi);
  }
  return make_uint2(lo, hi);
}
__device__ __forceinline__ unsigned char q8_one(float d) { return (unsigned char)(__builtin_amdgcn_cvt_pk_u8_f32(fmaf(d, COATI_DQ_SCALE, COATI_DQ_OFF), 0, 0u) & 0xffu); }
__device__ __forceinline__ float dq8_one(unsigned q) { return fmaf((float)q, 1.0f / COATI_DQ_SCALE, -COATI_DQ_OFF / COATI_DQ_SCALE); }
__device__ __forceinline__ void unpackq8(const uint2& u, float* v) {
  const float sc = 1.0f / COATI_DQ_SCALE, of = -COATI_DQ_OFF / COATI_DQ_SCALE;   // (the byte extracts compile to v_cvt_f32_ubyte0..3)
  v[0] = fmaf((float)((u.x >> 0) & 0xffu), sc, of); v[1] = fmaf((float)((u.x >> 8) & 0xffu), sc, of);
  v[2] = fmaf((float)((u.x >> 16) & 0xffu), sc, of); v[3] = fmaf((float)((u.x >> 24) & 0xffu), sc, of);
  v[4] = fmaf((float)((u.y >> 0) & 0xffu), sc, of); v[5] = fmaf((float)((u.y >> 8) & 0xffu), sc, of);
  v[6] = fmaf((float)((u.y >> 16) & 0xffu), sc, of); v[7] = fmaf((float)((u.y >> 24) & 0xffu), sc, of);
}

// ---- activations (fp32 math) --------------------------------------------------------------------
// The hardware transcendental unit does the exponentials (v_exp_f32 = 2^x) and the reciprocals (v_rcp_f32, 1 ulp): an
// IEEE `a / b` expands to ~10 VALU instructions, which made the GELU epilogues VALU-bound.  sigmoid saturates cleanly:
// 2^(+big) = inf -> rcp = 0 ; 2^(-big) = 0 -> 1.
#define COATI_LOG2E 1.4426950408889634f
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + __builtin_amdgcn_exp2f(-COATI_LOG2E * x)); }
// NewGELU, tanh form (reference basic_transformer.py:12-28): 0.5 x (1 + tanh(u)) = x * sigmoid(2u),
// u = sqrt(2/pi) (x + 0.044715 x^3)
__device__ __forceinline__ float gelu_f(float x) {
  const float k2 = 2.0f * 0.7978845608028654f;
  const float z = k2 * fmaf(0.044715f * x * x, x, x);
  return x * sigmoid_f(z);
}
// d/dx [x s(z)] = s + x s (1 - s) z',  z' = 2 sqrt(2/pi) (1 + 3 * 0.044715 x^2)
__device__ __forceinline__ float dgelu_f(float x) {
  const float k2 = 2.0f * 0.7978845608028654f;
  const float x2 = x * x;
  const float z = k2 * fmaf(0.044715f * x2, x, x);
  const float s = sigmoid_f(z);
  const float zp = k2 * fmaf(3.0f * 0.044715f, x2, 1.0f);
  return fmaf(x * s * (1.0f - s), zp, s);
}
// both at once (one sigmoid): the forward MLP stores NewGELU' next to NewGELU, so the backward is a plain multiply.
// Two elements per call: everything but the two transcendentals is packed fp32 (v_pk_mul / v_pk_fma / v_pk_add, two
// lanes' worth per issue slot) -- the activation epilogues are VALU-bound.
typedef float coati_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_and_grad_f2(coati_v2f x, coati_v2f& h, coati_v2f& d) {
  const float k2 = 2.0f * 0.7978845608028654f;
  const coati_v2f one = {1.0f, 1.0f};
  const coati_v2f x2 = x * x;
  const coati_v2f t = x * __builtin_elementwise_fma(coati_v2f{0.044715f, 0.044715f}, x2, one);    // x + 0.044715 x^3
  // (folding -log2(e) k2 into the polynomial saves one slot per pair and is the same function to 1e-7 -- and moved the gradient
  //  norms at step 10 of the 20-step reference curve from 6e-2 to 2.5e-1 of deviation (tests/test_gpu_grande.py: the late gradients
  //  are residuals of cancelling terms); the arithmetic below is the one the mid-curve pins were taken with)
  const coati_v2f a = t * coati_v2f{-COATI_LOG2E * k2, -COATI_LOG2E * k2};                        // -log2(e) z
  const coati_v2f den = coati_v2f{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + one;
  const coati_v2f s = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  const coati_v2f zp = __builtin_elementwise_fma(coati_v2f{3.0f * 0.044715f * k2, 3.0f * 0.044715f * k2}, x2, coati_v2f{k2, k2});
  h = x * s;
  d = __builtin_elementwise_fma(h * (one - s), zp, s);
}
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float dsilu_f(float x) {
  float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}

// ---- wave reductions (64 lanes) -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// value + / max the value of lane ^ 32 (the other half-wave): v_permlane32_swap_b32 (gfx950: exchanges the upper half of its first
// operand with the lower half of its second) instead of __shfl_xor's ds_bpermute trip through the LDS crossbar.  Same results bit for bit.
typedef unsigned coati_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float half_xchg_sum(float v) {
  const coati_v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float half_xchg_max(float v) {
  const coati_v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
