// Does one wave's independent VALU work overlap with its own in-flight MFMAs on gfx950?  Three loops per wave, 1..3 waves per SIMD:
//   M: 32 dependent-chain-free MFMAs (4 accumulators round robin)      V: 32 x NV independent v_pk_fma / v_exp
//   MV: the two interleaved (1 MFMA, then NV VALU instructions)
// prints cycles per iteration; overlap <=> MV ~ max(M, V), no overlap <=> MV ~ M + V.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/overlap_probe.hip -o tools/probes/overlap_probe && tools/probes/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE, int NV>
__global__ __launch_bounds__(768) void k(float* out, unsigned long long* cyc, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  v2f x[8];
  for (int i = 0; i < 8; ++i) x[i] = v2f{threadIdx.x * 0.01f + i, 1.0f + i};
  const v2f c1 = {1.0001f, 0.9999f}, c2 = {0.001f, -0.001f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      if (MODE & 1) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      if (MODE & 2) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if ((v & 7) == 7) { x[v & 7].x = __builtin_amdgcn_exp2f(x[v & 7].x * 0.001f); }
          else x[v & 7] = __builtin_elementwise_fma(x[v & 7], c1, c2);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int NV>
double run(int waves_per_simd) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 768 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 200;
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double tot = 0; for (int i = 0; i < 256; ++i) tot += h[i];
  hipFree(out); hipFree(cyc);
  return tot / 256 / iters;   // memtime ticks (100 MHz) per iteration of 32 MFMA-slots
}
int main() {
  for (int w = 1; w <= 3; ++w) {
    printf("waves/SIMD %d:  M %.1f   V(NV=8) %.1f  MV(8) %.1f |  V(16) %.1f  MV(16) %.1f |  V(24) %.1f  MV(24) %.1f   [memtime ticks per 32 MFMA slots]\n", w,
           run<1, 8>(w), run<2, 8>(w), run<3, 8>(w), run<2, 16>(w), run<3, 16>(w), run<2, 24>(w), run<3, 24>(w));
  }
  return 0;
}
