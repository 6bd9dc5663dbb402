// Do MFMAs of one wave and VALU work of ANOTHER wave on the same SIMD overlap on gfx950?  Workgroups of 8 waves (2 per SIMD:
// waves w and w + 4 share a SIMD, tools/probes/hwid_probe.hip); mode per half: 0 = idle, 1 = MFMA loop, 2 = VALU loop (packed fma +
// exp2 + rcp, the activation epilogue's mix), 3 = LDS read loop.  Wall time of the launch (HIP events), one workgroup per CU:
//   A alone, B alone, A and B together.  overlap <=> together ~ max(A, B); no overlap <=> together ~ A + B.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/coissue_probe.hip -o tools/probes/coissue_probe && tools/probes/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(float* out, int modeA, int modeB, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 0.001f;
  __syncthreads();
  const int mode = wave < 4 ? modeA : modeB;
  float s = 0.f;
  if (mode == 1) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 32; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  } else if (mode == 2) {
    v2f x[8];
    for (int i = 0; i < 8; ++i) x[i] = v2f{lane * 0.01f + i, 1.0f + i};
    const v2f c1 = {1.0001f, 0.9999f}, c2 = {0.001f, -0.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int v = 0; v < 128; ++v) {        // 128 VALU slots: 12 packed fma : 2 exp2 : 2 rcp per 16
        const int r = v & 15, j = v & 7;
        if (r == 6 || r == 14) x[j].x = __builtin_amdgcn_exp2f(x[j].x * 0.001f);
        else if (r == 7 || r == 15) x[j].y = __builtin_amdgcn_rcpf(x[j].y + 2.0f);
        else x[j] = __builtin_elementwise_fma(x[j], c1, c2);
      }
    }
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  } else if (mode == 4) {      // plain (unpacked) f32 fma: 128 slots
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = lane * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int v = 0; v < 128; ++v) x[v & 15] = fmaf(x[v & 15], 1.0001f, 0.001f);
    }
    for (int i = 0; i < 16; ++i) s += x[i];
  } else if (mode == 5) {      // transcendentals only: 32 slots
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = lane * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int v = 0; v < 32; ++v) x[v & 15] = __builtin_amdgcn_exp2f(x[v & 15]);
    }
    for (int i = 0; i < 16; ++i) s += x[i];
  } else if (mode == 6) {      // packed f32 fma only: 128 slots
    v2f x[8];
    for (int i = 0; i < 8; ++i) x[i] = v2f{lane * 0.01f + i, 1.0f + i};
    const v2f c1 = {1.0001f, 0.9999f}, c2 = {0.001f, -0.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int v = 0; v < 128; ++v) x[v & 7] = __builtin_elementwise_fma(x[v & 7], c1, c2);
    }
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  } else if (mode == 3) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int v = 0; v < 32; ++v) t += *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + v * 256 + it) & 8188));
    }
    s = t[0] + t[1] + t[2] + t[3];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
static float run(float* out, int a, int b, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, a, b, iters);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, a, b, iters);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  const char* nm[7] = {"idle", "MFMA", "VALU", "LDS", "fma", "exp2", "pkfma"};
  const int pairs[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 2}, {1, 3}, {2, 3}, {4, 0}, {5, 0}, {6, 0}, {4, 4}, {5, 5}, {6, 6}, {1, 4}, {1, 5}, {1, 6}, {4, 3}};
  for (auto& p : pairs) printf("waves 0-3: %-4s  waves 4-7: %-4s  %8.1f us\n", nm[p[0]], nm[p[1]], run(out, p[0], p[1], iters));
  return 0;
}
