// How fast can a kernel READ cold data from HBM on this chip, as a function of the bytes it keeps in flight?  (Calibrates the "achievable"
// line the cold-operand kernels of the step -- the grouped weight gradient, the backward's reads of the forward's saves -- are held against.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/read_bw_probe.hip -o /tmp/rbp && /tmp/rbp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
// clean = 1: the caches are flushed by READING 1.5 GB (they end up full of clean lines); 0: by rewriting it (full of dirty lines: every cold
// read that follows also pays a write-back)
__global__ void evict(float* j, size_t n, int clean, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (clean) acc += j[i]; else j[i] += 1.0f;
  }
  if (acc == 123.456f) *sink = acc;
}
template <int U>
__global__ __launch_bounds__(256) void rd(const v4u* __restrict__ x, size_t n, unsigned* out) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].w;
  }
  for (; i < n; i += stride) acc += x[i].x;
  if (acc == 0x12345678u) *out = acc;
}
template <int U>
static void run(const v4u* x, size_t n, float* j, size_t nj, unsigned* o, int wgs, int cold, hipEvent_t a, hipEvent_t b) {
  float t = 0;
  for (int it = 0; it < 5; ++it) {
    if (cold) evict<<<2048, 256>>>(j, nj, cold == 2, reinterpret_cast<float*>(o));
    else rd<U><<<wgs, 256>>>(x, n, o);
    hipEventRecord(a);
    rd<U><<<wgs, 256>>>(x, n, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (it) t += ms;
  }
  printf("  %s  %5d workgroups x 256 threads x %d loads of 16 B in flight per thread (%6.1f MB in flight chip-wide if all resident): %6.1f us  %5.2f TB/s\n",
         cold == 2 ? "cold, caches clean" : (cold ? "cold, caches dirty" : "warm              "), wgs, U, wgs * 256.0 * U * 16 / 1e6, t / 4 * 1e3, n * 16.0 / (t / 4 * 1e-3) / 1e12);
}
int main() {
  const size_t nx = 1024u * 1024 * 1024 / 16 / 5 * 1;   // ~ 205 MB: fits the Infinity Cache when warm
  const size_t nj = 1536u * 1024 * 1024 / 4;
  v4u* x; float* j; unsigned* o;
  if (hipMalloc(&x, nx * 16) != hipSuccess || hipMalloc(&j, nj * 4) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
  (void)hipMemset(j, 0, nj * 4); (void)hipMemset(x, 1, nx * 16);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int cold = 2; cold >= 0; --cold) {
    for (int wgs : {256, 1024, 4096}) {
      run<1>(x, nx, j, nj, o, wgs, cold, a, b);
      run<4>(x, nx, j, nj, o, wgs, cold, a, b);
      run<8>(x, nx, j, nj, o, wgs, cold, a, b);
    }
  }
  return 0;
}
