// Where does a weight-tile staging iteration spend its time?  s_memtime stamps around: load issue, vmcnt(0), ds_write, barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short bf16_t;
#define PITCH 264
template <int NTHREADS, int MODE>
__global__ __launch_bounds__(NTHREADS) void k(const bf16_t* Bw, unsigned long long* stamps, uint4* sink, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Bs = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x;
  constexpr int PER = (2048 + NTHREADS - 1) / NTHREADS;
  uint4 st[PER];
  uint4 accx = make_uint4(0, 0, 0, 0);
  unsigned long long t[5];
  for (int j = 0; j < ntiles; ++j) {
    t[0] = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NTHREADS * i, cc = c < 2048 ? c : 2047;
      st[i] = *reinterpret_cast<const uint4*>(Bw + (long long)(j * 64 + (cc >> 5)) * 256 + (cc & 31) * 8);
    }
    t[1] = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
    t[2] = __builtin_readcyclecounter();
    if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int c = tid + NTHREADS * i;
        if (MODE == 4) { if (c < 2048) *reinterpret_cast<uint4*>(Bs + (j & 1) * 64 * PITCH + c * 8) = st[i]; }
        else if (MODE != 3 && c < 2048) *reinterpret_cast<uint4*>(Bs + ((j & 1) * 64 + (c >> 5)) * PITCH + (c & 31) * 8) = st[i];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      t[3] = __builtin_readcyclecounter();
      if (MODE != 2) __syncthreads();
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) { accx.x ^= st[i].x; accx.y ^= st[i].y; accx.z ^= st[i].z; accx.w ^= st[i].w; }
      t[3] = __builtin_readcyclecounter();
    }
    t[4] = __builtin_readcyclecounter();
    if (blockIdx.x == 17 && tid == 0 && j < 16)
      for (int q = 0; q < 5; ++q) stamps[j * 5 + q] = t[q];
  }
  if (accx.x == 0x12345 && accx.y == 0x777) sink[0] = accx;
  if (MODE != 1 && Bs[tid] == 0x1234 && tid == 9999) sink[1] = accx;
}
template <int NT, int MODE>
void run(const bf16_t* B, unsigned long long* st, uint4* sink, const char* name) {
  auto kern = k<NT, MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 150 * 1024, 0, B, st, sink, 16);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 150 * 1024, 0, B, st, sink, 16);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[80];
  hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s %6.1f us/launch | iter cycles (issue, wait, write, barrier): ", name, ms * 50.f);
  for (int j = 1; j < 6; ++j) printf("[%llu %llu %llu %llu] ", h[j*5+1]-h[j*5], h[j*5+2]-h[j*5+1], h[j*5+3]-h[j*5+2], h[j*5+4]-h[j*5+3]);
  printf(" iter-to-iter %llu\n", h[10] - h[5]);
}
int main() {
  bf16_t* B; unsigned long long* st; uint4* sink;
  hipMalloc(&B, 1024 * 256 * 2); hipMalloc(&st, 4096); hipMalloc(&sink, 64);
  hipMemset(B, 0x3c, 1024 * 256 * 2);
  run<640, 0>(B, st, sink, "640 thr, lds+barrier");
  run<640, 1>(B, st, sink, "640 thr, no lds");
  run<640, 2>(B, st, sink, "640 thr, lds, no barrier");
  run<640, 3>(B, st, sink, "640 thr, barrier, no lds");
  run<640, 4>(B, st, sink, "640 thr, lds linear+barrier");
  run<512, 0>(B, st, sink, "512 thr, lds+barrier");
  run<256, 0>(B, st, sink, "256 thr, lds+barrier");
  run<256, 1>(B, st, sink, "256 thr, no lds");
  return 0;
}
