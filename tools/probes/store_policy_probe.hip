// Does the cache policy of a PRODUCER's stores change how fast the next kernel reads the data?  (MI355X: 32 MB of L2, 256 MB of Infinity
// Cache.)  For each store policy: rewrite a 1.5-GB buffer (evicts everything), write X (150 MB) with that policy, then time a plain
// streaming read of X.   hipcc --offload-arch=gfx950 -O3 tools/probes/store_policy_probe.hip -o /tmp/spp && /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int POL>
__global__ void produce(v4u* x, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    v4u v = {(unsigned)i, 1u, 2u, 3u};
    v4u* p = x + i;
    if (POL == 0) *p = v;
    else if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (POL == 6) { v4u o = *p; v.y += o.x; *p = v; }   // read-modify-write
  }
}
__global__ void evict(float* j, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) j[i] += 1.0f;
}
template <int LPOL>
__global__ void consume(const v4u* x, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    v4u v;
    if (LPOL == 0) v = x[i];
    else asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(x + i) : "memory");
    acc += v.x ^ v.w;
  }
  if (acc == 0x12345678u) *out = acc;
}
int main() {
  const size_t nx = 150u * 1024 * 1024 / 16, nj = 1536u * 1024 * 1024 / 4;
  v4u* x; float* j; unsigned* o;
  hipMalloc(&x, nx * 16); hipMalloc(&j, nj * 4); hipMalloc(&o, 4);
  hipMemset(j, 0, nj * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[7] = {"default", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "read-modify-write"};
  for (int pol = 0; pol < 7; ++pol) {
    float tp = 0, tc = 0;
    for (int it = 0; it < 6; ++it) {
      evict<<<2048, 256>>>(j, nj);
      hipEventRecord(a);
      switch (pol) {
        case 0: produce<0><<<2048, 256>>>(x, nx); break;
        case 1: produce<1><<<2048, 256>>>(x, nx); break;
        case 2: produce<2><<<2048, 256>>>(x, nx); break;
        case 3: produce<3><<<2048, 256>>>(x, nx); break;
        case 4: produce<4><<<2048, 256>>>(x, nx); break;
        case 5: produce<5><<<2048, 256>>>(x, nx); break;
        default: produce<6><<<2048, 256>>>(x, nx); break;
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (it) tp += ms;
      hipEventRecord(a);
      consume<0><<<2048, 256>>>(x, nx, o);
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b); if (it) tc += ms;
    }
    printf("stores %-18s  produce %6.1f us   the read that follows %6.1f us  (%.2f TB/s)\n", names[pol], tp / 5 * 1e3, tc / 5 * 1e3, 150.0 * 1.048576 / (tc / 5) / 1e3);
  }
  // reference points: the read on fully warm caches (read twice) and fully cold
  float tw = 0, tcold = 0;
  for (int it = 0; it < 6; ++it) {
    consume<0><<<2048, 256>>>(x, nx, o);
    hipEventRecord(a); consume<0><<<2048, 256>>>(x, nx, o); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (it) tw += ms;
    evict<<<2048, 256>>>(j, nj);
    hipEventRecord(a); consume<0><<<2048, 256>>>(x, nx, o); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); if (it) tcold += ms;
  }
  printf("read after a read (warm) %6.1f us (%.2f TB/s);   read after the eviction (cold) %6.1f us (%.2f TB/s)\n", tw / 5 * 1e3, 157.3 / (tw / 5) / 1e3, tcold / 5 * 1e3, 157.3 / (tcold / 5) / 1e3);
  return 0;
}
