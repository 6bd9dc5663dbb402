#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 16 * 4);
  for (int nw : {13, 14, 16, 8}) {
    hipMemset(d, 0, 4 * 16 * 4);
    k<<<4, 64 * nw, 0, 0>>>(d);
    unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) { printf("W=%d wg%d simd:", nw, b); for (int w = 0; w < nw; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3); printf("  wave_id:"); for (int w = 0; w < nw; ++w) printf(" %u", h[b * 16 + w] & 15); printf("\n"); }
  }
}
