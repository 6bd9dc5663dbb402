// probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value = its own element index.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int lane = threadIdx.x;
  // mode 0: every lane passes natural address lane*8 bytes (4 elems per lane)
  // mode 1: row-major tile T[m][n] with pitch 128 elements: lane supplies &T[(lane&15)>>2 + 4*(lane>>4)][4*(lane&3)]
  const unsigned short* p;
  if (mode == 0) p = lds + lane * 4;
  else p = lds + (((lane & 15) >> 2) + 4 * (lane >> 4)) * 128 + 4 * (lane & 3);
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)r[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 512);
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
