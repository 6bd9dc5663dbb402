// Probe: row-block GEMM (K=256) with 5 waves x 64 rows per workgroup (two 32-row slabs per wave, 512-register budget):
// every weight fragment feeds two MFMAs, and (MODE 1) the epilogue of tile j-1 is interleaved with the MFMAs of tile j
// inside each wave (double-buffered accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define RB_K 256
#define TILE_HALFS (64 * 256)
#define EPITCH 68
#define EFLOATS (16 * EPITCH)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 f = {a, b};
  b2 r = __builtin_convertvector(f, b2);
  return __builtin_bit_cast(unsigned, r);
}
template <typename F>
__device__ __forceinline__ void call_restrict(F&& f, int j, const bf16_t* __restrict__ cur, bf16_t* __restrict__ nxt) { f(j, cur, nxt); }

template <int MODE>
__global__ __launch_bounds__(320) void rb2_kernel(const bf16_t* A, const bf16_t* Bw, bf16_t* C, int M, int N) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* const Bs = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* const Es = reinterpret_cast<float*>(smem + 2 * TILE_HALFS * 2) + wave * EFLOATS;
  const int m0 = (blockIdx.x * 5 + wave) * 64;
  const int fr = lane & 31, fk = (lane >> 5) * 8, hk = lane >> 5, sw = fr & 31;
  bf16x8 af[2][16];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int row = m0 + s * 32 + fr, rc = row < M ? row : M - 1;
    const bf16_t* ap = A + (long long)rc * 256 + fk;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) af[s][ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 16);
  }
  const int ntiles = N / 64;
  auto load_tile = [&](int n0, bf16_t* S) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int k = wave + 5 * i;
      if (k < 32) {
        const int r = 2 * k + (lane >> 5), q = lane & 31;
        const int g = n0 + r, gc = g < N ? g : N - 1;
        __builtin_amdgcn_global_load_lds((gbl_void*)(Bw + (long long)gc * 256 + ((q ^ (r & 31)) * 8)), (lds_void*)(S + k * 512), 16, 0, 0);
      }
    }
  };
  f32x16 acc[2][2][2];   // [buffer][slab][col block]
  auto mfma_tile = [&](const bf16_t* cur, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][s][c][r] = 0.f;
    const bf16_t* wp0 = cur + fr * RB_K;
    const bf16_t* wp1 = wp0 + 32 * RB_K;
    bf16x8 wa[3], wb[3];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      wa[d] = *reinterpret_cast<const bf16x8*>(wp0 + (((2 * d + hk) ^ sw) * 8));
      wb[d] = *reinterpret_cast<const bf16x8*>(wp1 + (((2 * d + hk) ^ sw) * 8));
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 2 < 16) {
        wa[(ks + 2) % 3] = *reinterpret_cast<const bf16x8*>(wp0 + (((2 * (ks + 2) + hk) ^ sw) * 8));
        wb[(ks + 2) % 3] = *reinterpret_cast<const bf16x8*>(wp1 + (((2 * (ks + 2) + hk) ^ sw) * 8));
      }
      acc[b][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][ks], wa[ks % 3], acc[b][0][0], 0, 0, 0);
      acc[b][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][ks], wa[ks % 3], acc[b][1][0], 0, 0, 0);
      acc[b][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][ks], wb[ks % 3], acc[b][0][1], 0, 0, 0);
      acc[b][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][ks], wb[ks % 3], acc[b][1][1], 0, 0, 0);
      if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto epi_tile = [&](int jt, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int r = hf * 8 + rr, row = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
          Es[row * EPITCH + fr] = acc[b][s][0][r];
          Es[row * EPITCH + 32 + fr] = acc[b][s][1][r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int t = lane + 64 * i, rl = t >> 3, cg = t & 7;
          const int row = m0 + s * 32 + hf * 16 + rl;
          const float4 c0 = *reinterpret_cast<const float4*>(Es + rl * EPITCH + cg * 8);
          const float4 c1 = *reinterpret_cast<const float4*>(Es + rl * EPITCH + cg * 8 + 4);
          const uint4 o = make_uint4(pack2(c0.x, c0.y), pack2(c0.z, c0.w), pack2(c1.x, c1.y), pack2(c1.z, c1.w));
          if (row < M) *reinterpret_cast<uint4*>(C + (long long)row * N + jt * 64 + cg * 8) = o;
        }
        __builtin_amdgcn_wave_barrier();
      }
  };
  load_tile(0, Bs);
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  if (MODE == 0) {
    auto tile = [&](int j, const bf16_t* cur, bf16_t* nxt) __attribute__((always_inline)) {
      load_tile((j + 1) * 64, nxt);
      mfma_tile(cur, 0);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      epi_tile(j, 0);
      __syncthreads();
    };
    for (int j = 0; j < ntiles; ++j) call_restrict(tile, j, Bs + (j & 1) * TILE_HALFS, Bs + ((j + 1) & 1) * TILE_HALFS);
  } else {
    // software pipeline across tiles: MFMAs of tile j and the epilogue of tile j-1 sit in the same scheduling region
    auto tileA = [&](int j, const bf16_t* cur, bf16_t* nxt) __attribute__((always_inline)) {   // even j: MFMA -> buffer 0, epilogue of buffer 1
      load_tile((j + 1) * 64, nxt);
      if (j > 0) epi_tile(j - 1, 1);
      mfma_tile(cur, 0);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __syncthreads();
    };
    auto tileB = [&](int j, const bf16_t* cur, bf16_t* nxt) __attribute__((always_inline)) {
      load_tile((j + 1) * 64, nxt);
      epi_tile(j - 1, 0);
      mfma_tile(cur, 1);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __syncthreads();
    };
    for (int j = 0; j < ntiles; j += 2) {
      call_restrict(tileA, j, Bs, Bs + TILE_HALFS);
      call_restrict(tileB, j + 1, Bs + TILE_HALFS, Bs);
    }
    epi_tile(ntiles - 1, 1);
  }
}

template <int MODE>
float run(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N) {
  auto k = rb2_kernel<MODE>;
  const size_t lds = (size_t)2 * TILE_HALFS * 2 + (size_t)5 * EFLOATS * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = (M + 319) / 320;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(320), lds, 0, A, B, C, M, N);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(320), lds, 0, A, B, C, M, N);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 50.f;
}
int main() {
  const int M = 81920;
  bf16_t *A, *B, *C;
  hipMalloc(&A, (size_t)M * 256 * 2); hipMalloc(&B, (size_t)1024 * 256 * 2); hipMalloc(&C, (size_t)M * 1024 * 2);
  hipMemset(A, 0x3c, (size_t)M * 256 * 2); hipMemset(B, 0x3c, (size_t)1024 * 256 * 2);
  for (int N : {768, 1024}) {
    printf("N=%d: 5x64 plain %.1f us | 5x64 cross-tile interleave %.1f us\n", N, run<0>(A, B, C, M, N), run<1>(A, B, C, M, N));
  }
  return 0;
}
