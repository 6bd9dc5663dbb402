// Phase-ablation probe for the row-block GEMM (K=256): which of {A load, weight LDS reads + MFMA, LDS transpose, global stores}
// overlap?  Build: hipcc --offload-arch=gfx950 -O3 -o rb_probe rb_probe.hip ; run: ./rb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define RB_BN 64
#define RB_PITCH 264
#define RB_TILE_HALFS (RB_BN * RB_PITCH)
#define RB_EPITCH 68
#define RB_EFLOATS (32 * RB_EPITCH)
#ifndef PD
#define PD 3
#endif

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 f = {a, b};
  b2 r = __builtin_convertvector(f, b2);
  return __builtin_bit_cast(unsigned, r);
}

// FLAGS: bit0 = MFMA+weight reads, bit1 = LDS transpose, bit2 = global stores, bit3 = second output (GELU-like traffic)
template <int FLAGS>
__global__ __launch_bounds__(640) void rb_kernel(const bf16_t* A, const bf16_t* Bw, bf16_t* C, bf16_t* C2, int M, int N, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* const Bs = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x;
  float* const Es = reinterpret_cast<float*>(smem + 2 * RB_TILE_HALFS * 2) + wave * RB_EFLOATS;
  const int m0 = (blockIdx.x * W + wave) * 32;
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  bf16x8 af[16];
  {
    const int rc = (m0 + fr) < M ? (m0 + fr) : M - 1;
    const bf16_t* ap = A + (long long)((FLAGS & 32) ? (rc & 31) : rc) * 256 + fk;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 16);
  }
  const int ntiles = N / RB_BN;
  uint4 st0, st1, st2, st3;
#define LD1(R, i, n0) { const int c = tid + NT * i; const int cc = c < 2048 ? c : 2047; R = *reinterpret_cast<const uint4*>(Bw + (long long)((n0) + (cc >> 5)) * 256 + (cc & 31) * 8); }
#define ST1(R, i, S) { const int c = tid + NT * i; if (c < 2048) *reinterpret_cast<uint4*>((S) + (c >> 5) * RB_PITCH + (c & 31) * 8) = R; }
#define load_tile(n0) { LD1(st0, 0, n0) LD1(st1, 1, n0) LD1(st2, 2, n0) LD1(st3, 3, n0) }
#define store_tile(S) { ST1(st0, 0, S) ST1(st1, 1, S) ST1(st2, 2, S) ST1(st3, 3, S) }
  const int rot = (FLAGS & 64) ? (blockIdx.x >> 3) % ntiles : 0;
  load_tile(rot * RB_BN)
  store_tile(Bs)
  __syncthreads();
  f32x16 acc0, acc1;
  auto mfma_phase = [&](const bf16_t* cur, int j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (FLAGS & 1) {
      const bf16_t* wp0 = cur + fr * RB_PITCH + fk;
      const bf16_t* wp1 = wp0 + 32 * RB_PITCH;
      bf16x8 wa[PD], wb[PD];
#pragma unroll
      for (int d = 0; d < PD - 1; ++d) {
        wa[d] = *reinterpret_cast<const bf16x8*>(wp0 + d * 16);
        wb[d] = *reinterpret_cast<const bf16x8*>(wp1 + d * 16);
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks + PD - 1 < 16) {
          wa[(ks + PD - 1) % PD] = *reinterpret_cast<const bf16x8*>(wp0 + (ks + PD - 1) * 16);
          wb[(ks + PD - 1) % PD] = *reinterpret_cast<const bf16x8*>(wp1 + (ks + PD - 1) * 16);
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], wa[ks % PD], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], wb[ks % PD], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = (float)__builtin_bit_cast(unsigned short, af[r][0]); acc1[r] = acc0[r] + j; }
    }
  };
  auto epi_phase = [&](int jt) {
    if (FLAGS & 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Es[row * RB_EPITCH + fr] = acc0[r];
        Es[row * RB_EPITCH + 32 + fr] = acc1[r];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int task = lane + 64 * i, row = task >> 3, cg = task & 7;
      float v[8];
      if (FLAGS & 2) {
        const float4 c0 = *reinterpret_cast<const float4*>(Es + row * RB_EPITCH + cg * 8);
        const float4 c1 = *reinterpret_cast<const float4*>(Es + row * RB_EPITCH + cg * 8 + 4);
        v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc0[(i * 4 + e) & 15] + acc1[e];
      }
      const uint4 o = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
      const long long off = (long long)(m0 + row) * N + ((jt + rot) % ntiles) * RB_BN + cg * 8;
      if (FLAGS & 4) {
        if (m0 + row < M) {
          *reinterpret_cast<uint4*>(C + off) = o;
          if (FLAGS & 8) *reinterpret_cast<uint4*>(C2 + off) = o;
        }
      } else if (o.x == 0x12345678u && o.y == 0x9abcdef0u) {
        *reinterpret_cast<uint4*>(C + off) = o;
      }
    }
    if (FLAGS & 2) __builtin_amdgcn_wave_barrier();
  };
  const bool late = (FLAGS & 128) && ((wave >> 2) & 1);   // "late" waves run their epilogue one interval behind
  for (int j = 0; j < ntiles; ++j) {
    const bf16_t* cur = Bs + (j & 1) * RB_TILE_HALFS;
    if (!(FLAGS & 16) && j + 1 < ntiles) load_tile(((j + 1 + rot) % ntiles) * RB_BN)
    if (late) {
      if (j > 0) epi_phase(j - 1);
      mfma_phase(cur, j);
    } else {
      mfma_phase(cur, j);
      epi_phase(j);
    }
    if (!(FLAGS & 16)) {
      if (j + 1 < ntiles) store_tile(Bs + ((j + 1) & 1) * RB_TILE_HALFS)
      __syncthreads();
    }
  }
  if (late) epi_phase(ntiles - 1);
}

template <int FLAGS>
float run(const bf16_t* A, const bf16_t* B, bf16_t* C, bf16_t* C2, int M, int N, int W) {
  auto k = rb_kernel<FLAGS>;
  const size_t lds = (size_t)2 * RB_TILE_HALFS * 2 + (size_t)W * RB_EFLOATS * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = ((M + 31) / 32 + W - 1) / W;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * W), lds, 0, A, B, C, C2, M, N, W);
  hipEventRecord(e0, 0);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * W), lds, 0, A, B, C, C2, M, N, W);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / it;
}

int main() {
  const int M = 81920, N = 1024, W = 10;
  bf16_t *A, *B, *C, *C2;
  hipMalloc(&A, (size_t)M * 256 * 2); hipMalloc(&B, (size_t)N * 256 * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&C2, (size_t)M * N * 2);
  hipMemset(A, 0x3c, (size_t)M * 256 * 2); hipMemset(B, 0x3c, (size_t)N * 256 * 2);
  printf("full (mfma+lds+store)        %7.1f us\n", run<7>(A, B, C, C2, M, N, W));
  printf("full + second output         %7.1f us\n", run<15>(A, B, C, C2, M, N, W));
  printf("no store                     %7.1f us\n", run<3>(A, B, C, C2, M, N, W));
  printf("no mfma (lds transp + store) %7.1f us\n", run<6>(A, B, C, C2, M, N, W));
  printf("store only                   %7.1f us\n", run<4>(A, B, C, C2, M, N, W));
  printf("store only, two outputs      %7.1f us\n", run<12>(A, B, C, C2, M, N, W));
  printf("mfma only                    %7.1f us\n", run<1>(A, B, C, C2, M, N, W));
  printf("nothing (A load + W staging) %7.1f us\n", run<0>(A, B, C, C2, M, N, W));
  printf("A load only (no W loop)      %7.1f us\n", run<16>(A, B, C, C2, M, N, W));
  printf("W loop only (A from cache)   %7.1f us\n", run<32>(A, B, C, C2, M, N, W));
  printf("W loop only, rotated tiles   %7.1f us\n", run<32+64>(A, B, C, C2, M, N, W));
  printf("full, rotated tiles          %7.1f us\n", run<7+64>(A, B, C, C2, M, N, W));
  printf("full, phase-offset groups    %7.1f us\n", run<7+128>(A, B, C, C2, M, N, W));
  printf("full+2nd out, phase-offset   %7.1f us\n", run<15+128>(A, B, C, C2, M, N, W));
  printf("neither                      %7.1f us\n", run<48>(A, B, C, C2, M, N, W));
  printf("mfma only, A cached          %7.1f us\n", run<33>(A, B, C, C2, M, N, W));
  printf("full, A cached               %7.1f us\n", run<39>(A, B, C, C2, M, N, W));
  return 0;
}
