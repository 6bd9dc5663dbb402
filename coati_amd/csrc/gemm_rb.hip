// Row-block ("A-stationary") bf16 GEMM for K = 256:  C[M,N] = epilogue(A[M,256] * B[N,256]^T + bias).
//
// Every Linear that reads the 256-wide stream has a tiny weight matrix (<= 512 KB, L2 resident) and a huge M (81,920
// rows): its arithmetic intensity is below the HBM ridge, so the tiled kernel's per-workgroup load -> MFMA -> store
// chain (two chains per CU) leaves the memory system half idle.  Here one workgroup per CU owns a block of rows for the
// whole kernel:
//   * each wave keeps its 32 x 256 slab of A as 16 MFMA A-fragments in registers (A is read from HBM exactly once);
//   * the weight tile [64 cols][256] streams through a double-buffered LDS image (rows padded to 528 B: conflict-free
//     ds_read_b128), the next tile's loads in flight behind the current tile's 32 MFMAs per wave, one barrier per tile;
//   * every wave transposes its 32 x 64 accumulator block through a PRIVATE LDS region (no workgroup barrier) and runs
//     the fused epilogue (gemm_epi.h) on 8 consecutive columns per lane: 128-B row segments, 16-B stores;
//   * W waves per workgroup is chosen on the host so that the grid is a whole number of 256-CU rounds.
#include "gemm_epi.h"

#define RB_K 256
#define RB_BN 64
#define RB_PITCH 264                      // halfs per LDS row of the weight tile (528 B)
#define RB_TILE_HALFS (RB_BN * RB_PITCH)  // 33,792 B per tile
#define RB_EPITCH 68                      // floats per row of the per-wave transpose region
#define RB_EFLOATS (32 * RB_EPITCH)       // 8,704 B per wave
#define RB_MAX_W 10

template <int EPI>
__global__ __launch_bounds__(64 * RB_MAX_W) void gemm_rb256_kernel(GemmArgs p, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* const Bs = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x;
  float* const Es = reinterpret_cast<float*>(smem + 2 * RB_TILE_HALFS * 2) + wave * RB_EFLOATS;
  const int m0 = (blockIdx.x * W + wave) * 32;
  const int fr = lane & 31, fk = (lane >> 5) * 8;

  // resident A slab: A-operand fragments, lane (i = lane&31 -> row, kg = lane>>5) holds k = ks*16 + kg*8 .. +7
  bf16x8 af[16];
  {
    const int rc = (m0 + fr) < p.M ? (m0 + fr) : p.M - 1;
    const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.A) + (long long)rc * p.lda + fk;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 16);
  }

  const int ntiles = (p.N + RB_BN - 1) / RB_BN;
  uint4 st[4];   // weight-tile staging: 2048 16-B chunks over NT threads (NT >= 512 -> at most 4 each)
  auto load_tile = [&](int n0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + NT * i;
      if (c < 2048) {
        const int row = c >> 5, kc = (c & 31) * 8;
        const int g = n0 + row;
        const int gc = g < p.N ? g : p.N - 1;
        const uint4 t = *reinterpret_cast<const uint4*>(p.B + (long long)gc * p.ldb + kc);
        st[i] = (g < p.N) ? t : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_tile = [&](bf16_t* S) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + NT * i;
      if (c < 2048) *reinterpret_cast<uint4*>(S + (c >> 5) * RB_PITCH + (c & 31) * 8) = st[i];
    }
  };
  load_tile(0);
  store_tile(Bs);
  __syncthreads();

  for (int j = 0; j < ntiles; ++j) {
    const bf16_t* cur = Bs + (j & 1) * RB_TILE_HALFS;
    if (j + 1 < ntiles) load_tile((j + 1) * RB_BN);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(cur + fr * RB_PITCH + ks * 16 + fk);
      const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(cur + (32 + fr) * RB_PITCH + ks * 16 + fk);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], w0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], w1, acc1, 0, 0, 0);
    }
    // wave-private transpose: accumulator (lane = column, registers = rows) -> rows of 64 contiguous columns
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      Es[row * RB_EPITCH + fr] = acc0[r];
      Es[row * RB_EPITCH + 32 + fr] = acc1[r];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int task = lane + 64 * i, row = task >> 3, cg = task & 7;
      float v[8];
      const float4 c0 = *reinterpret_cast<const float4*>(Es + row * RB_EPITCH + cg * 8);
      const float4 c1 = *reinterpret_cast<const float4*>(Es + row * RB_EPITCH + cg * 8 + 4);
      v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
      epilogue8<EPI>(p, m0 + row, j * RB_BN + cg * 8, v, (m0 + row) < p.M, 0, 1);
    }
    __builtin_amdgcn_wave_barrier();      // the next tile's writes to Es stay behind these reads
    if (j + 1 < ntiles) store_tile(Bs + ((j + 1) & 1) * RB_TILE_HALFS);
    __syncthreads();
  }
}

static int rb_waves(int M) {
  const int slabs = (M + 31) / 32;
  const int rounds = (slabs + 256 * RB_MAX_W - 1) / (256 * RB_MAX_W);
  int W = (slabs + 256 * rounds - 1) / (256 * rounds);
  if (W > RB_MAX_W) W = RB_MAX_W;
  return W;
}

// true when (a, epi) can run on the row-block kernel
bool gemm_rb256_supported(const GemmArgs& a, int a_f32, int epi) {
  if (a_f32 || a.K != RB_K || epi == EPI_CE_PARTIAL) return false;
  if (a.N % 16 != 0 && epi != EPI_CE_BWD) return false;
  if (a.N < 512) return false;   // measured: with only 2-4 column tiles the tiled kernel is as fast or faster
  // epilogues that read or write extra row-major operands (f32 residuals, saved pre-activations): tiled kernel wins
  if (epi == EPI_RES_F32 || epi == EPI_ACC_F32 || epi == EPI_F32 || epi == EPI_DGELU || epi == EPI_DSILU || epi == EPI_EDGE_DPRE) return false;
  return rb_waves(a.M) >= 8;   // staging assumes >= 512 threads; smaller problems run on the tiled kernel
}

template <int EPI>
static int launch_rb_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_rb256_kernel<EPI>;
  const size_t lds_max = (size_t)2 * RB_TILE_HALFS * 2 + (size_t)RB_MAX_W * RB_EFLOATS * 4;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    if (e != hipSuccess) {
      coati_set_error("gemm_rb256: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return COATI_EHIP;
    }
    attr_set = true;
  }
  const int W = rb_waves(a.M);
  const int blocks = cdiv(cdiv(a.M, 32), W);
  const size_t lds = (size_t)2 * RB_TILE_HALFS * 2 + (size_t)W * RB_EFLOATS * 4;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), lds, s, a, W);
  COATI_LAUNCH_CHECK("gemm_rb256");
  return COATI_OK;
}

int launch_gemm_rb256(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return launch_rb_t<EPI_BF16>(a, s);
    case EPI_F32: return launch_rb_t<EPI_F32>(a, s);
    case EPI_RES_F32: return launch_rb_t<EPI_RES_F32>(a, s);
    case EPI_GELU: return launch_rb_t<EPI_GELU>(a, s);
    case EPI_DGELU: return launch_rb_t<EPI_DGELU>(a, s);
    case EPI_SILU: return launch_rb_t<EPI_SILU>(a, s);
    case EPI_DSILU: return launch_rb_t<EPI_DSILU>(a, s);
    case EPI_ACC_F32: return launch_rb_t<EPI_ACC_F32>(a, s);
    case EPI_CE_BWD: return launch_rb_t<EPI_CE_BWD>(a, s);
    case EPI_EDGE_DPRE: return launch_rb_t<EPI_EDGE_DPRE>(a, s);
    case EPI_QKV_ROPE: return launch_rb_t<EPI_QKV_ROPE>(a, s);
    default:
      coati_set_error("gemm_rb256: unsupported epilogue %d", epi);
      return COATI_EARG;
  }
}
