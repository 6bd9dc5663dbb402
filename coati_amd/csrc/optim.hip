// Optimiser over the flat fp32 parameter / gradient buffers (reference train_coati.py:145-152, 276-277):
// clip_grad_norm_(max_norm) followed by AdamW (decoupled weight decay on every parameter), plus the bf16
// "shadow" copies the MFMA GEMMs consume.  All HBM-bound streaming kernels: float4 per lane, grid-stride.
// The clip coefficient stays on the device (no host sync between backward and the update).
#include <math.h>
#include "kernels.h"

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partial, int n_partial, float* __restrict__ out_norm,
                                                           float max_norm, float* __restrict__ out_coef) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partial; i += 256) acc += (double)partial[i];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
    out_norm[0] = norm;
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    float coef = max_norm / (norm + 1e-6f);
    out_coef[0] = coef < 1.f ? coef : 1.f;
  }
}

int launch_grad_sqnorm(const float* g, long long n, float* partial, int n_partial, float* out_norm, float max_norm,
                       float* out_coef, hipStream_t s) {
  COATI_CHECK_ARG(g && partial && out_norm && out_coef && n_partial > 0, "grad_sqnorm: null operand");
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(n_partial), dim3(256), 0, s, g, n, partial);
  COATI_LAUNCH_CHECK("grad_sqnorm(partial)");
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, partial, n_partial, out_norm, max_norm, out_coef);
  COATI_LAUNCH_CHECK("grad_sqnorm(final)");
  return COATI_OK;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, long long n, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2,
                                                    const float* __restrict__ coef, float gscale, const int* __restrict__ skip) {
  // skip: the step's device-side error word (missing [STOP] / packed-row mismatch, engine.cpp err_flag): a step that ran on
  // wrong rows must not reach the weights -- the update is dropped here and the host raises when it reads the losses
  if (skip != nullptr && skip[0] != 0) return;
  const float cs = (coef ? coef[0] : 1.f) * gscale;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * cs;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    pi -= (lr * inv_bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (shadow) shadow[i] = f2bf(pi);
  }
}

int launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* shadow, long long n, float lr, float b1,
                 float b2, float eps, float wd, int step, const float* coef, float gscale, hipStream_t s, const int* skip) {
  COATI_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw: bad argument");
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, s, p, g, m, v, shadow, n, lr, b1, b2, eps, wd,
                     (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), coef, gscale, skip);
  COATI_LAUNCH_CHECK("adamw");
  return COATI_OK;
}

__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = f2bf(src[i]);
}
int launch_cast_bf16(const float* src, bf16_t* dst, long long n, hipStream_t s) {
  COATI_CHECK_ARG(src && dst && n > 0, "cast_bf16: bad argument");
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)blocks), dim3(256), 0, s, src, dst, n);
  COATI_LAUNCH_CHECK("cast_bf16");
  return COATI_OK;
}

// dst[c * ld_dst + r] = bf16(src[r * ld_src + c]), 32x32 tiles through LDS (coalesced both sides)
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, long long ld_src, bf16_t* __restrict__ dst,
                                                             long long ld_dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * ld_src + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(long long)c * ld_dst + r] = f2bf(tile[tx][i]);
  }
}
int launch_transpose_cast(const float* src, long long ld_src, bf16_t* dst, long long ld_dst, int rows, int cols, hipStream_t s) {
  COATI_CHECK_ARG(src && dst && rows > 0 && cols > 0, "transpose_cast: bad argument");
  hipLaunchKernelGGL(transpose_cast_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, s, src, ld_src, dst, ld_dst, rows, cols);
  COATI_LAUNCH_CHECK("transpose_cast");
  return COATI_OK;
}

__global__ void pack_rows_cast_kernel(const float* __restrict__ src, long long ld_src, bf16_t* __restrict__ dst, long long ld_dst,
                                      int rows, int cols) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long long)rows * cols) {
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    dst[(long long)r * ld_dst + c] = f2bf(src[(long long)r * ld_src + c]);
  }
}
int launch_pack_rows_cast(const float* src, long long ld_src, bf16_t* dst, long long ld_dst, int rows, int cols, hipStream_t s) {
  COATI_CHECK_ARG(src && dst && rows > 0 && cols > 0, "pack_rows_cast: bad argument");
  hipLaunchKernelGGL(pack_rows_cast_kernel, dim3(cdiv((long long)rows * cols, 256)), dim3(256), 0, s, src, ld_src, dst, ld_dst, rows, cols);
  COATI_LAUNCH_CHECK("pack_rows_cast");
  return COATI_OK;
}

// ---- batched transposes: one launch for every transposed / packed bf16 weight shadow ----------------------------
// Each job: dst[c * ld_dst + r] = bf16(src[r * ld_src + c]) (transpose) or dst[r * ld_dst + c] = bf16(src[r * ld_src + c]).
// blockIdx.x walks the concatenated 32x32 tile lists of all jobs (prefix sums in tile_start).
__global__ __launch_bounds__(256) void shadow_jobs_kernel(const ShadowJob* __restrict__ jobs, const int* __restrict__ tile_start,
                                                          int n_jobs, const float* __restrict__ P, bf16_t* __restrict__ S) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n_jobs - 1;
  const int b = blockIdx.x;
  while (lo < hi) {   // last job whose first tile <= b
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const ShadowJob j = jobs[lo];
  const int t = b - tile_start[lo];
  const int tiles_c = (j.cols + 31) >> 5;
  const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
  const float* src = P + j.src_off;
  bf16_t* dst = S + j.dst_off;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (j.transpose) {
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, c = c0 + tx;
      tile[i][tx] = (r < j.rows && c < j.cols) ? src[(long long)r * j.ld_src + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;
      if (c < j.cols && r < j.rows) dst[(long long)c * j.ld_dst + r] = f2bf(tile[tx][i]);
    }
  } else {
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, c = c0 + tx;
      if (r < j.rows && c < j.cols) dst[(long long)r * j.ld_dst + c] = f2bf(src[(long long)r * j.ld_src + c]);
    }
  }
}

int launch_shadow_jobs(const ShadowJob* jobs, const int* tile_start, int n_jobs, int n_tiles, const float* P, bf16_t* S,
                       hipStream_t s) {
  COATI_CHECK_ARG(jobs && tile_start && P && S && n_jobs > 0 && n_tiles > 0, "shadow_jobs: bad argument");
  hipLaunchKernelGGL(shadow_jobs_kernel, dim3(n_tiles), dim3(256), 0, s, jobs, tile_start, n_jobs, P, S);
  COATI_LAUNCH_CHECK("shadow_jobs");
  return COATI_OK;
}
