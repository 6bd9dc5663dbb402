// Small fp32 kernels around the contrastive head (reference clip_e2e.py:27-47, 431-435, 800-808).
// The [B,256] head maths stays fp32 end to end (raw, un-normalised dot products feed the InfoNCE logits, so bf16
// operands would cost ~1e-2 absolute logit error); the GEMMs themselves run on the exact-f32 MFMA (gemm.hip sgemm).
#include "kernels.h"

__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = silu_f(x[i]);
}
int launch_silu_fwd(const float* x, float* y, long long n, hipStream_t s) {
  COATI_CHECK_ARG(x && y, "silu_fwd: null operand");
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, n);
  COATI_LAUNCH_CHECK("silu_fwd");
  return COATI_OK;
}

__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                long long n, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float g = dy[i] * dsilu_f(x[i]);
    dx[i] = accumulate ? dx[i] + g : g;
  }
}
int launch_silu_bwd(const float* x, const float* dy, float* dx, long long n, int accumulate, hipStream_t s) {
  COATI_CHECK_ARG(x && dy && dx, "silu_bwd: null operand");
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, x, dy, dx, n, accumulate);
  COATI_LAUNCH_CHECK("silu_bwd");
  return COATI_OK;
}

__global__ void select_rows_kernel(const unsigned char* __restrict__ use_a, const float* __restrict__ a,
                                   const float* __restrict__ b, float* __restrict__ out, int B, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long long)B * C) out[i] = use_a[i / C] ? a[i] : b[i];
}
int launch_select_rows(const unsigned char* use_a, const float* a, const float* b, float* out, int B, int C, hipStream_t s) {
  COATI_CHECK_ARG(use_a && a && b && out, "select_rows: null operand");
  hipLaunchKernelGGL(select_rows_kernel, dim3(cdiv((long long)B * C, 256)), dim3(256), 0, s, use_a, a, b, out, B, C);
  COATI_LAUNCH_CHECK("select_rows");
  return COATI_OK;
}

__global__ void select_rows_bwd_kernel(const unsigned char* __restrict__ use_a, const float* __restrict__ dout,
                                       float* __restrict__ da, float* __restrict__ db, int B, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long long)B * C) {
    if (use_a[i / C]) da[i] += dout[i]; else db[i] += dout[i];
  }
}
int launch_select_rows_bwd(const unsigned char* use_a, const float* dout, float* da, float* db, int B, int C, hipStream_t s) {
  COATI_CHECK_ARG(use_a && dout && da && db, "select_rows_bwd: null operand");
  hipLaunchKernelGGL(select_rows_bwd_kernel, dim3(cdiv((long long)B * C, 256)), dim3(256), 0, s, use_a, dout, da, db, B, C);
  COATI_LAUNCH_CHECK("select_rows_bwd");
  return COATI_OK;
}

__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, float alpha, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += alpha * x[i];
}
int launch_axpy(const float* x, float* y, float alpha, long long n, hipStream_t s) {
  COATI_CHECK_ARG(x && y, "axpy: null operand");
  hipLaunchKernelGGL(axpy_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, x, y, alpha, n);
  COATI_LAUNCH_CHECK("axpy");
  return COATI_OK;
}

// One wave per logits row.  label(r) = label0 + r; rows whose label is a bad row contribute nothing
// (F.cross_entropy ignore_index=-1) but every column stays in the softmax as a negative.
__global__ __launch_bounds__(256) void infonce_rows_kernel(float* __restrict__ logits, long long ld, int R, int N, int label0,
                                                           const unsigned char* __restrict__ bad, float* __restrict__ loss_sum,
                                                           const float* __restrict__ inv_count, float gscale) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float* row = logits + (long long)r * ld;
  const int label = label0 + r;
  if (bad[label]) {
    for (int c = lane; c < N; c += 64) row[c] = 0.f;
    return;
  }
  float mx = -INFINITY;
  for (int c = lane; c < N; c += 64) mx = fmaxf(mx, row[c]);
  mx = wave_max(mx);
  float sm = 0.f;
  for (int c = lane; c < N; c += 64) sm += __expf(row[c] - mx);
  sm = wave_sum(sm);
  const float lse = mx + __logf(sm);
  const float g = gscale * inv_count[0];
  if (lane == 0) atomicAdd(loss_sum, lse - row[label]);
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < N; c += 64) {
    float p = __expf(row[c] - lse);
    if (c == label) p -= 1.f;
    row[c] = p * g;
  }
}
int launch_infonce_rows(float* logits, long long ld, int R, int N, int label0, const unsigned char* bad,
                        float* loss_sum, const float* inv_count, float gscale, hipStream_t s) {
  COATI_CHECK_ARG(logits && bad && loss_sum && inv_count, "infonce_rows: null operand");
  COATI_CHECK_SHAPE(R > 0 && N > 0 && label0 >= 0 && label0 + R <= N, "infonce_rows: labels out of range");
  hipLaunchKernelGGL(infonce_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, logits, ld, R, N, label0, bad, loss_sum, inv_count, gscale);
  COATI_LAUNCH_CHECK("infonce_rows");
  return COATI_OK;
}

__global__ void count_valid_kernel(const unsigned char* __restrict__ bad, int n, float* __restrict__ out_count,
                                   float* __restrict__ out_inv) {
  __shared__ int red[4];
  int c = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) c += bad[i] ? 0 : 1;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = red[0] + red[1] + red[2] + red[3];
    out_count[0] = (float)t;
    out_inv[0] = t > 0 ? 1.0f / (float)t : 0.f;
  }
}
int launch_count_valid(const unsigned char* bad, int n, float* out_count, float* out_inv, hipStream_t s) {
  COATI_CHECK_ARG(bad && out_count && out_inv, "count_valid: null operand");
  hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(256), 0, s, bad, n, out_count, out_inv);
  COATI_LAUNCH_CHECK("count_valid");
  return COATI_OK;
}
