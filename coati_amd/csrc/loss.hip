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
// (F.cross_entropy ignore_index=-1) but every column stays in the softmax as a negative.  blockIdx.y = 0 / 1: the two directions
// (logits / logits2, loss_sum / loss_sum2) in one launch.  NR > 0: the row (N <= 64 NR columns) is read ONCE and stays in registers
// for the three passes (max, sum, gradient); NR = 0: any N, three passes over memory.
template <int NR>
__global__ __launch_bounds__(256) void infonce_rows_kernel(float* __restrict__ logits, float* __restrict__ logits2, long long ld, int R, int N,
                                                           int label0, const unsigned char* __restrict__ bad, float* __restrict__ loss_sum,
                                                           float* __restrict__ loss_sum2, const float* __restrict__ inv_count, float gscale) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float* row = (blockIdx.y ? logits2 : logits) + (long long)r * ld;
  float* ls = blockIdx.y ? loss_sum2 : loss_sum;
  const int label = label0 + r;
  if (bad[label]) {
    for (int c = lane; c < N; c += 64) row[c] = 0.f;
    return;
  }
  const float g = gscale * inv_count[0];
  if constexpr (NR > 0) {
    float v[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) v[i] = lane + 64 * i < N ? row[lane + 64 * i] : -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NR; ++i) mx = fmaxf(mx, v[i]);
    mx = wave_max(mx);
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) sm += __expf(v[i] - mx);   // (exp(-inf) = 0 for the columns past N)
    sm = wave_sum(sm);
    const float lse = mx + __logf(sm);
    float at_label = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int c = lane + 64 * i;
      float p = __expf(v[i] - lse);
      if (c == label) { at_label = v[i]; p -= 1.f; }
      if (c < N) row[c] = p * g;
    }
    at_label = wave_sum(at_label);   // (one lane holds it)
    if (lane == 0) atomicAdd(ls, lse - at_label);
  } else {
    float mx = -INFINITY;
    for (int c = lane; c < N; c += 64) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int c = lane; c < N; c += 64) sm += __expf(row[c] - mx);
    sm = wave_sum(sm);
    const float lse = mx + __logf(sm);
    if (lane == 0) atomicAdd(ls, lse - row[label]);
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < N; c += 64) {
      float p = __expf(row[c] - lse);
      if (c == label) p -= 1.f;
      row[c] = p * g;
    }
  }
}
int launch_infonce_rows2(float* logits, float* logits2, long long ld, int R, int N, int label0, const unsigned char* bad,
                         float* loss_sum, float* loss_sum2, const float* inv_count, float gscale, hipStream_t s) {
  COATI_CHECK_ARG(logits && bad && loss_sum && inv_count && (logits2 == nullptr || loss_sum2 != nullptr), "infonce_rows: null operand");
  COATI_CHECK_SHAPE(R > 0 && N > 0 && label0 >= 0 && label0 + R <= N, "infonce_rows: labels out of range");
  const dim3 grid(cdiv(R, 4), logits2 ? 2 : 1);
  if (N <= 64 * 16) hipLaunchKernelGGL(infonce_rows_kernel<16>, grid, dim3(256), 0, s, logits, logits2, ld, R, N, label0, bad, loss_sum, loss_sum2, inv_count, gscale);
  else if (N <= 64 * 32) hipLaunchKernelGGL(infonce_rows_kernel<32>, grid, dim3(256), 0, s, logits, logits2, ld, R, N, label0, bad, loss_sum, loss_sum2, inv_count, gscale);
  else hipLaunchKernelGGL(infonce_rows_kernel<0>, grid, dim3(256), 0, s, logits, logits2, ld, R, N, label0, bad, loss_sum, loss_sum2, inv_count, gscale);
  COATI_LAUNCH_CHECK("infonce_rows");
  return COATI_OK;
}
int launch_infonce_rows(float* logits, long long ld, int R, int N, int label0, const unsigned char* bad,
                        float* loss_sum, const float* inv_count, float gscale, hipStream_t s) {
  return launch_infonce_rows2(logits, nullptr, ld, R, N, label0, bad, loss_sum, nullptr, inv_count, gscale, s);
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

// ---------------------------------------------------------------------------------------------------------------------
// Barlow-Twins head over the two [B,E] embeddings (BASELINE.json configs[3]; the reference holds no Barlow code, so this
// follows the Barlow-Twins formulation and is PARITY UNPINNED -- see oracle.barlow_loss):
//   z~ = keep * (z - mu) / sqrt(var + eps) per embedding dim over the valid rows of the GLOBAL batch (biased variance),
//   C = Za~^T Zb~ / n,  L = sum_i (1 - C_ii)^2 + lambda * sum_{i != j} C_ij^2.
// Column statistics and C are small ([2E], [E,E]) and are what a multi-GPU run all-reduces.
// ---------------------------------------------------------------------------------------------------------------------
// out[c] = sum_b keep[b] a[b,c] ; out[E + c] = sum_b keep[b] a[b,c] * (b2 ? b2[b,c] : a[b,c]).  Deterministic (no atomics).
__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ a, const float* __restrict__ b2,
                                                      const unsigned char* __restrict__ bad, float* __restrict__ out, int B, int E) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f;
  if (c < E)
    for (int r = rg; r < B; r += 4) {
      if (bad[r]) continue;
      const float x = a[(long long)r * E + c];
      s1 += x;
      s2 += x * (b2 ? b2[(long long)r * E + c] : x);
    }
  red[0][rg][threadIdx.x & 63] = s1;
  red[1][rg][threadIdx.x & 63] = s2;
  __syncthreads();
  if (rg == 0 && c < E) {
    const int t = threadIdx.x;
    out[c] = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t];
    out[E + c] = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
  }
}
int launch_colsum2(const float* a, const float* b2, const unsigned char* bad, float* out, int B, int E, hipStream_t s) {
  COATI_CHECK_ARG(a && bad && out, "colsum2: null operand");
  hipLaunchKernelGGL(colsum2_kernel, dim3(cdiv(E, 64)), dim3(256), 0, s, a, b2, bad, out, B, E);
  COATI_LAUNCH_CHECK("colsum2");
  return COATI_OK;
}

// zc = keep * (z - sum[c] / n): the first of two passes over the batch statistics.  The variance is then taken from the
// CENTRED rows (sumsq of zc), not as E[z^2] - E[z]^2: with |mean| >> sigma the one-pass form loses the variance to
// cancellation, and the summation order of the partial sums (1 rank vs N ranks) then shows up at 1e-4 in the gradients.
__global__ void center_rows_kernel(const float* __restrict__ z, const unsigned char* __restrict__ bad, const float* __restrict__ sum,
                                   const float* __restrict__ count, float* __restrict__ zc, int B, int E) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * E) return;
  const int c = (int)(i % E), r = (int)(i / E);
  zc[i] = bad[r] ? 0.f : z[i] - sum[c] / fmaxf(count[0], 1.f);
}
int launch_center_rows(const float* z, const unsigned char* bad, const float* sum, const float* count, float* zc, int B, int E,
                       hipStream_t s) {
  COATI_CHECK_ARG(z && bad && sum && count && zc, "center_rows: null operand");
  hipLaunchKernelGGL(center_rows_kernel, dim3(cdiv((long long)B * E, 256)), dim3(256), 0, s, z, bad, sum, count, zc, B, E);
  COATI_LAUNCH_CHECK("center_rows");
  return COATI_OK;
}

// stats = [sum | sumsq] over the global valid rows, count[0] = n.  zt = keep (z - mu) * rsigma ; rsigma[c] written.
__global__ void standardize_kernel(const float* __restrict__ z, const unsigned char* __restrict__ bad, const float* __restrict__ stats,
                                   const float* __restrict__ count, float* __restrict__ zt, float* __restrict__ rsigma, int B, int E) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * E) return;
  const int c = (int)(i % E), r = (int)(i / E);
  const float n = fmaxf(count[0], 1.f);
  const float mu = stats[c] / n;
  const float var = fmaxf(stats[E + c] / n - mu * mu, 0.f);
  const float rs = 1.0f / sqrtf(var + 1e-5f);
  if (r == 0) rsigma[c] = rs;
  zt[i] = bad[r] ? 0.f : (z[i] - mu) * rs;
}
int launch_standardize(const float* z, const unsigned char* bad, const float* stats, const float* count, float* zt, float* rsigma,
                       int B, int E, hipStream_t s) {
  COATI_CHECK_ARG(z && bad && stats && count && zt && rsigma, "standardize: null operand");
  hipLaunchKernelGGL(standardize_kernel, dim3(cdiv((long long)B * E, 256)), dim3(256), 0, s, z, bad, stats, count, zt, rsigma, B, E);
  COATI_LAUNCH_CHECK("standardize");
  return COATI_OK;
}

// C holds the raw Za~^T Zb~ (already summed over ranks); in place -> G = dL/dC / n (so that dZ~ = Z~ G needs no further scaling);
// loss[0] = L.
__global__ __launch_bounds__(256) void barlow_dc_kernel(float* __restrict__ C, const float* __restrict__ count, float lam,
                                                        float* __restrict__ loss, int E) {
  __shared__ float red[4];
  const float n = fmaxf(count[0], 1.f);
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)E * E; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / E), c = (int)(i % E);
    const float v = C[i] / n;
    float g;
    if (r == c) { acc += (1.f - v) * (1.f - v); g = -2.f * (1.f - v); } else { acc += lam * v * v; g = 2.f * lam * v; }
    C[i] = g / n;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
}
int launch_barlow_dc(float* C, const float* count, float lam, float* loss, int E, hipStream_t s) {
  COATI_CHECK_ARG(C && count && loss, "barlow_dc: null operand");
  hipLaunchKernelGGL(barlow_dc_kernel, dim3(64), dim3(256), 0, s, C, count, lam, loss, E);
  COATI_LAUNCH_CHECK("barlow_dc");
  return COATI_OK;
}

// dz = keep * rsigma * (dzt - m1 - zt * m2) * scale, with m = [sum keep dzt | sum keep dzt zt] / n (batch-norm backward)
__global__ void standardize_bwd_kernel(const float* __restrict__ dzt, const float* __restrict__ zt, const unsigned char* __restrict__ bad,
                                       const float* __restrict__ rsigma, const float* __restrict__ m, const float* __restrict__ count,
                                       float scale, float* __restrict__ dz, int B, int E) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * E) return;
  const int c = (int)(i % E), r = (int)(i / E);
  const float n = fmaxf(count[0], 1.f);
  dz[i] = bad[r] ? 0.f : scale * rsigma[c] * (dzt[i] - m[c] / n - zt[i] * (m[E + c] / n));
}
int launch_standardize_bwd(const float* dzt, const float* zt, const unsigned char* bad, const float* rsigma, const float* m,
                           const float* count, float scale, float* dz, int B, int E, hipStream_t s) {
  COATI_CHECK_ARG(dzt && zt && bad && rsigma && m && count && dz, "standardize_bwd: null operand");
  hipLaunchKernelGGL(standardize_bwd_kernel, dim3(cdiv((long long)B * E, 256)), dim3(256), 0, s, dzt, zt, bad, rsigma, m, count, scale, dz, B, E);
  COATI_LAUNCH_CHECK("standardize_bwd");
  return COATI_OK;
}
