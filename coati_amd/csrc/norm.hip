// LayerNorm / affine-free "instance" norm over the hidden dimension (eps 1e-5, biased variance).
// One 64-lane wave per row, float4 per lane per 256-column chunk, rows of up to 1024 columns.
// HBM-bound: every element is read once and written once; statistics are wave shuffles.
#include "kernels.h"

#define LN_EPS 1e-5f

// One wave normalises LN_ROWS rows: all their loads are issued before the first reduction (with one row per wave the
// kernel ran at 4.1 TB/s, latency-bound: 32 waves x 1 KiB in flight per CU).
#define LN_ROWS 4
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16_t* __restrict__ y16, long long ld16, float* __restrict__ y32,
                                                     long long ld32, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int M, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * LN_ROWS;
  if (row0 >= M) return;
  float4 v[LN_ROWS][NCH];
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    const int row = row0 + r < M ? row0 + r : M - 1;   // clamped: the tail rows are loaded twice and stored never
    const float* xr = x + (long long)row * ldx;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int idx = (j * 64 + lane) * 4;
      v[r][j] = (idx < C) ? *reinterpret_cast<const float4*>(xr + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 g[NCH], bt[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int idx = (j * 64 + lane) * 4;
    g[j] = (gamma && idx < C) ? *reinterpret_cast<const float4*>(gamma + idx) : make_float4(1.f, 1.f, 1.f, 1.f);
    bt[j] = (gamma && idx < C) ? *reinterpret_cast<const float4*>(beta + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    const int row = row0 + r;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) s += v[r][j].x + v[r][j].y + v[r][j].z + v[r][j].w;
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int idx = (j * 64 + lane) * 4;
      if (idx < C) {
        const float a = v[r][j].x - mean, b = v[r][j].y - mean, c = v[r][j].z - mean, d = v[r][j].w - mean;
        q += a * a + b * b + c * c + d * d;
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + LN_EPS);
    if (row < M) {
      if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
      }
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int idx = (j * 64 + lane) * 4;
        if (idx < C) {
          float4 o = make_float4((v[r][j].x - mean) * rstd, (v[r][j].y - mean) * rstd, (v[r][j].z - mean) * rstd,
                                 (v[r][j].w - mean) * rstd);
          if (gamma) {
            o.x = o.x * g[j].x + bt[j].x; o.y = o.y * g[j].y + bt[j].y; o.z = o.z * g[j].z + bt[j].z; o.w = o.w * g[j].w + bt[j].w;
          }
          if (y32) *reinterpret_cast<float4*>(y32 + (long long)row * ld32 + idx) = o;
          if (y16) *reinterpret_cast<uint2*>(y16 + (long long)row * ld16 + idx) = make_uint2(pack2bf(o.x, o.y), pack2bf(o.z, o.w));
        }
      }
    }
  }
}

int launch_layernorm_fwd(const float* x, long long ldx, const float* gamma, const float* beta, bf16_t* y16,
                         long long ld16, float* y32, long long ld32, float* mean, float* rstd, int M, int C,
                         hipStream_t s) {
  COATI_CHECK_ARG(x && (y16 || y32), "layernorm_fwd: null operand");
  COATI_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "layernorm_fwd: gamma/beta must both be given or both null");
  COATI_CHECK_SHAPE(M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ld16 % 4 == 0 && ld32 % 4 == 0,
                    "layernorm_fwd: unsupported shape M=%d C=%d", M, C);
  const int nch = cdiv(C, 256);
  dim3 grid(cdiv(M, 4 * LN_ROWS)), block(256);
#define LN_F(N) hipLaunchKernelGGL(ln_fwd_kernel<N>, grid, block, 0, s, x, ldx, gamma, beta, y16, ld16, y32, ld32, mean, rstd, M, C)
  if (nch == 1) LN_F(1); else if (nch == 2) LN_F(2); else if (nch == 3) LN_F(3); else LN_F(4);
#undef LN_F
  COATI_LAUNCH_CHECK("layernorm_fwd");
  return COATI_OK;
}

template <int NCH, bool DY_F32>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_, long long lddy,
                                                     const float* __restrict__ x, long long ldx, int x_is_xhat,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* dres, float* dx,
                                                     bf16_t* __restrict__ dx16, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ partial, int M, int C) {
  __shared__ float red[2][4][NCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 ag[NCH], ab[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gm[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int idx = (j * 64 + lane) * 4;
    gm[j] = (gamma && idx < C) ? *reinterpret_cast<const float4*>(gamma + idx) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float rs = rstd[row];
    const float mu = x_is_xhat ? 0.f : mean[row];
    float4 dyv[NCH], xh[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int idx = (j * 64 + lane) * 4;
      if (idx < C) {
        if (DY_F32) {
          dyv[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + (long long)row * lddy + idx);
        } else {
          const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(dy_) + (long long)row * lddy + idx);
          dyv[j] = make_float4(bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y));
        }
        const float4 xv = *reinterpret_cast<const float4*>(x + (long long)row * ldx + idx);
        xh[j] = x_is_xhat ? xv : make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        const float4 g = make_float4(dyv[j].x * gm[j].x, dyv[j].y * gm[j].y, dyv[j].z * gm[j].z, dyv[j].w * gm[j].w);
        s1 += g.x + g.y + g.z + g.w;
        s2 += g.x * xh[j].x + g.y * xh[j].y + g.z * xh[j].z + g.w * xh[j].w;
      } else {
        dyv[j] = xh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int idx = (j * 64 + lane) * 4;
      if (idx < C) {
        float4 o;
        o.x = rs * (dyv[j].x * gm[j].x - s1 - xh[j].x * s2);
        o.y = rs * (dyv[j].y * gm[j].y - s1 - xh[j].y * s2);
        o.z = rs * (dyv[j].z * gm[j].z - s1 - xh[j].z * s2);
        o.w = rs * (dyv[j].w * gm[j].w - s1 - xh[j].w * s2);
        if (dres) {
          const float4 r = *reinterpret_cast<const float4*>(dres + (long long)row * C + idx);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        *reinterpret_cast<float4*>(dx + (long long)row * C + idx) = o;
        if (dx16) *reinterpret_cast<uint2*>(dx16 + (long long)row * C + idx) = make_uint2(pack2bf(o.x, o.y), pack2bf(o.z, o.w));
        ag[j].x += dyv[j].x * xh[j].x; ag[j].y += dyv[j].y * xh[j].y; ag[j].z += dyv[j].z * xh[j].z; ag[j].w += dyv[j].w * xh[j].w;
        ab[j].x += dyv[j].x; ab[j].y += dyv[j].y; ab[j].z += dyv[j].z; ab[j].w += dyv[j].w;
      }
    }
  }
  if (dgamma == nullptr) return;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int idx = (j * 64 + lane) * 4;
    *reinterpret_cast<float4*>(&red[0][wave][idx]) = ag[j];
    *reinterpret_cast<float4*>(&red[1][wave][idx]) = ab[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float g = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    if (partial) {   // stage 1 of the deterministic two-stage reduction: one row of [gamma | beta] sums per workgroup
      partial[(long long)blockIdx.x * 2 * C + c] = g;
      partial[(long long)blockIdx.x * 2 * C + C + c] = b;
    } else {         // thousands of workgroups hammering 2C addresses: ~45 us of serialised L2 atomics at C = 256
      atomicAdd(dgamma + c, g);
      atomicAdd(dbeta + c, b);
    }
  }
}

// stage 2: column sums of partial[nblk][2C] added into dgamma | dbeta.  grid (2C/64, 32 row chunks): 32 atomics per column
// instead of one per stage-1 workgroup.
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int C) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per;
  int r1 = r0 + per;
  if (r1 > nblk) r1 = nblk;
  float acc = 0.f;
  if (col < 2 * C) {
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 4) acc += partial[(long long)r * 2 * C + col];
  }
  red[rg][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rg == 0 && col < 2 * C) {
    const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(col < C ? dgamma + col : dbeta + (col - C), t);
  }
}

static int ln_bwd_stage1(const void* dy, int dy_f32, long long lddy, const float* x, long long ldx, int x_is_xhat,
                         const float* mean, const float* rstd, const float* gamma, const float* dres,
                         float* dx, bf16_t* dx16, float* dgamma, float* dbeta, float* partial, int M, int C, int* nblk_out, hipStream_t s) {
  COATI_CHECK_ARG(dy && x && rstd && dx, "layernorm_bwd: null operand");
  COATI_CHECK_ARG(x_is_xhat || mean, "layernorm_bwd: mean required unless x holds xhat");
  COATI_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma/dbeta must both be given or both null");
  COATI_CHECK_SHAPE(M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && lddy % 4 == 0,
                    "layernorm_bwd: unsupported shape M=%d C=%d", M, C);
  const int nch = cdiv(C, 256);
  int blocks = cdiv(M, 4);
  if (blocks > 2048) blocks = 2048;
  dim3 grid(blocks), block(256);
#define LN_B(N, F) hipLaunchKernelGGL((ln_bwd_kernel<N, F>), grid, block, 0, s, dy, lddy, x, ldx, x_is_xhat, mean, rstd, gamma, dres, dx, dx16, dgamma, dbeta, partial, M, C)
  if (dy_f32) {
    if (nch == 1) LN_B(1, true); else if (nch == 2) LN_B(2, true); else if (nch == 3) LN_B(3, true); else LN_B(4, true);
  } else {
    if (nch == 1) LN_B(1, false); else if (nch == 2) LN_B(2, false); else if (nch == 3) LN_B(3, false); else LN_B(4, false);
  }
#undef LN_B
  COATI_LAUNCH_CHECK("layernorm_bwd");
  if (nblk_out) *nblk_out = blocks;
  return COATI_OK;
}

int launch_layernorm_bwd(const void* dy, int dy_f32, long long lddy, const float* x, long long ldx, int x_is_xhat,
                         const float* mean, const float* rstd, const float* gamma, const float* dres,
                         float* dx, bf16_t* dx16, float* dgamma, float* dbeta, float* partial, int M, int C, hipStream_t s) {
  int blocks = 0;
  const int rc = ln_bwd_stage1(dy, dy_f32, lddy, x, ldx, x_is_xhat, mean, rstd, gamma, dres, dx, dx16, dgamma, dbeta, partial, M, C, &blocks, s);
  if (rc != COATI_OK) return rc;
  if (dgamma && partial) {
    hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3(cdiv(2 * C, 64), 32), dim3(256), 0, s, partial, blocks, dgamma, dbeta, C);
    COATI_LAUNCH_CHECK("layernorm_bwd(finish)");
  }
  return COATI_OK;
}

int launch_layernorm_bwd_deferred(const void* dy, int dy_f32, long long lddy, const float* x, long long ldx, int x_is_xhat,
                                  const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                                  bf16_t* dx16, float* partial, int* nblk_out, int M, int C, hipStream_t s) {
  COATI_CHECK_ARG(partial && nblk_out, "layernorm_bwd_deferred: partial buffer missing");
  // any non-null dgamma / dbeta selects the partial-sum path of the kernel; they are not written there
  return ln_bwd_stage1(dy, dy_f32, lddy, x, ldx, x_is_xhat, mean, rstd, gamma, dres, dx, dx16, partial, partial, partial, M, C, nblk_out, s);
}

// stage 2 for a batch of LayerNorms: blockIdx.z = slot
__global__ __launch_bounds__(256) void ln_bwd_finish_batched_kernel(const float* __restrict__ partial, long long slot_stride, int nblk,
                                                                    float* __restrict__ grad_base, LnFinishBatch b, int C) {
  __shared__ float red[4][64];
  const int slot = blockIdx.z;
  const float* ps = partial + (long long)slot * slot_stride;
  if (b.nblk[slot] > 0) nblk = b.nblk[slot];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per;
  int r1 = r0 + per;
  if (r1 > nblk) r1 = nblk;
  float acc = 0.f;
  if (col < 2 * C) {
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 4) acc += ps[(long long)r * 2 * C + col];
  }
  red[rg][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rg == 0 && col < 2 * C) {
    const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(col < C ? grad_base + b.dg_off[slot] + col : grad_base + b.db_off[slot] + (col - C), t);
  }
}

int launch_ln_finish_batched(const float* partial, long long slot_stride, int nblk, float* grad_base, const LnFinishBatch& b,
                             int C, hipStream_t s) {
  COATI_CHECK_ARG(partial && grad_base && b.n > 0 && b.n <= COATI_LN_MAX_SLOTS, "ln_finish_batched: bad batch");
  hipLaunchKernelGGL(ln_bwd_finish_batched_kernel, dim3(cdiv(2 * C, 64), 32, b.n), dim3(256), 0, s, partial, slot_stride, nblk,
                     grad_base, b, C);
  COATI_LAUNCH_CHECK("ln_finish_batched");
  return COATI_OK;
}
