// Batch tail of clip_ar_xform on the device (reference clip_e2e.py:312-329):
//   tokens = tokens[:, : (tokens.sum(0) > 0).sum()]       -> coati_batch_ncols
//   y_next[:, :-1] = tokens[:, 1:]; y_next[:, -1] = 0; y_next[y_next in {clip, pad, unk, suffix, middle}] = -1
// Integer work, HBM-bound and tiny ([B, n_seq] int64): one pass each.
#include "kernels.h"

// ncols = number of columns whose sum over the batch is > 0 (token ids are >= 0: "any non-zero entry").
// one workgroup; thread = column (strided), rows walked with coalesced 8-B loads across the threads of a row
__global__ __launch_bounds__(256) void batch_ncols_kernel(const long long* __restrict__ tok, int B, int S, int* __restrict__ out) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int mine = 0;
  for (int c = threadIdx.x; c < S; c += blockDim.x) {
    long long sum = 0;
    for (int b = 0; b < B; ++b) sum += tok[(long long)b * S + c];
    mine += sum > 0 ? 1 : 0;
  }
  if (mine) atomicAdd(&cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) out[0] = cnt;
}

int launch_batch_ncols(const long long* tok, int B, int S, int* ncols, hipStream_t s) {
  COATI_CHECK_ARG(tok && ncols, "batch_ncols: null operand");
  COATI_CHECK_SHAPE(B > 0 && S > 0, "batch_ncols: empty batch");
  hipLaunchKernelGGL(batch_ncols_kernel, dim3(1), dim3(256), 0, s, tok, B, S, ncols);
  COATI_LAUNCH_CHECK("batch_ncols");
  return COATI_OK;
}

// compaction to [B, ncol] (+ optional y_next): thread = output element
__global__ __launch_bounds__(256) void batch_tail_kernel(const long long* __restrict__ tok, int B, int S, int ncol,
                                                         long long* __restrict__ tok_out, long long* __restrict__ y_out,
                                                         const long long* __restrict__ masked, int n_masked) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)B * ncol) return;
  const int b = (int)(e / ncol), t = (int)(e - (long long)b * ncol);
  tok_out[e] = tok[(long long)b * S + t];
  if (y_out) {
    long long y = (t + 1 < ncol) ? tok[(long long)b * S + t + 1] : 0;
    for (int i = 0; i < n_masked; ++i)
      if (y == masked[i]) y = -1;
    y_out[e] = y;
  }
}

int launch_batch_tail(const long long* tok, int B, int S, int ncol, long long* tok_out, long long* y_out,
                      const long long* masked, int n_masked, hipStream_t s) {
  COATI_CHECK_ARG(tok && tok_out && (masked || n_masked == 0), "batch_tail: null operand");
  COATI_CHECK_SHAPE(B > 0 && S > 0 && ncol > 0 && ncol <= S && n_masked >= 0, "batch_tail: bad shape B=%d S=%d ncol=%d", B, S, ncol);
  const long long n = (long long)B * ncol;
  hipLaunchKernelGGL(batch_tail_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, tok, B, S, ncol, tok_out, y_out, masked, n_masked);
  COATI_LAUNCH_CHECK("batch_tail");
  return COATI_OK;
}
