// Softmax kernels of the DENSE (multi-token) bi-directional vision-language attention
// (ape/layers/fuse_helper.py:84-131; phrase / expression prompts fuse against L >= 256 text tokens).
// The score tensor is kept as ONE matrix S [T, H*L] (token-major; column (h, l)), produced by a single GEMM.
//   segment_softmax  : vision side, softmax over the L columns of each head segment of every row
//                      (attn_weights.softmax(dim=-1), :131), after the global-max subtraction and clamps (:89-99)
//   colstats         : language side statistics over the T rows of every column (max and sum of exp; :101-116)
//   colsoftmax_t     : writes softmax_T(S)^T as [H*L, T] (the A operand of the token-reduction GEMM), or a plain
//                      transposed copy when no statistics are given
// Wave-shuffle reductions, fp32 math, online (max, sum) merging for the column statistics.
#include "common.h"
#include "../../include/ape_hip.h"

#define CLAMP5E4(x) fminf(fmaxf((x), -50000.f), 50000.f)

// one wave per (row, segment); 3 passes over the L values (L2 resident)
template <typename TO>
__global__ __launch_bounds__(256) void segment_softmax_kernel(const float* __restrict__ S, int lds, int T, int nseg, int L,
                                                              const float* __restrict__ gmax_p, TO* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const size_t task = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task >= (size_t)T * nseg) return;
  const int row = (int)(task / nseg), seg = (int)(task % nseg);
  const float gmax = *gmax_p;
  const float* s = S + (size_t)row * lds + (size_t)seg * L;
  TO* o = out + (size_t)row * ldo + (size_t)seg * L;
  float m = -INFINITY;
  for (int c = lane; c < L; c += 64) m = fmaxf(m, CLAMP5E4(s[c] - gmax));
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < L; c += 64) sum += expf(CLAMP5E4(s[c] - gmax) - m);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int c = lane; c < L; c += 64) stf<TO>(o + c, expf(CLAMP5E4(s[c] - gmax) - m) * inv);
}

extern "C" int ape_hip_segment_softmax(const float* S, int lds, int T, int nseg, int L, const float* gmax, void* out, int ldo,
                                       int out_dt, void* stream) {
  APE_CHECK_ARG(S && gmax && out && T > 0 && nseg > 0 && L > 0, "ape_hip_segment_softmax: bad args");
  const size_t tasks = (size_t)T * nseg;
  const dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
  if (out_dt == APE_DT_F16) APE_LAUNCH(segment_softmax_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, S, lds, T, nseg, L, gmax, (f16_t*)out, ldo);
  else if (out_dt == APE_DT_BF16) APE_LAUNCH(segment_softmax_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, S, lds, T, nseg, L, gmax, (bf16_t*)out, ldo);
  else APE_LAUNCH(segment_softmax_kernel<float>, grid, block, 0, (hipStream_t)stream, S, lds, T, nseg, L, gmax, (float*)out, ldo);
  APE_CHECK_LAUNCH("ape_hip_segment_softmax");
  return 0;
}

// column statistics over rows: per 256-row chunk an online (max, sum) pair per column, then a merge kernel
#define CS_ROWS 256
__global__ __launch_bounds__(256) void colstats_partial_kernel(const float* __restrict__ S, int lds, int T, int C,
                                                               const float* __restrict__ gmax_p, float* __restrict__ part) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const float gmax = *gmax_p;
  const int r0 = blockIdx.x * CS_ROWS, r1 = min(r0 + CS_ROWS, T);
  float m = -INFINITY, s = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float a = CLAMP5E4(S[(size_t)r * lds + c] - gmax);
    if (a > m) { s = s * expf(m - a) + 1.f; m = a; } else { s += expf(a - m); }
  }
  part[((size_t)blockIdx.x * C + c) * 2 + 0] = m;
  part[((size_t)blockIdx.x * C + c) * 2 + 1] = s;
}

__global__ __launch_bounds__(256) void colstats_merge_kernel(const float* __restrict__ part, int nchunk, int C, float* __restrict__ colmax,
                                                             float* __restrict__ colsum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float m = -INFINITY, s = 0.f;
  for (int k = 0; k < nchunk; ++k) {
    const float mk = part[((size_t)k * C + c) * 2 + 0], sk = part[((size_t)k * C + c) * 2 + 1];
    if (mk > m) { s = s * expf(m - mk) + sk; m = mk; } else { s += sk * expf(mk - m); }
  }
  colmax[c] = m;
  colsum[c] = s;
}

extern "C" int ape_hip_colstats_workspace_floats(int T, int C) { return ceil_div(T, CS_ROWS) * C * 2; }

extern "C" int ape_hip_colstats(const float* S, int lds, int T, int C, const float* gmax, float* workspace, float* colmax,
                                float* colsum, void* stream) {
  APE_CHECK_ARG(S && gmax && workspace && colmax && colsum && T > 0 && C > 0, "ape_hip_colstats: bad args");
  const int nchunk = ceil_div(T, CS_ROWS);
  hipStream_t s = (hipStream_t)stream;
  APE_LAUNCH(colstats_partial_kernel, dim3(nchunk, ceil_div(C, 256)), dim3(256), 0, s, S, lds, T, C, gmax, workspace);
  APE_LAUNCH(colstats_merge_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, workspace, nchunk, C, colmax, colsum);
  APE_CHECK_LAUNCH("ape_hip_colstats");
  return 0;
}

// out[c][t] = f(S[t][c]); 64x64 tiles through LDS.  With colmax/colsum: f = exp(clamp(clamp(x - gmax) - colmax[c])) / colsum[c]
// (softmax over rows, transposed); without: f = identity (plain transposed copy).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ S, int lds, int T, int C, const float* __restrict__ gmax_p,
                                                        const float* __restrict__ colmax, const float* __restrict__ colsum,
                                                        TO* __restrict__ out, int ldo) {
  __shared__ float tile[64][65];
  const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float gmax = gmax_p ? *gmax_p : 0.f;
  for (int r = ty; r < 64; r += 4) {
    const int t = t0 + r, c = c0 + tx;
    float v = 0.f;
    if (t < T && c < C) {
      v = ldf<TI>(S + (size_t)t * lds + c);
      if (colmax != nullptr) v = expf(CLAMP5E4(CLAMP5E4(v - gmax) - colmax[c])) / colsum[c];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, t = t0 + tx;
    if (c < C && t < T) stf<TO>(out + (size_t)c * ldo + t, tile[tx][r]);
  }
}

extern "C" int ape_hip_transpose(const void* S, int lds, int in_dt, int T, int C, const float* gmax, const float* colmax,
                                 const float* colsum, void* out, int ldo, int out_dt, void* stream) {
  APE_CHECK_ARG(S && out && T > 0 && C > 0 && ((colmax == nullptr) == (colsum == nullptr)), "ape_hip_transpose: bad args");
  APE_CHECK_ARG(colmax == nullptr || gmax != nullptr, "ape_hip_transpose: softmax mode needs gmax");
  const dim3 grid(ceil_div(T, 64), ceil_div(C, 64)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int hk = APE_H16_KIND(in_dt, out_dt);
  if (hk < 0) { ape_set_error("ape_hip_transpose: dtypes must be f32 or ONE 16-bit type (in %d, out %d)", in_dt, out_dt); return -1; }
  const int key = (ape_is16(in_dt) ? 2 : 0) + (ape_is16(out_dt) ? 1 : 0);
  if (key == 0) { APE_LAUNCH((transpose_kernel<float, float>), grid, block, 0, s, (const float*)S, lds, T, C, gmax, colmax, colsum, (float*)out, ldo); }
  else if (hk == APE_DT_F16) {
    if (key == 1) { APE_LAUNCH((transpose_kernel<float, f16_t>), grid, block, 0, s, (const float*)S, lds, T, C, gmax, colmax, colsum, (f16_t*)out, ldo); } else if (key == 2) { APE_LAUNCH((transpose_kernel<f16_t, float>), grid, block, 0, s, (const f16_t*)S, lds, T, C, gmax, colmax, colsum, (float*)out, ldo); } else { APE_LAUNCH((transpose_kernel<f16_t, f16_t>), grid, block, 0, s, (const f16_t*)S, lds, T, C, gmax, colmax, colsum, (f16_t*)out, ldo); }
  } else {
    if (key == 1) { APE_LAUNCH((transpose_kernel<float, bf16_t>), grid, block, 0, s, (const float*)S, lds, T, C, gmax, colmax, colsum, (bf16_t*)out, ldo); } else if (key == 2) { APE_LAUNCH((transpose_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)S, lds, T, C, gmax, colmax, colsum, (float*)out, ldo); } else { APE_LAUNCH((transpose_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)S, lds, T, C, gmax, colmax, colsum, (bf16_t*)out, ldo); }
  }
  APE_CHECK_LAUNCH("ape_hip_transpose");
  return 0;
}
