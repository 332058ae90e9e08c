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

// ---------------------------------------------------------------------------------------------------------------
// Per-query class scores of the semantic / panoptic branches (round 6: these were ~15 tensor-library launches per image inside the captured
// step -- sigmoid, division, softmax, max, comparisons, cat / min, zero-filled transposes).
//   stuff_collapse    : get_stuff_score (deformable_detr_segm_vl.py:1251-1271) with a leading "things" stuff class: the nt thing columns
//                       collapse into ONE column (their minimum), the stuff columns follow:  [Q, K] -> [Q, K - nt + 1]
//   sem_class_weights : _postprocess_semantic (:891-894) for the kept queries: w = softmax_c(sigmoid(logits[qidx[r], c]) / temp) * valid[r] (valid = the kept detection's score >= 0; NULL: all),
//                       written TRANSPOSED as the A operand of the class x query product: A[c, r] (16-bit or fp32), zero columns r >= k
//   pan_class_scores  : _postprocess_panoptic (:944-949) for the panoptic queries: (score, label) = max_c sigmoid(logits) -- or, with
//                       transform_eval, of softmax_c(sigmoid / temp) --, keep = valid & (max_c sigmoid > object_mask_threshold)
// One wave per query row; the class rows are L2 resident (K <= a few thousand); fp32 math, the same operation order as the tensor-level
// definitions (tests/ref_ops.py).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stuff_collapse_kernel(const float* __restrict__ x, int ldx, int Q, int K, int nt, float* __restrict__ out,
                                                             int ldo) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const float* r = x + (size_t)q * ldx;
  float m = INFINITY;
  for (int c = lane; c < nt; c += 64) m = fminf(m, r[c]);
  m = -wave_max(-m);
  float* o = out + (size_t)q * ldo;
  if (lane == 0) o[0] = m;
  for (int c = nt + lane; c < K; c += 64) o[c - nt + 1] = r[c];
}

extern "C" int ape_hip_stuff_collapse(const float* logits, int ldl, int Q, int K, int nt, float* out, int ldo, void* stream) {
  APE_CHECK_ARG(logits && out && Q > 0 && nt >= 1 && nt <= K && ldo >= K - nt + 1, "ape_hip_stuff_collapse: bad args");
  APE_LAUNCH(stuff_collapse_kernel, dim3(ceil_div(Q, 4)), dim3(256), 0, (hipStream_t)stream, logits, ldl, Q, K, nt, out, ldo);
  APE_CHECK_LAUNCH("ape_hip_stuff_collapse");
  return 0;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

template <typename TO>
__global__ __launch_bounds__(256) void sem_class_weights_kernel(const float* __restrict__ x, int ldx, const int64_t* __restrict__ qidx,
                                                                const float* __restrict__ valid, int k, int kp, int K, float temp,
                                                                TO* __restrict__ A, int lda) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= kp) return;
  if (r >= k) {                                                  // padding columns of the operand
    for (int c = lane; c < K; c += 64) stf<TO>(A + (size_t)c * lda + r, 0.f);
    return;
  }
  const float* row = x + (size_t)qidx[r] * ldx;
  float m = -INFINITY;
  for (int c = lane; c < K; c += 64) m = fmaxf(m, sigmoidf_(row[c]) / temp);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < K; c += 64) sum += expf(sigmoidf_(row[c]) / temp - m);
  sum = wave_sum(sum);
  const float sc = (valid == nullptr || valid[r] >= 0.f) ? 1.f / sum : 0.f;
  for (int c = lane; c < K; c += 64) stf<TO>(A + (size_t)c * lda + r, expf(sigmoidf_(row[c]) / temp - m) * sc);
}

extern "C" int ape_hip_sem_class_weights(const float* logits, int ldl, const int64_t* qidx, const float* valid, int k, int kp, int K,
                                         float temp, void* A, int lda, int out_dt, void* stream) {
  APE_CHECK_ARG(logits && qidx && A && k > 0 && kp >= k && K > 0 && lda >= kp, "ape_hip_sem_class_weights: bad args");
  const dim3 grid(ceil_div(kp, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (out_dt == APE_DT_F16) APE_LAUNCH(sem_class_weights_kernel<f16_t>, grid, block, 0, s, logits, ldl, qidx, valid, k, kp, K, temp, (f16_t*)A, lda);
  else if (out_dt == APE_DT_BF16) APE_LAUNCH(sem_class_weights_kernel<bf16_t>, grid, block, 0, s, logits, ldl, qidx, valid, k, kp, K, temp, (bf16_t*)A, lda);
  else if (out_dt == APE_DT_F32) APE_LAUNCH(sem_class_weights_kernel<float>, grid, block, 0, s, logits, ldl, qidx, valid, k, kp, K, temp, (float*)A, lda);
  else APE_CHECK_ARG(false, "ape_hip_sem_class_weights: bad out_dt %d", out_dt);
  APE_CHECK_LAUNCH("ape_hip_sem_class_weights");
  return 0;
}

__global__ __launch_bounds__(256) void pan_class_scores_kernel(const float* __restrict__ x, int ldx, const int64_t* __restrict__ qidx,
                                                               const float* __restrict__ valid, int k, int K, float thresh, int transform,
                                                               float temp, float* __restrict__ score, int64_t* __restrict__ label,
                                                               int32_t* __restrict__ label32, uint8_t* __restrict__ keep) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= k) return;
  const float* row = x + (size_t)(qidx != nullptr ? qidx[r] : r) * ldx;
  // max of sigmoid, first index on ties (torch.max)
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = lane; c < K; c += 64) {
    const float v = sigmoidf_(row[c]);
    if (v > m) { m = v; mi = c; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  const bool kp = (valid == nullptr || valid[r] >= 0.f) && (m > thresh);
  float sc = m;
  if (transform) {                                               // softmax(sigmoid / temp).max(-1): the same argmax, the score renormalised
    float sum = 0.f;
    for (int c = lane; c < K; c += 64) sum += expf(sigmoidf_(row[c]) / temp - m / temp);
    sum = wave_sum(sum);
    sc = 1.f / sum;
  }
  if (lane == 0) { score[r] = sc; label[r] = mi; if (label32 != nullptr) label32[r] = mi; keep[r] = kp ? 1 : 0; }
}

extern "C" int ape_hip_pan_class_scores(const float* logits, int ldl, const int64_t* qidx, const float* valid, int k, int K, float thresh,
                                        int transform, float temp, float* score, int64_t* label, int32_t* label32, uint8_t* keep, void* stream) {
  APE_CHECK_ARG(logits && score && label && keep && k > 0 && K > 0, "ape_hip_pan_class_scores: bad args");
  APE_LAUNCH(pan_class_scores_kernel, dim3(ceil_div(k, 4)), dim3(256), 0, (hipStream_t)stream, logits, ldl, qidx, valid, k, K, thresh, transform,
             temp, score, label, label32, keep);
  APE_CHECK_LAUNCH("ape_hip_pan_class_scores");
  return 0;
}

// argmax over the class axis of a [C, H * W] fp32 score volume -> int16 labels (the label map every semantic evaluator reduces the scores to;
// first index on ties like torch.argmax); `class0` (NaN = off) REPLACES the scores of class 0 (deformable_detr_segm_vl.py:654-663: the constant
// "things" logit of stuff-only evaluation).
__global__ __launch_bounds__(256) void argmax_labels_kernel(const float* __restrict__ x, size_t ldc, int C, size_t n, float class0,
                                                            int16_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float m = (class0 == class0) ? class0 : x[i];
  int mi = 0;
  for (int c = 1; c < C; ++c) {
    const float v = x[(size_t)c * ldc + i];
    if (v > m) { m = v; mi = c; }
  }
  out[i] = (int16_t)mi;
}

extern "C" int ape_hip_argmax_labels(const float* x, size_t ld_class, int C, size_t n, float class0, int16_t* out, void* stream) {
  APE_CHECK_ARG(x && out && C > 0 && C < 32768 && n > 0 && ld_class >= n, "ape_hip_argmax_labels: bad args");
  APE_LAUNCH(argmax_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld_class, C, n, class0, out);
  APE_CHECK_LAUNCH("ape_hip_argmax_labels");
  return 0;
}
