// Language-side of the bi-directional vision-language attention for a SINGLE text token (name-prompt mode,
// L = 1: deformable_detr_segm_vl.py:349-352 feeds one zero token; ape/layers/fuse_helper.py:67-166).
//
// With one text token the [8, T, 1] score tensor is a [T, 8] matrix S, the vision side's softmax over L is
// identically 1, and the language side is an attention POOL over all T vision tokens per head:
//     p[t,h]   = softmax_t( clamp( clamp(S - max(S)) - max_t(...) ) )          (fuse_helper.py:89-111, 116)
//     pooled_h = sum_t p[t,h] * x[t,:]                                           (bmm at :140, before values_v_proj,
//                                                                                 which is linear and applied after)
// i.e. a [8,T] x [T,256] reduction: three small kernels (chunk maxima, chunk partial sums with the shared
// maxima, final combine).  Vision padding is NOT masked (use_attention_mask_v=False), exactly like the reference.
#include "common.h"
#include "../../include/ape_hip.h"

#define VL_CHUNK 128
#define VL_H 8

__global__ __launch_bounds__(256) void vl_smax_partial_kernel(const float* __restrict__ S, int lds, int T, float* __restrict__ pmax) {
  __shared__ float sh[256];
  const int t = threadIdx.x, h = t & 7;
  const int base = blockIdx.x * VL_CHUNK;
  float m = -INFINITY;
  for (int k = 0; k < VL_CHUNK / 32; ++k) {
    const int tok = base + k * 32 + (t >> 3);
    if (tok < T) m = fmaxf(m, S[(size_t)tok * lds + h]);
  }
  sh[t] = m;
  __syncthreads();
  if (t < VL_H) {
    float a = -INFINITY;
    for (int j = t; j < 256; j += 8) a = fmaxf(a, sh[j]);
    pmax[blockIdx.x * VL_H + t] = a;
  }
}

template <typename TX>
__global__ __launch_bounds__(256) void vl_pool_partial_kernel(const float* __restrict__ S, int lds, const TX* __restrict__ x, int ldx,
                                                              int T, int C, const float* __restrict__ pmax, int nchunk,
                                                              float* __restrict__ pacc, float* __restrict__ psum) {
  __shared__ float hmax[VL_H];
  __shared__ float gmax_s;
  __shared__ float P[VL_CHUNK][VL_H];
  const int t = threadIdx.x;
  {
    // per-head maxima over all chunks: 32 threads per head, LDS tree
    float a = -INFINITY;
    for (int k = t >> 3; k < nchunk; k += 32) a = fmaxf(a, pmax[k * VL_H + (t & 7)]);
    P[t >> 3][t & 7] = a;
    __syncthreads();
    if (t < VL_H) {
      float m = -INFINITY;
      for (int j = 0; j < 32; ++j) m = fmaxf(m, P[j][t]);
      hmax[t] = m;
    }
  }
  __syncthreads();
  if (t == 0) {
    float g = hmax[0];
    for (int h = 1; h < VL_H; ++h) g = fmaxf(g, hmax[h]);
    gmax_s = g;
  }
  __syncthreads();
  const float gmax = gmax_s;
  const int base = blockIdx.x * VL_CHUNK;
  const int cnt = min(VL_CHUNK, T - base);
  // probabilities (unnormalised) for this chunk, reference clamp sequence
  for (int idx = t; idx < VL_CHUNK * VL_H; idx += 256) {
    const int r = idx >> 3, h = idx & 7;
    float p = 0.f;
    if (r < cnt) {
      float a = S[(size_t)(base + r) * lds + h] - gmax;
      a = fminf(fmaxf(a, -50000.f), 50000.f);
      const float mh = fminf(fmaxf(hmax[h] - gmax, -50000.f), 50000.f);
      float b = a - mh;
      b = fminf(fmaxf(b, -50000.f), 50000.f);
      p = expf(b);
    }
    P[r][h] = p;
  }
  __syncthreads();
  if (t < VL_H) {
    float s = 0.f;
    for (int r = 0; r < cnt; ++r) s += P[r][t];
    psum[blockIdx.x * VL_H + t] = s;
  }
  for (int c = t; c < C; c += 256) {
    float acc[VL_H];
#pragma unroll
    for (int h = 0; h < VL_H; ++h) acc[h] = 0.f;
    for (int r = 0; r < cnt; ++r) {
      const float xv = ldf<TX>(x + (size_t)(base + r) * ldx + c);
#pragma unroll
      for (int h = 0; h < VL_H; ++h) acc[h] = fmaf(P[r][h], xv, acc[h]);
    }
#pragma unroll
    for (int h = 0; h < VL_H; ++h) pacc[((size_t)blockIdx.x * VL_H + h) * C + c] = acc[h];
  }
}

// grid (8 heads, C/32): 8 k-groups x 32 columns per block, LDS tree over the k-groups
__global__ __launch_bounds__(256) void vl_pool_final_kernel(const float* __restrict__ pacc, const float* __restrict__ psum, int nchunk,
                                                            int C, const float* __restrict__ sub, float* __restrict__ out) {
  __shared__ float sa[8][32];
  __shared__ float sl[256];
  const int h = blockIdx.x, t = threadIdx.x;
  const int kg = t >> 5, cl = t & 31;
  const int c = blockIdx.y * 32 + cl;
  float l = 0.f;
  for (int k = t; k < nchunk; k += 256) l += psum[k * VL_H + h];
  sl[t] = l;
  float a = 0.f;
  if (c < C) for (int k = kg; k < nchunk; k += 8) a += pacc[((size_t)k * VL_H + h) * C + c];
  sa[kg][cl] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) sl[t] += sl[t + o]; __syncthreads(); }
  if (t < 32 && c < C) {
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) acc += sa[g][t];
    out[h * C + c] = acc / sl[0] - (sub != nullptr ? sub[c] : 0.f);
  }
}

extern "C" int ape_hip_vl_pool_workspace_floats(int T, int C) {
  const int nchunk = ceil_div(T, VL_CHUNK);
  return nchunk * VL_H * (C + 2);
}

// S [T, 8] fp32 scores (already scaled), x [T, C] -> out [8, C] fp32: softmax-over-T weighted mean of x per head, minus the
// vector sub [C] when given (the pooled rows are taken on x - sub: the folded gamma_v * delta_v of layers/fuse_helper.py)
extern "C" int ape_hip_vl_pool(const float* S, int lds, const void* x, int ldx, int x_dt, int T, int C, float* workspace,
                               const float* sub, float* out, void* stream) {
  APE_CHECK_ARG(S && x && workspace && out && T > 0 && C > 0, "ape_hip_vl_pool: bad args");
  const int nchunk = ceil_div(T, VL_CHUNK);
  float* pmax = workspace;
  float* psum = pmax + nchunk * VL_H;
  float* pacc = psum + nchunk * VL_H;
  hipStream_t s = (hipStream_t)stream;
  APE_LAUNCH(vl_smax_partial_kernel, dim3(nchunk), dim3(256), 0, s, S, lds, T, pmax);
  if (x_dt == APE_DT_F16) APE_LAUNCH(vl_pool_partial_kernel<f16_t>, dim3(nchunk), dim3(256), 0, s, S, lds, (const f16_t*)x, ldx, T, C, pmax, nchunk, pacc, psum);
  else if (x_dt == APE_DT_BF16) APE_LAUNCH(vl_pool_partial_kernel<bf16_t>, dim3(nchunk), dim3(256), 0, s, S, lds, (const bf16_t*)x, ldx, T, C, pmax, nchunk, pacc, psum);
  else
    APE_LAUNCH(vl_pool_partial_kernel<float>, dim3(nchunk), dim3(256), 0, s, S, lds, (const float*)x, ldx, T, C, pmax, nchunk, pacc, psum);
  APE_LAUNCH(vl_pool_final_kernel, dim3(VL_H, ceil_div(C, 32)), dim3(256), 0, s, pacc, psum, nchunk, C, sub, out);
  APE_CHECK_LAUNCH("ape_hip_vl_pool");
  return 0;
}


// ------------------------------------------------------------------------------------------
// Per-head matrix-vector products of the single-token language side (ape/layers/fuse_helper.py:70-73,140,160-161 after the
// reassociation of layers/fuse_helper.py): out[h][n] = alpha * sum_d x[h][d] * W[h][n][d] + bias[h][n], all fp32
// (optionally a 16-bit copy of out, bf16 or f16: the GEMM operand of the score product).
// One wave per (h, n); replaces torch.einsum / matmul (rocBLAS launches inside the captured forward).
// ------------------------------------------------------------------------------------------
template <typename TC>
__global__ __launch_bounds__(256) void head_gemv_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out, int ldo, int H,
                                                        int N, int D, float alpha, TC* __restrict__ out_bf16, int ldob) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);        // over H * N
  if (row >= H * N) return;
  const int h = row / N, n = row % N;
  const float* w = W + (size_t)row * D;
  const float* xr = x + (size_t)h * ldx;
  float s = 0.f;
  if ((D & 3) == 0 && (ldx & 3) == 0) {
    for (int d = lane * 4; d < D; d += 256) {
      const float4 a = *reinterpret_cast<const float4*>(w + d), b = *reinterpret_cast<const float4*>(xr + d);
      s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
    }
  } else {
    for (int d = lane; d < D; d += 64) s = fmaf(w[d], xr[d], s);
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float r = s * alpha + (bias != nullptr ? bias[row] : 0.f);
    out[(size_t)h * ldo + n] = r;
    if (out_bf16 != nullptr) stf<TC>(out_bf16 + (size_t)h * ldob + n, r);
  }
}

extern "C" int ape_hip_head_gemv(const float* x, int ldx, const float* W, const float* bias, float* out, int ldo, int H, int N,
                                 int D, float alpha, void* out_bf16, int ldob, int copy_dt, void* stream) {
  APE_CHECK_ARG(x && W && out && H > 0 && N > 0 && D > 0, "ape_hip_head_gemv: bad args");
  APE_CHECK_ARG(out_bf16 == nullptr || ape_is16(copy_dt), "ape_hip_head_gemv: the copy is bf16 or f16 (copy_dt %d)", copy_dt);
  APE_CHECK_ARG(((uintptr_t)x) % 16 == 0 && ((uintptr_t)W) % 16 == 0, "ape_hip_head_gemv: x / W must be 16-byte aligned");
  if (out_bf16 != nullptr && copy_dt == APE_DT_F16)
    APE_LAUNCH(head_gemv_kernel<f16_t>, dim3(ceil_div(H * N, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, W, bias, out, ldo, H, N, D, alpha,
                       (f16_t*)out_bf16, ldob);
  else
    APE_LAUNCH(head_gemv_kernel<bf16_t>, dim3(ceil_div(H * N, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, W, bias, out, ldo, H, N, D, alpha,
                       (bf16_t*)out_bf16, ldob);
  APE_CHECK_LAUNCH("ape_hip_head_gemv");
  return 0;
}
