// Iterative box refinement of the deformable decoder (deformable_transformer_vl.py:232-246, with_box_refine):
//     new_ref = sigmoid(delta + inverse_sigmoid(ref))          detrex.utils.inverse_sigmoid, eps = 1e-3
//     ref_in  = new_ref[:, None, :] * cat(valid_ratios, valid_ratios)[None]   (:203-210, the next layer's MSDA reference)
// One tiny kernel instead of nine elementwise launches per decoder layer (the decoder is launch-latency bound).
#include "common.h"
#include "../../include/ape_hip.h"

__global__ __launch_bounds__(256) void box_refine_kernel(const float* __restrict__ delta, int ldd, const float* __restrict__ ref,
                                                         const float* __restrict__ vr4, int L, int Q, float eps,
                                                         float* __restrict__ new_ref, float* __restrict__ ref_in) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= Q * 4) return;
  const int q = gid >> 2, c = gid & 3;
  float x = ref[gid];
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
  float z = logf(x1 / x2);
  if (delta != nullptr) z += delta[(size_t)q * ldd + c];
  const float r = delta != nullptr ? 1.f / (1.f + expf(-z)) : ref[gid];
  if (new_ref != nullptr) new_ref[gid] = r;
  if (ref_in != nullptr)
    for (int l = 0; l < L; ++l) ref_in[((size_t)q * L + l) * 4 + c] = r * vr4[l * 4 + c];
}

// delta [Q,4] fp32 (row stride ldd) or NULL (then new_ref = ref: only the per-level reference is produced),
// ref [Q,4], vr4 [L,4] -> new_ref [Q,4] (may be NULL), ref_in [Q,L,4] (may be NULL)
extern "C" int ape_hip_box_refine(const float* delta, int ldd, const float* ref, const float* vr4, int L, int Q, float eps,
                                  float* new_ref, float* ref_in, void* stream) {
  APE_CHECK_ARG(ref && Q > 0 && (ref_in == nullptr || (vr4 != nullptr && L > 0)) && (new_ref || ref_in),
                "ape_hip_box_refine: bad args");
  hipLaunchKernelGGL(box_refine_kernel, dim3(ceil_div(Q * 4, 256)), dim3(256), 0, (hipStream_t)stream, delta, ldd, ref, vr4, L, Q, eps,
                     new_ref, ref_in);
  APE_CHECK_LAUNCH("ape_hip_box_refine");
  return 0;
}
