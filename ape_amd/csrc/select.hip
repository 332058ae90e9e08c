// Greedy NMS on the GPU (torchvision.ops.nms semantics: suppress when IoU > threshold, candidates visited in
// descending score order).  torchvision is not available on this target and NMS is on the hot path twice:
//   * encoder proposals: per-level NMS at IoU 0.9 over <= 5 x 1000 candidates
//     (ape/modeling/ape_deta/deformable_transformer_vl.py:592-597)
//   * final detections: class-wise NMS at IoU 0.7 over 900 x K (query, class) pairs whose boxes are class
//     agnostic (ape/modeling/ape_deta/fast_rcnn.py:192; deformable_detr_segm_vl.py:759-810) -- so ONE 900x900
//     IoU bit-matrix serves all K classes and every class is scanned by its own wavefront.
// Step 1 builds the suppression bit matrix (64x64 tiles, column boxes staged in LDS); step 2 is the sequential
// greedy scan, one 64-lane wave per independent problem, the "removed" bit-set living in registers (lane w
// owns words w and w+64) and the matrix rows prefetched 16 deep ahead of the dependent decision.
#include "common.h"
#include "../../include/ape_hip.h"

__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
  const float area_a = (a.z - a.x) * (a.w - a.y);
  const float area_b = (b.z - b.x) * (b.w - b.y);
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
  const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
  const float inter = w * h;
  return inter / (area_a + area_b - inter) > thr;
}

// mask[i][w] bit j: box (w*64+j) is suppressed by box i  (same group only; j != i)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, const int* __restrict__ groups, int n,
                                                      float thr, unsigned long long* __restrict__ mask, int nw) {
  __shared__ float4 cb[64];
  __shared__ int cg[64];
  const int rb = blockIdx.y, cbk = blockIdx.x, t = threadIdx.x;
  const int j0 = cbk * 64;
  if (j0 + t < n) {
    cb[t] = boxes[j0 + t];
    cg[t] = groups ? groups[j0 + t] : 0;
  }
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= n) return;
  const float4 bi = boxes[i];
  const int gi = groups ? groups[i] : 0;
  unsigned long long bits = 0ull;
  const int cnt = min(64, n - j0);
  for (int j = 0; j < cnt; ++j) {
    if (j0 + j != i && cg[j] == gi && iou_gt(bi, cb[j], thr)) bits |= (1ull << j);
  }
  mask[(size_t)i * nw + cbk] = bits;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane_uniform) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), lane_uniform);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane_uniform);
  return ((unsigned long long)hi << 32) | lo;
}

#define NMS_PF 16
// segments: wave g scans candidates [seg[g], seg[g+1]) in order; valid[i]==0 candidates are skipped entirely.
// The bit-matrix rows of chunk c+1 are fetched while chunk c is being decided (software pipelined).
struct NmsRows { unsigned long long r0[NMS_PF], r1[NMS_PF]; };

__device__ __forceinline__ void nms_load_rows(NmsRows& R, const unsigned long long* __restrict__ mask, int nw, int base, int s1,
                                              int w0, int nws, int lane) {
#pragma unroll
  for (int u = 0; u < NMS_PF; ++u) {
    const int i = base + u;
    R.r0[u] = (i < s1 && lane < nws) ? mask[(size_t)i * nw + w0 + lane] : 0ull;
    R.r1[u] = (i < s1 && lane + 64 < nws) ? mask[(size_t)i * nw + w0 + lane + 64] : 0ull;
  }
}

__device__ __forceinline__ void nms_decide_rows(const NmsRows& R, int base, int s1, int w0, int lane, const uint8_t* __restrict__ valid,
                                                uint8_t* __restrict__ keep, unsigned long long& rem0, unsigned long long& rem1) {
#pragma unroll
  for (int u = 0; u < NMS_PF; ++u) {
    const int i = base + u;
    if (i < s1) {  // wave-uniform
      const int word = (i >> 6) - w0, bit = i & 63;
      const unsigned long long rv = (word >= 64) ? rem1 : rem0;
      const unsigned long long r = readlane64(rv, word & 63);   // `word` is wave-uniform: v_readlane, not a bpermute
      const bool ok = (valid == nullptr) || (valid[i] != 0);
      const bool kept = ok && !((r >> bit) & 1ull);
      if (kept) { rem0 |= R.r0[u]; rem1 |= R.r1[u]; }
      if (lane == 0) keep[i] = kept ? 1 : 0;
    }
  }
}

__global__ __launch_bounds__(64) void nms_scan_segments_kernel(const unsigned long long* __restrict__ mask, int nw,
                                                               const int* __restrict__ seg, const uint8_t* __restrict__ valid,
                                                               uint8_t* __restrict__ keep) {
  const int lane = threadIdx.x;
  const int s0 = seg[blockIdx.x], s1 = seg[blockIdx.x + 1];
  if (s1 <= s0) return;
  const int w0 = s0 >> 6, w1 = (s1 - 1) >> 6;
  const int nws = w1 - w0 + 1;  // <= 128 (checked by the launcher through max segment length)
  unsigned long long rem0 = 0ull, rem1 = 0ull;
  NmsRows A, B;
  nms_load_rows(A, mask, nw, s0, s1, w0, nws, lane);
  for (int base = s0; base < s1; base += 2 * NMS_PF) {
    nms_load_rows(B, mask, nw, base + NMS_PF, s1, w0, nws, lane);
    nms_decide_rows(A, base, s1, w0, lane, valid, keep, rem0, rem1);
    nms_load_rows(A, mask, nw, base + 2 * NMS_PF, s1, w0, nws, lane);
    nms_decide_rows(B, base + NMS_PF, s1, w0, lane, valid, keep, rem0, rem1);
  }
}

// class-wise scan over shared boxes: the whole n x nw bit matrix is staged in LDS once per workgroup (900 boxes:
// 108 KB) and each of the 4 waves scans one class; per step only LDS + cross-lane traffic.
__global__ __launch_bounds__(256) void nms_scan_classes_kernel(const unsigned long long* __restrict__ mask, int nw, int n,
                                                               const int* __restrict__ order, const uint8_t* __restrict__ valid,
                                                               uint8_t* __restrict__ keep, int num_classes) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long smask[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n * nw; i += 256) smask[i] = mask[i];
  __syncthreads();
  const int c = blockIdx.x * 4 + wave;
  if (c >= num_classes) return;
  const int* ord = order + (size_t)c * n;
  unsigned long long rem = 0ull;  // lane w owns word w (n <= 4096 -> nw <= 64)
  for (int base = 0; base < n; base += 64) {
    const int p = base + lane;
    const int qv = p < n ? ord[p] : 0;
    const int okv = (p < n) && (valid == nullptr || valid[(size_t)c * n + p] != 0);
    int keptv = 0;
    const int cnt = min(64, n - base);
    for (int u = 0; u < cnt; ++u) {
      const int qi = __builtin_amdgcn_readlane(qv, u);     // u is wave-uniform
      const int ok = __builtin_amdgcn_readlane(okv, u);
      const unsigned long long row = lane < nw ? smask[(size_t)qi * nw + lane] : 0ull;
      const unsigned long long r = readlane64(rem, qi >> 6);
      const bool kept = ok && !((r >> (qi & 63)) & 1ull);
      if (kept) rem |= row;
      if (lane == u) keptv = kept ? 1 : 0;
    }
    if (p < n) keep[(size_t)c * n + p] = (uint8_t)keptv;
  }
}

extern "C" int ape_hip_nms_mask_words(int n) { return ceil_div(n, 64); }

extern "C" int ape_hip_nms_mask(const float* boxes, const int* groups, int n, float iou_thr, uint64_t* mask, void* stream) {
  APE_CHECK_ARG(boxes && mask && n > 0 && ((uintptr_t)boxes) % 16 == 0, "ape_hip_nms_mask: bad args");
  const int nw = ceil_div(n, 64);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nw, nw), dim3(64), 0, (hipStream_t)stream, (const float4*)boxes, groups, n, iou_thr,
                     (unsigned long long*)mask, nw);
  APE_CHECK_LAUNCH("ape_hip_nms_mask");
  return 0;
}

extern "C" int ape_hip_nms_scan_segments(const uint64_t* mask, int n, const int* seg_offsets, int num_segments, int max_segment,
                                         const uint8_t* valid, uint8_t* keep, void* stream) {
  APE_CHECK_ARG(mask && seg_offsets && keep && n > 0 && num_segments > 0, "ape_hip_nms_scan_segments: bad args");
  APE_CHECK_ARG(max_segment <= 126 * 64, "ape_hip_nms_scan_segments: segment longer than %d candidates", 126 * 64);
  hipLaunchKernelGGL(nms_scan_segments_kernel, dim3(num_segments), dim3(64), 0, (hipStream_t)stream,
                     (const unsigned long long*)mask, ceil_div(n, 64), seg_offsets, valid, keep);
  APE_CHECK_LAUNCH("ape_hip_nms_scan_segments");
  return 0;
}

extern "C" int ape_hip_nms_scan_classes(const uint64_t* mask, int n, const int* order, int num_classes, const uint8_t* valid,
                                        uint8_t* keep, void* stream) {
  const int nw = ceil_div(n, 64);
  const size_t lds = (size_t)n * nw * 8;
  APE_CHECK_ARG(mask && order && keep && n > 0 && num_classes > 0 && lds <= 160 * 1024,
                "ape_hip_nms_scan_classes: bad args (the n x n/64 bit matrix must fit the 160 KiB LDS: n <= 1280)");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)nms_scan_classes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(nms_scan_classes_kernel, dim3(ceil_div(num_classes, 4)), dim3(256), lds, (hipStream_t)stream,
                     (const unsigned long long*)mask, nw, n, order, valid, keep, num_classes);
  APE_CHECK_LAUNCH("ape_hip_nms_scan_classes");
  return 0;
}
