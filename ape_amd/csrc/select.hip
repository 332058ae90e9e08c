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

// segments: wave g scans candidates [seg[g], seg[g+1]) in index order; valid[i]==0 candidates are skipped entirely.
// Chunked greedy scan, one 64-candidate word at a time:
//   1. removed-word of the chunk = OR over the survivors of EARLIER chunks of their row's word for this chunk: lane j
//      tests bit j of each earlier chunk's survivor mask and loads at most one word per earlier chunk -- all loads are
//      independent, then a wave-wide OR (the one-candidate-per-step scan paid ~250 ns of load latency per candidate);
//   2. lane j holds the word of row j that covers the chunk itself; the sequential part walks only over the candidates
//      that SURVIVE (find-first-set on the alive mask, one uniform v_readlane per survivor).
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffull));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
  unsigned lo = (unsigned)(v & 0xffffffffull), hi = (unsigned)(v >> 32);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo |= (unsigned)__shfl_xor((int)lo, o, 64); hi |= (unsigned)__shfl_xor((int)hi, o, 64); }
  return ((unsigned long long)hi << 32) | lo;
}

// USE_LDS: the segment's sub-matrix (rows s0..s1-1 x words w0..w1, 125 KiB for 1000 candidates) is first copied to LDS by
// the whole workgroup (bulk, pipelined loads); the scan, which is a chain of dependent reads, then never waits on
// global memory (with 5 waves on the GPU every such read used to be a ~2 us L2 miss).
template <bool USE_LDS>
__global__ __launch_bounds__(256) void nms_scan_segments_kernel(const unsigned long long* __restrict__ mask, int nw, int n,
                                                                const int* __restrict__ seg, const uint8_t* __restrict__ valid,
                                                                uint8_t* __restrict__ keep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sub[];
  const int s0 = seg[blockIdx.x], s1 = seg[blockIdx.x + 1];
  if (s1 <= s0) return;
  const int w0 = s0 >> 6, w1 = (s1 - 1) >> 6;   // <= 128 words (checked by the launcher through max segment length)
  const int nws = w1 - w0 + 1;
  const int r_base = w0 * 64;                    // sub-matrix row 0 = candidate w0*64 (chunk aligned)
  if (USE_LDS) {
    const int nrows = (w1 + 1) * 64 - r_base;
    for (int e = threadIdx.x; e < nrows * nws; e += 256) {
      const int r = e / nws, w = e - r * nws;
      const int row = r_base + r;
      sub[e] = row < n ? mask[(size_t)row * nw + w0 + w] : 0ull;
    }
    __syncthreads();
  }
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  auto word = [&](int row, int c) -> unsigned long long {      // mask[row][c] for rows / words of this segment
    if (USE_LDS) return sub[(size_t)(row - r_base) * nws + (c - w0)];
    return mask[(size_t)(row < n ? row : n - 1) * nw + c];
  };
  unsigned long long kept0 = 0ull, kept1 = 0ull;   // lane w stores the survivor mask of chunk w0 + w (and w0 + w + 64)
  for (int c = w0; c <= w1; ++c) {
    // 1. what the survivors so far remove from this chunk
    unsigned long long acc = 0ull;
    for (int cp = w0; cp < c; cp += 8) {                           // 8 independent reads per step
      unsigned long long t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int cq = cp + u;
        const int cqc = cq < c ? cq : cp;                          // uniform; out-of-range steps re-read `cp` and are masked
        const int rel = cqc - w0;
        const unsigned long long km = readlane64(rel >= 64 ? kept1 : kept0, rel & 63);
        const bool take = cq < c && ((km >> lane) & 1ull);
        const unsigned long long v = word(cqc * 64 + lane, c);     // unconditional read, masked afterwards (predicated loads
        t[u] = take ? v : 0ull;                                    //  made hipcc wait on each one)
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc |= t[u];
    }
    // readfirstlane tells hipcc the value is wave-uniform, so the resolve loop below runs on the scalar unit
    // (s_ff1 / s_andn2 + one v_readlane pair per survivor) instead of a ~20-instruction VALU chain per step
    const unsigned long long remc = uniform64(wave_or64(acc));
    // 2. resolve the chunk
    const int i = c * 64 + lane;                                   // this lane's candidate
    const bool in_seg = i >= s0 && i < s1;
    const bool ok = in_seg && (valid == nullptr || valid[i] != 0);
    const unsigned long long row_c = ok ? word(i, c) : 0ull;       // intra-chunk suppression word of row i
    unsigned long long alive = __builtin_amdgcn_ballot_w64(ok) & ~remc;
    unsigned long long kept = 0ull;
    while (alive != 0ull) {                                        // uniform loop over the survivors of this chunk
      const int j = __builtin_ctzll(alive);
      kept |= 1ull << j;
      alive &= ~(1ull << j);
      alive &= ~readlane64(row_c, j);
    }
    if (in_seg) keep[i] = (uint8_t)((kept >> lane) & 1ull);
    const int rel = c - w0;
    if (lane == (rel & 63)) { if (rel >= 64) kept1 = kept; else kept0 = kept; }
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
  APE_LAUNCH(nms_mask_kernel, dim3(nw, nw), dim3(64), 0, (hipStream_t)stream, (const float4*)boxes, groups, n, iou_thr,
                     (unsigned long long*)mask, nw);
  APE_CHECK_LAUNCH("ape_hip_nms_mask");
  return 0;
}

extern "C" int ape_hip_nms_scan_segments(const uint64_t* mask, int n, const int* seg_offsets, int num_segments, int max_segment,
                                         const uint8_t* valid, uint8_t* keep, void* stream) {
  APE_CHECK_ARG(mask && seg_offsets && keep && n > 0 && num_segments > 0, "ape_hip_nms_scan_segments: bad args");
  APE_CHECK_ARG(max_segment <= 126 * 64, "ape_hip_nms_scan_segments: segment longer than %d candidates", 126 * 64);
  // LDS copy of a segment's sub-matrix: (max_segment rounded out to chunks + 64) rows x (max_segment / 64 + 2) words
  const size_t sub_rows = (size_t)(ceil_div(max_segment, 64) + 1) * 64, sub_words = (size_t)ceil_div(max_segment, 64) + 1;
  const size_t lds = sub_rows * sub_words * 8;
  if (lds <= 160 * 1024) {
    static ApeOncePerDevice attr_done;
    if (attr_done.first()) {
      (void)hipFuncSetAttribute((const void*)nms_scan_segments_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    APE_LAUNCH(nms_scan_segments_kernel<true>, dim3(num_segments), dim3(256), lds, (hipStream_t)stream,
                       (const unsigned long long*)mask, ceil_div(n, 64), n, seg_offsets, valid, keep);
  } else {
    APE_LAUNCH(nms_scan_segments_kernel<false>, dim3(num_segments), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)mask, ceil_div(n, 64), n, seg_offsets, valid, keep);
  }
  APE_CHECK_LAUNCH("ape_hip_nms_scan_segments");
  return 0;
}

extern "C" int ape_hip_nms_scan_classes(const uint64_t* mask, int n, const int* order, int num_classes, const uint8_t* valid,
                                        uint8_t* keep, void* stream) {
  const int nw = ceil_div(n, 64);
  const size_t lds = (size_t)n * nw * 8;
  APE_CHECK_ARG(mask && order && keep && n > 0 && num_classes > 0 && lds <= 160 * 1024,
                "ape_hip_nms_scan_classes: bad args (the n x n/64 bit matrix must fit the 160 KiB LDS: n <= 1280)");
  static ApeOncePerDevice attr_done;
  if (attr_done.first()) {
    (void)hipFuncSetAttribute((const void*)nms_scan_classes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  APE_LAUNCH(nms_scan_classes_kernel, dim3(ceil_div(num_classes, 4)), dim3(256), lds, (hipStream_t)stream,
                     (const unsigned long long*)mask, nw, n, order, valid, keep, num_classes);
  APE_CHECK_LAUNCH("ape_hip_nms_scan_classes");
  return 0;
}
