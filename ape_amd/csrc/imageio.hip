// imageio.hip -- the byte / integer work on either side of the forward pass, gfx950.
//
//  * resize_u8_kernel: the predictor's test-time resize (ape/engine/defaults.py:213-222 ->
//    ResizeShortestEdge -> PIL Image.resize(BILINEAR)) bit for bit: a separable triangle filter with 22-bit fixed-point
//    coefficients, horizontal pass rounded to uint8, then vertical pass rounded to uint8.  One workgroup owns a
//    TH x 64 output tile: it resamples the source rows the tile needs horizontally into LDS (uint8, the rounding point of
//    the two-pass definition), then runs the vertical pass out of LDS and writes either uint8 HWC or the model's input
//    format (float32 CHW, optional BGR -> RGB flip).  The source is read once (plus the vertical filter overlap between
//    neighbouring tiles), the intermediate image never exists in HBM.  HBM bound: bytes = H*W*3 in + out.
//  * rle_*: COCO run-length encoding of the pasted instance masks (instances_to_coco_json,
//    ape/evaluation/d3_evaluation.py:441-493 -> pycocotools rleEncode): column-major runs.  A lane walks a 128-row
//    segment of one column (lanes = adjacent columns, so loads coalesce in the row-major mask), counts the value changes,
//    a per-mask scan turns the counts into offsets, a second walk writes the change positions, a last kernel turns
//    positions into run lengths.  Only the run lengths leave the device (KBs instead of 1 MB per mask).
#include "common.h"
#include "../../include/ape_hip.h"
#include <math.h>

#define RS_PRECISION_BITS 22  // Pillow Resample.c: 32 - 8 - 2
#define RS_TW 64              // output columns per workgroup

// ------------------------------------------------------------------------------------------------- coefficients (host)
// Pillow precompute_coeffs + normalize_coeffs_8bpc for the triangle filter over the whole axis, in the same double
// arithmetic.  bounds [out, 2] = (first source index, count), kk [out, ksize] (zero padded).  Returns ksize.
extern "C" int ape_hip_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int ksize_cap) {
  APE_CHECK_ARG(in_size > 0 && out_size > 0, "resize_coeffs: sizes must be positive");
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  if (!bounds || !kk) return ksize;
  APE_CHECK_ARG(ksize_cap >= ksize, "resize_coeffs: kk holds %d coefficients per output, %d needed", ksize_cap, ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int32_t* k = kk + (size_t)xx * ksize_cap;
    double w[4096];
    double* wp = w;
    double* heap = nullptr;
    if (xmax > 4096) wp = heap = new double[xmax];
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double t = (x + xmin - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      wp[x] = t < 1.0 ? 1.0 - t : 0.0;
      ww += wp[x];
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) wp[x] /= ww;
      const double v = wp[x] * (double)(1 << RS_PRECISION_BITS);
      k[x] = wp[x] < 0 ? (int32_t)(-0.5 + v) : (int32_t)(0.5 + v);
    }
    for (int x = xmax; x < ksize_cap; ++x) k[x] = 0;
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
    delete[] heap;
  }
  return ksize;
}

// ------------------------------------------------------------------------------------------------- resize kernel
__device__ __forceinline__ uint32_t clip8(int32_t acc) {
  const int32_t v = acc >> RS_PRECISION_BITS;
  return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src [H, src_ld bytes] HWC uint8 (3 channels).  bounds_h / kk_h over the W axis (NULL: width unchanged), bounds_v / kk_v
// over the H axis (NULL: height unchanged).  dst_kind 0: uint8 HWC, dst_ld = bytes per row; 1: float32 CHW, dst_ld =
// floats per row, dst_plane = floats per channel plane.  flip: output channel c = source channel 2 - c.
template <int DST_KIND>
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, int H, int W, int src_ld,
                                                        const int32_t* __restrict__ bounds_h, const int32_t* __restrict__ kk_h,
                                                        int ks_h, const int32_t* __restrict__ bounds_v,
                                                        const int32_t* __restrict__ kk_v, int ks_v, int newh, int neww, int TH,
                                                        void* __restrict__ dst_, int dst_ld, int dst_plane, int flip) {
  extern __shared__ uint8_t lds[];  // [rows][RS_TW * 4]: horizontally resampled pixels of the tile's source rows
  const int x0 = blockIdx.x * RS_TW, y0 = blockIdx.y * TH;
  const int y1 = min(y0 + TH, newh) - 1;
  int r0, r1;
  if (bounds_v) {
    r0 = bounds_v[2 * y0];
    r1 = bounds_v[2 * y1] + bounds_v[2 * y1 + 1];
  } else {
    r0 = y0;
    r1 = y1 + 1;
  }
  const int nrows = r1 - r0;
  const int tx = threadIdx.x & (RS_TW - 1);
  const int x = x0 + tx;
  uint32_t* lds32 = reinterpret_cast<uint32_t*>(lds);
  // ---- horizontal pass into LDS: thread = (column tx, rows ty, ty + 4, ...) ----
  if (x < neww) {
    int xmin = x, cnt = 1;
    const int32_t* k = nullptr;
    if (bounds_h) {
      xmin = bounds_h[2 * x];
      cnt = bounds_h[2 * x + 1];
      k = kk_h + (size_t)x * ks_h;
    }
    for (int r = threadIdx.x / RS_TW; r < nrows; r += 256 / RS_TW) {
      const uint8_t* p = src + (size_t)(r0 + r) * src_ld + (size_t)xmin * 3;
      uint32_t px;
      if (k) {
        int32_t a0 = 1 << (RS_PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int i = 0; i < cnt; ++i) {
          const int32_t c = k[i];
          a0 += (int32_t)p[3 * i] * c;
          a1 += (int32_t)p[3 * i + 1] * c;
          a2 += (int32_t)p[3 * i + 2] * c;
        }
        px = clip8(a0) | (clip8(a1) << 8) | (clip8(a2) << 16);
      } else {
        px = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
      }
      lds32[r * RS_TW + tx] = px;
    }
  }
  __syncthreads();
  // ---- vertical pass out of LDS ----
  if (x >= neww) return;
  for (int y = y0 + threadIdx.x / RS_TW; y <= y1; y += 256 / RS_TW) {
    uint32_t c0, c1, c2;
    if (bounds_v) {
      const int ymin = bounds_v[2 * y] - r0, cnt = bounds_v[2 * y + 1];
      const int32_t* k = kk_v + (size_t)y * ks_v;
      int32_t a0 = 1 << (RS_PRECISION_BITS - 1), a1 = a0, a2 = a0;
      for (int i = 0; i < cnt; ++i) {
        const uint32_t px = lds32[(ymin + i) * RS_TW + tx];
        const int32_t c = k[i];
        a0 += (int32_t)(px & 255u) * c;
        a1 += (int32_t)((px >> 8) & 255u) * c;
        a2 += (int32_t)((px >> 16) & 255u) * c;
      }
      c0 = clip8(a0); c1 = clip8(a1); c2 = clip8(a2);
    } else {
      const uint32_t px = lds32[(y - r0) * RS_TW + tx];
      c0 = px & 255u; c1 = (px >> 8) & 255u; c2 = (px >> 16) & 255u;
    }
    if (flip) { const uint32_t t = c0; c0 = c2; c2 = t; }
    if (DST_KIND == 0) {
      uint8_t* d = reinterpret_cast<uint8_t*>(dst_) + (size_t)y * dst_ld + (size_t)x * 3;
      d[0] = (uint8_t)c0; d[1] = (uint8_t)c1; d[2] = (uint8_t)c2;
    } else {
      float* d = reinterpret_cast<float*>(dst_) + (size_t)y * dst_ld + x;
      d[0] = (float)c0;
      d[(size_t)dst_plane] = (float)c1;
      d[2 * (size_t)dst_plane] = (float)c2;
    }
  }
}

// rows of source (after the horizontal pass) the tallest tile of TH output rows needs
static int tile_rows(const int32_t* bounds_v_host, int newh, int TH) {
  int m = 0;
  for (int y0 = 0; y0 < newh; y0 += TH) {
    const int y1 = (y0 + TH < newh ? y0 + TH : newh) - 1;
    const int n = bounds_v_host[2 * y1] + bounds_v_host[2 * y1 + 1] - bounds_v_host[2 * y0];
    if (n > m) m = n;
  }
  return m;
}

extern "C" int ape_hip_resize_tile_rows(const int32_t* bounds_v_host, int newh, int* tile_h) {
  // picks the output-tile height whose source rows fit 64 KB of LDS (256 B per row); returns the LDS rows needed
  APE_CHECK_ARG(bounds_v_host && tile_h && newh > 0, "resize_tile_rows: bad arguments");
  int TH = 32;
  int rows = tile_rows(bounds_v_host, newh, TH);
  while (rows > 256 && TH > 1) {
    TH >>= 1;
    rows = tile_rows(bounds_v_host, newh, TH);
  }
  APE_CHECK_ARG(rows <= 256, "resize: a single output row needs %d source rows (> 256): down-scaling factor too large", rows);
  *tile_h = TH;
  return rows;
}

extern "C" int ape_hip_resize_bilinear_u8(const uint8_t* src, int H, int W, int src_ld, const int32_t* bounds_h,
                                          const int32_t* kk_h, int ks_h, const int32_t* bounds_v, const int32_t* kk_v, int ks_v,
                                          int newh, int neww, int tile_h, int lds_rows, void* dst, int dst_kind, int dst_ld,
                                          int dst_plane, int flip, void* stream) {
  APE_CHECK_ARG(src && dst && H > 0 && W > 0 && newh > 0 && neww > 0, "resize: bad arguments");
  APE_CHECK_ARG(dst_kind == 0 || dst_kind == 1, "resize: dst_kind 0 (uint8 HWC) or 1 (float32 CHW)");
  APE_CHECK_ARG((bounds_h != nullptr) == (kk_h != nullptr) && (bounds_v != nullptr) == (kk_v != nullptr),
                "resize: bounds and coefficients come in pairs");
  APE_CHECK_ARG(bounds_h || neww == W, "resize: the width changes but no horizontal coefficients were given");
  APE_CHECK_ARG(bounds_v || newh == H, "resize: the height changes but no vertical coefficients were given");
  if (!bounds_v) { tile_h = 32; lds_rows = 32; }
  APE_CHECK_ARG(tile_h >= 1 && tile_h <= 32 && lds_rows >= 1 && lds_rows <= 256, "resize: tile_h / lds_rows out of range");
  const dim3 grid(ceil_div(neww, RS_TW), ceil_div(newh, tile_h));
  const size_t shm = (size_t)lds_rows * RS_TW * 4;
  hipStream_t st = (hipStream_t)stream;
  if (dst_kind == 0)
    APE_LAUNCH(resize_u8_kernel<0>, grid, dim3(256), shm, st, src, H, W, src_ld, bounds_h, kk_h, ks_h, bounds_v, kk_v,
                       ks_v, newh, neww, tile_h, dst, dst_ld, dst_plane, flip);
  else
    APE_LAUNCH(resize_u8_kernel<1>, grid, dim3(256), shm, st, src, H, W, src_ld, bounds_h, kk_h, ks_h, bounds_v, kk_v,
                       ks_v, newh, neww, tile_h, dst, dst_ld, dst_plane, flip);
  APE_CHECK_LAUNCH("resize_u8_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------------- COCO RLE
#define RLE_SEG 128  // rows per (column, segment) unit

// value that precedes element (x, y) in column-major order (0 before the first element)
__device__ __forceinline__ uint32_t rle_prev(const uint8_t* m, int H, int W, int x, int y) {
  if (y > 0) return m[(size_t)(y - 1) * W + x] != 0;
  if (x > 0) return m[(size_t)(H - 1) * W + (x - 1)] != 0;
  return 0u;
}

// unit u = x * nseg + s; MODE 0: cnt[mask, u] = value changes inside the unit; MODE 1: write their column-major positions
template <int MODE>
__global__ __launch_bounds__(256) void rle_walk_kernel(const uint8_t* __restrict__ masks, int H, int W, int nseg,
                                                       uint32_t* __restrict__ cnt, uint32_t* __restrict__ pos, int cap) {
  const int mi = blockIdx.z, s = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const uint8_t* m = masks + (size_t)mi * H * W;
  const int ya = s * RLE_SEG, yb = min(ya + RLE_SEG, H);
  uint32_t prev = rle_prev(m, H, W, x, ya);
  const size_t u = (size_t)mi * W * nseg + (size_t)x * nseg + s;
  uint32_t n = 0, o = 0;
  if (MODE == 1) o = cnt[u];  // exclusive offsets after the scan
  uint32_t* out = pos + (size_t)mi * cap;
#pragma unroll 8
  for (int y = ya; y < yb; ++y) {
    const uint32_t v = m[(size_t)y * W + x] != 0;
    if (v != prev) {
      if (MODE == 1) {
        if (o < (uint32_t)cap) out[o] = (uint32_t)x * (uint32_t)H + (uint32_t)y;
        ++o;
      }
      ++n;
    }
    prev = v;
  }
  if (MODE == 0) cnt[u] = n;
}

// one workgroup per mask: exclusive scan of its U unit counts in place, total -> nchg[mask]
__global__ __launch_bounds__(1024) void rle_scan_kernel(uint32_t* __restrict__ cnt, int U, uint32_t* __restrict__ nchg) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  uint32_t* c = cnt + (size_t)blockIdx.x * U;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < U; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < U ? c[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    const uint32_t before = carry;
    if (i < U) c[i] = before + wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) nchg[blockIdx.x] = carry;
}

// positions of value changes -> run lengths: counts[0] = pos[0], counts[j] = pos[j] - pos[j-1], counts[n] = H*W - pos[n-1]
__global__ void rle_runs_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ nchg, int cap, uint32_t total,
                                uint32_t* __restrict__ counts, uint32_t* __restrict__ nruns) {
  const int mi = blockIdx.y;
  const uint32_t n = nchg[mi];
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) nruns[mi] = n + 1;
  if (j > n || j >= (uint32_t)cap) return;
  const uint32_t* p = pos + (size_t)mi * cap;
  const uint32_t lo = j == 0 ? 0u : p[j - 1];
  const uint32_t hi = j == n ? total : p[j];
  counts[(size_t)mi * cap + j] = hi - lo;
}

extern "C" int ape_hip_rle_workspace_words(int n, int H, int W, int cap) {
  // cnt [n, W * nseg] + pos [n, cap] + nchg [n]
  const int nseg = ceil_div(H, RLE_SEG);
  const long long words = (long long)n * W * nseg + (long long)n * cap + n;
  return words > 0x7fffffffLL ? -1 : (int)words;
}

extern "C" int ape_hip_rle_encode(const uint8_t* masks, int n, int H, int W, uint32_t* workspace, uint32_t* counts, int cap,
                                  uint32_t* nruns, void* stream) {
  if (n == 0) return 0;
  APE_CHECK_ARG(masks && workspace && counts && nruns && n > 0 && H > 0 && W > 0 && cap >= 2, "rle_encode: bad arguments");
  APE_CHECK_ARG((long long)H * W < 0xffffffffLL, "rle_encode: mask too large");
  const int nseg = ceil_div(H, RLE_SEG);
  const int U = W * nseg;
  uint32_t* cnt = workspace;
  uint32_t* pos = cnt + (size_t)n * U;
  uint32_t* nchg = pos + (size_t)n * cap;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ceil_div(W, 256), nseg, n);
  APE_LAUNCH(rle_walk_kernel<0>, grid, dim3(256), 0, st, masks, H, W, nseg, cnt, pos, cap);
  APE_LAUNCH(rle_scan_kernel, dim3(n), dim3(1024), 0, st, cnt, U, nchg);
  APE_LAUNCH(rle_walk_kernel<1>, grid, dim3(256), 0, st, masks, H, W, nseg, cnt, pos, cap);
  APE_LAUNCH(rle_runs_kernel, dim3(ceil_div(cap, 256), n), dim3(256), 0, st, pos, nchg, cap, (uint32_t)((long long)H * W),
                     counts, nruns);
  APE_CHECK_LAUNCH("rle_encode");
  return 0;
}

// cocoapi maskApi.c rleToString on the host: counts -> ASCII (5-bit groups, low first, 0x20 = continuation, +48; from the
// fourth count on the difference to the count two back).  Returns the string length, or -(needed) if cap is too small.
extern "C" int ape_hip_rle_to_string(const uint32_t* counts, int n, char* out, int cap) {
  int p = 0;
  for (int i = 0; i < n; ++i) {
    long long x = (long long)counts[i];
    if (i > 2) x -= (long long)counts[i - 2];
    bool more = true;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      c += 48;
      if (p < cap) out[p] = c;
      ++p;
    }
  }
  return p <= cap ? p : -p;
}
