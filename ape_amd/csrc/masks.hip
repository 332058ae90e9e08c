// Instance-mask post-processing for the kept detections only (bilinear interpolation is per channel, so
// upsampling just the kept queries gives the same result as the reference's 900-channel upsample).
//   mask_upsample_bits : F.interpolate(mask_pred, size=(S,S), bilinear, align_corners=False) then sigmoid > 0.5
//                        (deformable_detr_segm_vl.py:569-572, 605)  -> uint8 bitmask [n,S,S]
//   roi_align_bits     : detectron2 BitMasks.crop_and_resize(boxes, 128) = torchvision roi_align(aligned=True,
//                        sampling_ratio=0) of the bitmask, >= 0.5   (deformable_detr_segm_vl.py:606-608)
//   paste_bits         : detectron2 paste_masks_in_image / _do_paste_mask (F.grid_sample, align_corners=False,
//                        zeros) at the output resolution, >= 0.5 (deformable_detr_segm_vl.py:869-871)
#include "common.h"
#include "../../include/ape_hip.h"

// one thread = 16 consecutive output pixels of one row (one 16-byte store)
template <typename T>
__global__ __launch_bounds__(256) void mask_upsample_bits_kernel(const T* __restrict__ logits, int ldl, int h0, int w0, int S,
                                                                 int n, uint8_t* __restrict__ out) {
  const int xg = (S + 15) / 16;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)n * S * xg) return;
  const int xb = (int)(gid % xg) * 16, y = (int)((gid / xg) % S), q = (int)(gid / ((size_t)xg * S));
  const float sy = (float)h0 / (float)S, sx = (float)w0 / (float)S;
  float fy = sy * ((float)y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  const int y0 = (int)fy;
  const int y1 = y0 + (y0 < h0 - 1 ? 1 : 0);
  const float ly = fy - (float)y0;
  const T* p0 = logits + (size_t)q * ldl + (size_t)y0 * w0;
  const T* p1 = logits + (size_t)q * ldl + (size_t)y1 * w0;
  uint8_t b[16];
  if (S == 4 * w0 && xb + 16 <= S) {
    // x4 upsampling (the mask features live at stride 4): 16 outputs need source columns kb-1 .. kb+4; load the two
    // source rows once and blend in registers.  src = 0.25*(x+0.5)-0.5 -> (x0, lx) = (k-1, .625), (k-1, .875), (k, .125), (k, .375)
    const int kb = xb >> 2;
    float r0[6], r1[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      int xs = kb - 1 + j; xs = xs < 0 ? 0 : (xs > w0 - 1 ? w0 - 1 : xs);
      r0[j] = ldf<T>(p0 + xs);
      r1[j] = ldf<T>(p1 + xs);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int x = xb + i;
      float fx = sx * ((float)x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
      const float lx = fx - (float)(int)fx;
      // source column of the left tap relative to kb-1 is a compile-time function of i; the clamped preloads make the
      // image borders come out right (left border: lx == 0; right border: both taps read column w0-1)
      const int j0 = (i >> 2) + ((i & 3) >= 2 ? 1 : 0);
      const float v00 = r0[j0], v01 = r0[j0 + 1], v10 = r1[j0], v11 = r1[j0 + 1];
      const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
      b[i] = v > 0.f ? 1 : 0;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int x = xb + i;
      float fx = sx * ((float)x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
      int x0 = (int)fx; x0 = x0 < w0 - 1 ? x0 : w0 - 1;
      const int x1 = x0 + (x0 < w0 - 1 ? 1 : 0);
      const float lx = fx - (float)x0;
      const float v = (1.f - ly) * ((1.f - lx) * ldf<T>(p0 + x0) + lx * ldf<T>(p0 + x1)) +
                      ly * ((1.f - lx) * ldf<T>(p1 + x0) + lx * ldf<T>(p1 + x1));
      b[i] = v > 0.f ? 1 : 0;
    }
  }
  uint8_t* dst = out + ((size_t)q * S + y) * S + xb;
  if (xb + 16 <= S && (S % 16) == 0) {
    uint4 pk;
    pk.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24);
    pk.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
    pk.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((uint32_t)b[11] << 24);
    pk.w = b[12] | (b[13] << 8) | (b[14] << 16) | ((uint32_t)b[15] << 24);
    *reinterpret_cast<uint4*>(dst) = pk;
  } else {
    for (int i = 0; i < 16 && xb + i < S; ++i) dst[i] = b[i];
  }
}

extern "C" int ape_hip_mask_upsample_bits(const void* logits, int ldl, int dt, int h0, int w0, int S, int n, uint8_t* out,
                                          void* stream) {
  APE_CHECK_ARG(logits && out && h0 > 0 && w0 > 0 && S > 0 && n > 0, "ape_hip_mask_upsample_bits: bad args");
  const size_t total = (size_t)n * S * ((S + 15) / 16);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dt == APE_DT_F16) APE_LAUNCH(mask_upsample_bits_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, (const f16_t*)logits, ldl, h0, w0, S, n, out);
  else if (dt == APE_DT_BF16) APE_LAUNCH(mask_upsample_bits_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)logits, ldl, h0, w0, S, n, out);
  else APE_LAUNCH(mask_upsample_bits_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)logits, ldl, h0, w0, S, n, out);
  APE_CHECK_LAUNCH("ape_hip_mask_upsample_bits");
  return 0;
}

__device__ __forceinline__ float roi_bilinear(const uint8_t* __restrict__ img, int H, int W, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
  const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
  return hy * hx * (float)img[y_low * W + x_low] + hy * lx * (float)img[y_low * W + x_high] +
         ly * hx * (float)img[y_high * W + x_low] + ly * lx * (float)img[y_high * W + x_high];
}

__global__ __launch_bounds__(256) void roi_align_bits_kernel(const uint8_t* __restrict__ bits, int H, int W, const float* __restrict__ boxes,
                                                             int n, int P, uint8_t* __restrict__ out) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= n * P * P) return;
  const int pw = gid % P, ph = (gid / P) % P, q = gid / (P * P);
  const float x1 = boxes[q * 4 + 0], y1 = boxes[q * 4 + 1], x2 = boxes[q * 4 + 2], y2 = boxes[q * 4 + 3];
  const float sw = x1 - 0.5f, sh = y1 - 0.5f;
  const float rw = (x2 - 0.5f) - sw, rh = (y2 - 0.5f) - sh;
  const float bw = rw / (float)P, bh = rh / (float)P;
  const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
  const float count = (float)max(gh * gw, 1);
  const uint8_t* img = bits + (size_t)q * H * W;
  float sum = 0.f;
  for (int iy = 0; iy < gh; ++iy) {
    const float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
    for (int ix = 0; ix < gw; ++ix) {
      const float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
      sum += roi_bilinear(img, H, W, y, x);
    }
  }
  out[gid] = (sum / count) >= 0.5f ? 1 : 0;
}

extern "C" int ape_hip_roi_align_bits(const uint8_t* bits, int H, int W, const float* boxes, int n, int P, uint8_t* out, void* stream) {
  APE_CHECK_ARG(bits && boxes && out && n > 0 && P > 0, "ape_hip_roi_align_bits: bad args");
  APE_LAUNCH(roi_align_bits_kernel, dim3(ceil_div(n * P * P, 256)), dim3(256), 0, (hipStream_t)stream, bits, H, W, boxes, n, P, out);
  APE_CHECK_LAUNCH("ape_hip_roi_align_bits");
  return 0;
}

// one thread = 16 consecutive output pixels of one row
__global__ __launch_bounds__(256) void paste_bits_kernel(const uint8_t* __restrict__ m, int P, const float* __restrict__ boxes, int n,
                                                         int Ho, int Wo, uint8_t* __restrict__ out) {
  const int xg = (Wo + 15) / 16;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)n * Ho * xg) return;
  const int xb = (int)(gid % xg) * 16, y = (int)((gid / xg) % Ho), q = (int)(gid / ((size_t)xg * Ho));
  const float x0 = boxes[q * 4 + 0], y0 = boxes[q * 4 + 1], x1 = boxes[q * 4 + 2], y1 = boxes[q * 4 + 3];
  const float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2.f - 1.f;
  // grid_sample, align_corners=False: ix = ((g + 1) * size - 1) / 2 ; zeros padding
  const float fy = ((gy + 1.f) * (float)P - 1.f) * 0.5f;
  const float fly = floorf(fy);
  const int iy0 = (int)fly, iy1 = iy0 + 1;
  const float ly = fy - fly;
  const uint8_t* img = m + (size_t)q * P * P;
  auto at = [&](int yy, int xx) -> float { return (yy >= 0 && yy < P && xx >= 0 && xx < P) ? (float)img[yy * P + xx] : 0.f; };
  uint8_t* dst = out + ((size_t)q * Ho + y) * Wo + xb;
  uint8_t b[16];
  {
    // most pixels lie outside the detection's box: if no tap of this 16-pixel run can touch the P x P mask, store zeros
    // (fx is monotonic in x for x1 > x0; degenerate boxes take the general path)
    const float fxa = ((((float)xb + 0.5f - x0) / (x1 - x0) * 2.f - 1.f) + 1.f) * (float)P * 0.5f - 0.5f;
    const float fxb = ((((float)(xb + 15) + 0.5f - x0) / (x1 - x0) * 2.f - 1.f) + 1.f) * (float)P * 0.5f - 0.5f;
    const bool yout = iy1 < 0 || iy0 >= P;
    const bool xout = (x1 > x0) && (fxb < -1.f || fxa >= (float)P);
    if ((yout || xout) && xb + 16 <= Wo && (Wo % 16) == 0) {
      *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int x = xb + i;
    const float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2.f - 1.f;
    const float fx = ((gx + 1.f) * (float)P - 1.f) * 0.5f;
    const float flx = floorf(fx);
    const int ix0 = (int)flx, ix1 = ix0 + 1;
    const float lx = fx - flx;
    const float v = at(iy0, ix0) * (1.f - lx) * (1.f - ly) + at(iy0, ix1) * lx * (1.f - ly) + at(iy1, ix0) * (1.f - lx) * ly +
                    at(iy1, ix1) * lx * ly;
    b[i] = v >= 0.5f ? 1 : 0;
  }
  if (xb + 16 <= Wo && (Wo % 16) == 0) {
    uint4 pk;
    pk.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24);
    pk.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
    pk.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((uint32_t)b[11] << 24);
    pk.w = b[12] | (b[13] << 8) | (b[14] << 16) | ((uint32_t)b[15] << 24);
    *reinterpret_cast<uint4*>(dst) = pk;
  } else {
    for (int i = 0; i < 16 && xb + i < Wo; ++i) dst[i] = b[i];
  }
}

extern "C" int ape_hip_paste_bits(const uint8_t* masks, int P, const float* boxes, int n, int Ho, int Wo, uint8_t* out, void* stream) {
  APE_CHECK_ARG(masks && boxes && out && n > 0 && P > 0 && Ho > 0 && Wo > 0, "ape_hip_paste_bits: bad args");
  const size_t total = (size_t)n * Ho * ((Wo + 15) / 16);
  APE_LAUNCH(paste_bits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, masks, P, boxes, n, Ho, Wo, out);
  APE_CHECK_LAUNCH("ape_hip_paste_bits");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Semantic branch (deformable_detr_segm_vl.py:628-666, 875-918): sem[c, y, x] = sum_q P[q, c] * sigmoid(up(mask_q))(y, x).
//   mask_upsample_sigmoid : pixel-major probabilities  out[(y, x), q] = sigmoid(bilinear_up(logits)[q, y, x])  for the
//                           un-padded image region only (sem_seg_postprocess crops the padding away, so those pixels are
//                           never produced).  Input logits are pixel-major [h0*w0, n] (the mask GEMM writes them that
//                           way), so the 4 taps of a pixel are contiguous n-vectors.  The einsum is then ONE GEMM
//                           [K, n] x [n, h*w] through ape_hip_gemm.
//   bilinear_resize       : F.interpolate(bilinear, align_corners=False) of the [C, h, w] result to the output size
// ---------------------------------------------------------------------------------------------------------------
// one bilinear tap set + sigmoid, shared by the scalar and the vector kernel below: contraction off, so both evaluate the SAME sequence of
// fp32 operations (hipcc otherwise fuses the products into FMAs differently in the two loop shapes: last-bit differences)
__device__ __forceinline__ float bilerp_sigmoid(float a, float b, float c, float d, float lx, float ly) {
#pragma clang fp contract(off)
  const float top = (1.f - lx) * a + lx * b;
  const float bot = (1.f - lx) * c + lx * d;
  const float v = (1.f - ly) * top + ly * bot;
  return 1.f / (1.f + expf(-v));
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void mask_upsample_sigmoid_kernel(const TI* __restrict__ logits, int ldl, int h0, int w0, int S,
                                                                    int ch, int cw, int n, TO* __restrict__ out, int ldo) {
  const int ng = (n + 3) / 4;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)ch * cw * ng) return;
  const int g = (int)(gid % ng);
  const size_t pix = gid / ng;
  const int x = (int)(pix % cw), y = (int)(pix / cw);
  const float sy = (float)h0 / (float)S, sx = (float)w0 / (float)S;
  float fy = sy * ((float)y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  float fx = sx * ((float)x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
  int y0 = (int)fy; y0 = y0 < h0 - 1 ? y0 : h0 - 1;
  int x0 = (int)fx; x0 = x0 < w0 - 1 ? x0 : w0 - 1;
  const int y1 = y0 + (y0 < h0 - 1 ? 1 : 0), x1 = x0 + (x0 < w0 - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const TI* p00 = logits + ((size_t)y0 * w0 + x0) * ldl;
  const TI* p01 = logits + ((size_t)y0 * w0 + x1) * ldl;
  const TI* p10 = logits + ((size_t)y1 * w0 + x0) * ldl;
  const TI* p11 = logits + ((size_t)y1 * w0 + x1) * ldl;
  TO* o = out + pix * ldo;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = g * 4 + i;
    if (q < n) {
      stf<TO>(o + q, bilerp_sigmoid(ldf<TI>(p00 + q), ldf<TI>(p01 + q), ldf<TI>(p10 + q), ldf<TI>(p11 + q), lx, ly));
    }
  }
}

// Vector form (round 6): one thread = 8 consecutive queries of one output pixel -- the 4 taps are 2 x 16-byte loads each (fp32 logits),
// the 8 probabilities leave as ONE 16-byte store; threads 0..63 of a wave walk the query groups of a pixel, the 4 waves of a workgroup 4
// neighbouring pixels of a row (they share their taps: 4 x upsampling), grid = (pixel quads of a row, rows): no integer division.
// The scalar kernel above wrote 2-byte pieces behind 64-bit div / mod per thread: 4.27 ms for the 2.36 GB of a 1536^2 image with 500 kept
// queries (0.55 TB/s); same arithmetic per element, so the outputs are bit-identical.
template <typename TO>
__global__ __launch_bounds__(256) void mask_upsample_sigmoid8_kernel(const float* __restrict__ logits, int ldl, int h0, int w0, int S,
                                                                     int cw, int n, TO* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const int x = blockIdx.x * 4 + (threadIdx.x >> 6), y = blockIdx.y;
  if (x >= cw) return;
  const float sy = (float)h0 / (float)S, sx = (float)w0 / (float)S;
  float fy = sy * ((float)y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  float fx = sx * ((float)x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
  int y0 = (int)fy; y0 = y0 < h0 - 1 ? y0 : h0 - 1;
  int x0 = (int)fx; x0 = x0 < w0 - 1 ? x0 : w0 - 1;
  const int y1 = y0 + (y0 < h0 - 1 ? 1 : 0), x1 = x0 + (x0 < w0 - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float* p00 = logits + ((size_t)y0 * w0 + x0) * ldl;
  const float* p01 = logits + ((size_t)y0 * w0 + x1) * ldl;
  const float* p10 = logits + ((size_t)y1 * w0 + x0) * ldl;
  const float* p11 = logits + ((size_t)y1 * w0 + x1) * ldl;
  TO* o = out + ((size_t)y * cw + x) * ldo;
  for (int q = lane * 8; q < n; q += 512) {
    float a[8], b[8], c[8], d[8], r[8];
    *reinterpret_cast<float4*>(a) = *reinterpret_cast<const float4*>(p00 + q); *reinterpret_cast<float4*>(a + 4) = *reinterpret_cast<const float4*>(p00 + q + 4);
    *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(p01 + q); *reinterpret_cast<float4*>(b + 4) = *reinterpret_cast<const float4*>(p01 + q + 4);
    *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(p10 + q); *reinterpret_cast<float4*>(c + 4) = *reinterpret_cast<const float4*>(p10 + q + 4);
    *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(p11 + q); *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(p11 + q + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = bilerp_sigmoid(a[i], b[i], c[i], d[i], lx, ly);
    st8<TO>(o + q, r);
  }
}

extern "C" int ape_hip_mask_upsample_sigmoid(const void* logits, int ldl, int in_dt, int h0, int w0, int S, int crop_h, int crop_w,
                                             int n, void* out, int ldo, int out_dt, void* stream) {
  APE_CHECK_ARG(logits && out && h0 > 0 && w0 > 0 && S > 0 && n > 0 && crop_h > 0 && crop_w > 0 && crop_h <= S && crop_w <= S,
                "ape_hip_mask_upsample_sigmoid: bad args");
  const size_t total = (size_t)crop_h * crop_w * ((n + 3) / 4);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int hk = APE_H16_KIND(in_dt, out_dt);
  if (hk < 0) { ape_set_error("ape_hip_mask_upsample_sigmoid: dtypes must be f32 or ONE 16-bit type (in %d, out %d)", in_dt, out_dt); return -1; }
  const int key = (ape_is16(in_dt) ? 2 : 0) + (ape_is16(out_dt) ? 1 : 0);
  // the vector kernel: fp32 logits, 8 queries per thread (APE_MASK_UP8=0: the scalar kernel, A/B and tests)
  {
    const char* e8 = getenv("APE_MASK_UP8");
    if (!(e8 != nullptr && atoi(e8) == 0) && in_dt == APE_DT_F32 && n % 8 == 0 && ldl % 4 == 0 && ldo % 8 == 0 && ((uintptr_t)logits) % 16 == 0 &&
        ((uintptr_t)out) % 16 == 0 && crop_h <= 65535) {
      const dim3 g8((unsigned)((crop_w + 3) / 4), (unsigned)crop_h);
      if (out_dt == APE_DT_F32) APE_LAUNCH((mask_upsample_sigmoid8_kernel<float>), g8, block, 0, s, (const float*)logits, ldl, h0, w0, S, crop_w, n, (float*)out, ldo);
      else if (out_dt == APE_DT_F16) APE_LAUNCH((mask_upsample_sigmoid8_kernel<f16_t>), g8, block, 0, s, (const float*)logits, ldl, h0, w0, S, crop_w, n, (f16_t*)out, ldo);
      else APE_LAUNCH((mask_upsample_sigmoid8_kernel<bf16_t>), g8, block, 0, s, (const float*)logits, ldl, h0, w0, S, crop_w, n, (bf16_t*)out, ldo);
      APE_CHECK_LAUNCH("ape_hip_mask_upsample_sigmoid");
      return 0;
    }
  }
  if (key == 0) { APE_LAUNCH((mask_upsample_sigmoid_kernel<float, float>), grid, block, 0, s, (const float*)logits, ldl, h0, w0, S, crop_h, crop_w, n, (float*)out, ldo); }
  else if (hk == APE_DT_F16) {
    if (key == 1) { APE_LAUNCH((mask_upsample_sigmoid_kernel<float, f16_t>), grid, block, 0, s, (const float*)logits, ldl, h0, w0, S, crop_h, crop_w, n, (f16_t*)out, ldo); } else if (key == 2) { APE_LAUNCH((mask_upsample_sigmoid_kernel<f16_t, float>), grid, block, 0, s, (const f16_t*)logits, ldl, h0, w0, S, crop_h, crop_w, n, (float*)out, ldo); } else { APE_LAUNCH((mask_upsample_sigmoid_kernel<f16_t, f16_t>), grid, block, 0, s, (const f16_t*)logits, ldl, h0, w0, S, crop_h, crop_w, n, (f16_t*)out, ldo); }
  } else {
    if (key == 1) { APE_LAUNCH((mask_upsample_sigmoid_kernel<float, bf16_t>), grid, block, 0, s, (const float*)logits, ldl, h0, w0, S, crop_h, crop_w, n, (bf16_t*)out, ldo); } else if (key == 2) { APE_LAUNCH((mask_upsample_sigmoid_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)logits, ldl, h0, w0, S, crop_h, crop_w, n, (float*)out, ldo); } else { APE_LAUNCH((mask_upsample_sigmoid_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)logits, ldl, h0, w0, S, crop_h, crop_w, n, (bf16_t*)out, ldo); }
  }
  APE_CHECK_LAUNCH("ape_hip_mask_upsample_sigmoid");
  return 0;
}

__global__ __launch_bounds__(256) void bilinear_resize_kernel(const float* __restrict__ in, int ldc, int ldr, int h, int w, int C,
                                                              float* __restrict__ out, int H, int W) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)C * H * W) return;
  const int x = (int)(gid % W), y = (int)((gid / W) % H), c = (int)(gid / ((size_t)W * H));
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  float fy = sy * ((float)y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  float fx = sx * ((float)x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
  int y0 = (int)fy; y0 = y0 < h - 1 ? y0 : h - 1;
  int x0 = (int)fx; x0 = x0 < w - 1 ? x0 : w - 1;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float* p = in + (size_t)c * ldc;
  out[gid] = (1.f - ly) * ((1.f - lx) * p[(size_t)y0 * ldr + x0] + lx * p[(size_t)y0 * ldr + x1]) +
             ly * ((1.f - lx) * p[(size_t)y1 * ldr + x0] + lx * p[(size_t)y1 * ldr + x1]);
}

extern "C" int ape_hip_bilinear_resize(const float* in, int ld_channel, int ld_row, int h, int w, int C, float* out, int H, int W,
                                       void* stream) {
  APE_CHECK_ARG(in && out && h > 0 && w > 0 && C > 0 && H > 0 && W > 0, "ape_hip_bilinear_resize: bad args");
  const size_t total = (size_t)C * H * W;
  APE_LAUNCH(bilinear_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, ld_channel,
                     ld_row, h, w, C, out, H, W);
  APE_CHECK_LAUNCH("ape_hip_bilinear_resize");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Panoptic merge on the device (_postprocess_panoptic, ape/modeling/ape_deta/deformable_detr_segm_vl.py:921-998): the reference walks
// over the kept queries with three `.item()` read-backs each; here the merge is three launches and no host round trip, so it can
// sit inside a captured step (ape_amd/runtime.py, GraphedForward(panoptic=meta)).
//   panoptic_pixels : per output pixel, over the kept queries q in index order: p_q = sigmoid(bilinear(mask logits of q)) (the
//                     second resize of sem_seg_postprocess, :942, fused: the [k, H, W] tensor never exists), owner = first argmax of
//                     score_q p_q (:957-959), conf = p_owner >= prob; areas[q] = (#pixels owned, #pixels with p_q >= prob, #both)
//                     (:965-967) by wave-aggregated atomics
//   panoptic_decide : the sequential walk (:963-995) on one thread: overlap test in double like Python's int / int, stuff classes
//                     merged through a per-class memory, category remap of the "things first" stuff vocabulary -> segment id per
//                     query + the segments_info table
//   panoptic_write  : panoptic_seg[pixel] = id[owner] where conf, else 0
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void panoptic_pixels_kernel(const float* __restrict__ masks, int ldq, int ldr, int h, int w, int k,
                                                              const float* __restrict__ score, const uint8_t* __restrict__ keep, float prob,
                                                              int H, int W, int16_t* __restrict__ owner, uint8_t* __restrict__ conf,
                                                              int* __restrict__ areas) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = gid < (size_t)H * W;
  const size_t pix = in ? gid : 0;
  const int x = (int)(pix % W), y = (int)(pix / W);
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  float fy = sy * ((float)y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  float fx = sx * ((float)x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
  int y0 = (int)fy; y0 = y0 < h - 1 ? y0 : h - 1;
  int x0 = (int)fx; x0 = x0 < w - 1 ? x0 : w - 1;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const size_t o00 = (size_t)y0 * ldr + x0, o01 = (size_t)y0 * ldr + x1, o10 = (size_t)y1 * ldr + x0, o11 = (size_t)y1 * ldr + x1;
  float bestv = -INFINITY;
  int best = -1;
  bool bconf = false;
  for (int q = 0; q < k; ++q) {
    if (keep[q] == 0) continue;                                     // uniform
    const float* p = masks + (size_t)q * ldq;
    const float v = (1.f - ly) * ((1.f - lx) * p[o00] + lx * p[o01]) + ly * ((1.f - lx) * p[o10] + lx * p[o11]);
    const float pr = 1.f / (1.f + expf(-v));
    const bool c = in && pr >= prob;
    const unsigned long long b = __ballot(c);
    if (b != 0ull && lane == (int)__builtin_ctzll(b)) atomicAdd(&areas[q * 3 + 1], (int)__builtin_popcountll(b));
    const float val = score[q] * pr;
    if (val > bestv) { bestv = val; best = q; bconf = c; }
  }
  if (in) { owner[gid] = (int16_t)best; conf[gid] = bconf ? 1 : 0; }
  const int mine = in ? best : -1;
  unsigned long long todo = __ballot(mine >= 0);
  while (todo != 0ull) {                                            // one atomic pair per distinct owner in the wave
    const int leader = (int)__builtin_ctzll(todo);
    const int qv = __shfl(mine, leader, 64);
    const unsigned long long same = __ballot(mine == qv);
    const unsigned long long both = __ballot(mine == qv && bconf);
    if (lane == leader) {
      atomicAdd(&areas[qv * 3], (int)__builtin_popcountll(same));
      if (both != 0ull) atomicAdd(&areas[qv * 3 + 2], (int)__builtin_popcountll(both));
    }
    todo &= ~same;
  }
}

__global__ __launch_bounds__(256) void panoptic_decide_kernel(const int* __restrict__ areas, const int* __restrict__ classes, const uint8_t* __restrict__ keep,
                                                              int k, const uint8_t* __restrict__ isthing, int num_classes, double overlap_thr,
                                                              int stuff_offset, int* __restrict__ seg_id, int* __restrict__ info, int* __restrict__ count) {
  extern __shared__ int stuff_mem[];                                // segment id a stuff class already owns (0 = none)
  for (int i = threadIdx.x; i < num_classes; i += 256) stuff_mem[i] = 0;
  for (int i = threadIdx.x; i < k * 3; i += 256) info[i] = 0;
  __syncthreads();
  if (threadIdx.x != 0) return;
  int cur = 0, n = 0;
  for (int q = 0; q < k; ++q) {
    seg_id[q] = 0;
    if (keep[q] == 0) continue;
    const int ma = areas[q * 3], oa = areas[q * 3 + 1], ba = areas[q * 3 + 2];
    if (!(ma > 0 && oa > 0 && ba > 0)) continue;
    if ((double)ma / (double)oa < overlap_thr) continue;
    const int c = classes[q];
    const bool thing = c >= 0 && c < num_classes && isthing[c] != 0;
    if (!thing && c >= 0 && c < num_classes) {
      if (stuff_mem[c] != 0) { seg_id[q] = stuff_mem[c]; continue; }
      stuff_mem[c] = cur + 1;
    }
    ++cur;
    seg_id[q] = cur;
    info[n * 3] = cur;
    info[n * 3 + 1] = thing ? 1 : 0;
    info[n * 3 + 2] = (!thing && stuff_offset >= 0) ? c - stuff_offset + 1 : c;
    ++n;
  }
  *count = n;
}

__global__ __launch_bounds__(256) void panoptic_write_kernel(const int16_t* __restrict__ owner, const uint8_t* __restrict__ conf,
                                                             const int* __restrict__ seg_id, size_t total, int* __restrict__ out) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int q = owner[gid];
  out[gid] = (q >= 0 && conf[gid] != 0) ? seg_id[q] : 0;
}

extern "C" int ape_hip_panoptic_pixels(const float* masks, int ld_query, int ld_row, int h, int w, int k, const float* scores, const uint8_t* keep,
                                       float prob, int H, int W, int16_t* owner, uint8_t* conf, int* areas, void* stream) {
  APE_CHECK_ARG(masks && scores && keep && owner && conf && areas && h > 0 && w > 0 && k > 0 && k < 32768 && H > 0 && W > 0,
                "ape_hip_panoptic_pixels: bad args (1 <= k < 32768 queries)");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(areas, 0, (size_t)k * 3 * sizeof(int), s) != hipSuccess) { ape_set_error("ape_hip_panoptic_pixels: memset failed"); return -1; }
  const size_t total = (size_t)H * W;
  APE_LAUNCH(panoptic_pixels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, masks, ld_query, ld_row, h, w, k, scores, keep, prob,
                     H, W, owner, conf, areas);
  APE_CHECK_LAUNCH("ape_hip_panoptic_pixels");
  return 0;
}

extern "C" int ape_hip_panoptic_decide(const int* areas, const int* classes, const uint8_t* keep, int k, const uint8_t* isthing, int num_classes,
                                       double overlap_threshold, int stuff_offset, int* seg_id, int* info, int* count, void* stream) {
  APE_CHECK_ARG(areas && classes && keep && isthing && seg_id && info && count && k > 0 && num_classes > 0 && num_classes <= 16384,
                "ape_hip_panoptic_decide: bad args (num_classes <= 16384)");
  APE_LAUNCH(panoptic_decide_kernel, dim3(1), dim3(256), (size_t)num_classes * sizeof(int), (hipStream_t)stream, areas, classes, keep, k, isthing,
                     num_classes, overlap_threshold, stuff_offset, seg_id, info, count);
  APE_CHECK_LAUNCH("ape_hip_panoptic_decide");
  return 0;
}

extern "C" int ape_hip_panoptic_write(const int16_t* owner, const uint8_t* conf, const int* seg_id, int H, int W, int* out, void* stream) {
  APE_CHECK_ARG(owner && conf && seg_id && out && H > 0 && W > 0, "ape_hip_panoptic_write: bad args");
  const size_t total = (size_t)H * W;
  APE_LAUNCH(panoptic_write_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, owner, conf, seg_id, total, out);
  APE_CHECK_LAUNCH("ape_hip_panoptic_write");
  return 0;
}
