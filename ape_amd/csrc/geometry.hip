// Per-image-size constants of the deformable encoder, generated on the device straight into the buffers a captured graph
// reads (ape_amd/modeling/ape_deta/geometry.py StaticGeometry), so that a stream of differently sized images costs one small
// launch per image instead of ~40 tensor-library launches (or a per-size cache of ~140 MB per entry).
//
// Restates, per token t = (level l, y, x) of the L-level pyramid inside the S x S pad, for an image of (h, w) pixels:
//   * padding masks: nearest-neighbour resize of the image mask, F.interpolate default (deformable_detr_segm_vl.py:382-388);
//   * PositionEmbeddingSine(num_pos_feats, temperature, normalize=True, offset, eps, scale) of detrex
//     (ape_deta_r50.py:35-40) + level embedding (deformable_transformer_vl.py:461);
//   * valid ratios (deformable_transformer_vl.py:402-410), encoder reference points (:371-400);
//   * anchors of gen_encoder_output_proposals (:321-369) in logit space, +inf where unusable, and the unusable mask.
// The arithmetic follows geometry.py operation by operation (fp32, same association), so the values agree with the
// tensor-library evaluation to the last bit for every usable token (tests/test_ops_gpu.py::test_geometry_kernel).
#include "common.h"
#include "../../include/ape_hip.h"

struct GeoParams {
  int L, S, h, w, T;
  int H[8], W[8], start[8];
  const float* dim_t;        // [npf] temperature ** (2 * (i / 2) / npf), host-computed table
  const float* level_embeds; // [L, 2 * npf]
  int npf;
  float offset, eps, scale;
  void* lvl_pos; int lp_dt;  // [T, 2 * npf] compute dtype
  uint8_t* mask_u8;          // [T]
  uint8_t* mask_b;           // [T] torch.bool storage (same bytes)
  uint8_t* invalid_u8;       // [T]
  float* enc_ref;            // [T, L, 2]
  float* proposals;          // [T, 4]
  float* valid_ratios;       // [L, 2]
  float* vr4;                // [L, 4]
  float* box_scale;          // [4]
};

__device__ __forceinline__ int valid_count(int n, int S, int lim) {
  // number of i in [0, n) with floor(i * (S / n)) < lim   (S / n is an exact power of two for the pyramid strides)
  const float step = (float)S / (float)n;
  int c = 0;
  // closed form: i * step < lim  <=>  i < lim / step; kept as a loop-free expression with a guard for rounding
  c = (int)ceilf((float)lim / step);
  c = c < 0 ? 0 : (c > n ? n : c);
  while (c > 0 && floorf((float)(c - 1) * step) >= (float)lim) --c;
  while (c < n && floorf((float)c * step) < (float)lim) ++c;
  return c;
}

__global__ __launch_bounds__(256) void geometry_kernel(const GeoParams p) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x < p.L) {
    const int l = threadIdx.x;
    const float vw = (float)valid_count(p.W[l], p.S, p.w) / (float)p.W[l];
    const float vh = (float)valid_count(p.H[l], p.S, p.h) / (float)p.H[l];
    p.valid_ratios[l * 2] = vw; p.valid_ratios[l * 2 + 1] = vh;
    p.vr4[l * 4] = vw; p.vr4[l * 4 + 1] = vh; p.vr4[l * 4 + 2] = vw; p.vr4[l * 4 + 3] = vh;
    if (l == 0) { p.box_scale[0] = (float)p.w; p.box_scale[1] = (float)p.h; p.box_scale[2] = (float)p.w; p.box_scale[3] = (float)p.h; }
  }
  if (t >= p.T) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) if (i < p.L && t >= p.start[i]) l = i;
  const int H = p.H[l], W = p.W[l];
  const int r = t - p.start[l];
  const int y = r / W, x = r % W;
  const int vH = valid_count(H, p.S, p.h), vW = valid_count(W, p.S, p.w);
  const bool masked = y >= vH || x >= vW;
  // ---- sine position embedding (normalised), + level embedding
  //   y_embed = cumsum over rows of ~mask: min(y + 1, vH) in an unmasked column, 0 in a masked one; last row value = vH or 0
  const float ye = x < vW ? (float)min(y + 1, vH) : 0.f, yl = x < vW ? (float)vH : 0.f;
  const float xe = y < vH ? (float)min(x + 1, vW) : 0.f, xl = y < vH ? (float)vW : 0.f;
  const float yn = (ye + p.offset) / (yl + p.eps) * p.scale;
  const float xn = (xe + p.offset) / (xl + p.eps) * p.scale;
  const int C = 2 * p.npf;
  for (int c = lane; c < C; c += 64) {
    const int i = c < p.npf ? c : c - p.npf;
    const float a = (c < p.npf ? yn : xn) / p.dim_t[i];
    const float v = ((i & 1) ? cosf(a) : sinf(a)) + p.level_embeds[l * C + c];
    if (p.lp_dt == APE_DT_F32) reinterpret_cast<float*>(p.lvl_pos)[(size_t)t * C + c] = v;
    else if (p.lp_dt == APE_DT_F16) stf<f16_t>(reinterpret_cast<f16_t*>(p.lvl_pos) + (size_t)t * C + c, v);
    else reinterpret_cast<bf16_t*>(p.lvl_pos)[(size_t)t * C + c] = f2bf(v);
  }
  // ---- encoder reference points: ((x + 0.5) / (vr_w * W), (y + 0.5) / (vr_h * H)) * valid_ratios[l']
  if (lane < p.L) {
    const float vrw = (float)vW / (float)W, vrh = (float)vH / (float)H;
    const float rx = ((float)x + 0.5f) / (vrw * (float)W), ry = ((float)y + 0.5f) / (vrh * (float)H);
    const int l2 = lane;
    const float vw2 = (float)valid_count(p.W[l2], p.S, p.w) / (float)p.W[l2];
    const float vh2 = (float)valid_count(p.H[l2], p.S, p.h) / (float)p.H[l2];
    p.enc_ref[((size_t)t * p.L + l2) * 2] = rx * vw2;
    p.enc_ref[((size_t)t * p.L + l2) * 2 + 1] = ry * vh2;
  }
  // ---- anchors (logit space) and masks
  if (lane == 63) {
    const float gx = ((float)x + 0.5f) / (float)vW, gy = ((float)y + 0.5f) / (float)vH;
    const float wh = 0.05f * (float)(1 << l);
    const float pr[4] = {gx, gy, wh, wh};
    bool valid = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) valid = valid && pr[k] > 0.01f && pr[k] < 0.99f;
    const bool bad = masked || !valid;
#pragma unroll
    for (int k = 0; k < 4; ++k) p.proposals[(size_t)t * 4 + k] = bad ? INFINITY : logf(pr[k] / (1.f - pr[k]));
    p.mask_u8[t] = masked ? 1 : 0;
    p.mask_b[t] = masked ? 1 : 0;
    p.invalid_u8[t] = bad ? 1 : 0;
  }
}

extern "C" int ape_hip_geometry(int S, int h, int w, int L, const int* level_hw /* [L, 2] host */, const float* dim_t, int npf,
                                const float* level_embeds, float offset, float eps, float scale, void* lvl_pos, int lvl_pos_dt,
                                uint8_t* mask_u8, uint8_t* mask_bool, uint8_t* invalid_u8, float* enc_ref, float* proposals,
                                float* valid_ratios, float* vr4, float* box_scale, void* stream) {
  APE_CHECK_ARG(L >= 1 && L <= 8 && S > 0 && h > 0 && w > 0 && h <= S && w <= S, "ape_hip_geometry: bad sizes (L=%d S=%d h=%d w=%d)", L, S, h, w);
  APE_CHECK_ARG(level_hw && dim_t && level_embeds && lvl_pos && mask_u8 && mask_bool && invalid_u8 && enc_ref && proposals &&
                    valid_ratios && vr4 && box_scale, "ape_hip_geometry: null pointer");
  APE_CHECK_ARG(lvl_pos_dt == APE_DT_F32 || ape_is16(lvl_pos_dt), "ape_hip_geometry: lvl_pos dtype %d", lvl_pos_dt);
  GeoParams p;
  memset(&p, 0, sizeof(p));
  p.L = L; p.S = S; p.h = h; p.w = w; p.npf = npf; p.offset = offset; p.eps = eps; p.scale = scale;
  int tot = 0;
  for (int l = 0; l < L; ++l) { p.H[l] = level_hw[2 * l]; p.W[l] = level_hw[2 * l + 1]; p.start[l] = tot; tot += p.H[l] * p.W[l]; }
  p.T = tot;
  p.dim_t = dim_t; p.level_embeds = level_embeds; p.lvl_pos = lvl_pos; p.lp_dt = lvl_pos_dt; p.mask_u8 = mask_u8; p.mask_b = mask_bool;
  p.invalid_u8 = invalid_u8; p.enc_ref = enc_ref; p.proposals = proposals; p.valid_ratios = valid_ratios; p.vr4 = vr4; p.box_scale = box_scale;
  APE_LAUNCH(geometry_kernel, dim3(ceil_div(tot, 4)), dim3(256), 0, (hipStream_t)stream, p);
  APE_CHECK_LAUNCH("ape_hip_geometry");
  return 0;
}
