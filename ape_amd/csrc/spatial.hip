// Spatial gather kernels that turn the reference's NCHW convolutions into token-major GEMM operands.
//
// Everything on the hot path is kept token-major ([H*W, C], channels contiguous = NHWC), and the ViT token
// stream is kept in WINDOW-MAJOR order for the whole backbone (so window_partition / window_unpartition,
// utils_eva02.py:19-63, cost nothing).  These kernels absorb the permutations:
//   patchify     : uint8/float image -> normalised, zero-padded 16x16 patch rows (PatchEmbed im2col,
//                  utils_eva02.py:208-216 + preprocess_image deformable_detr_segm_vl.py:846-855)
//   im2col3x3    : 3x3 / pad 1 convolution operand with an optional raster->row permutation of the source
//                  (SimpleFeaturePyramid 3x3 convs vit_eva_clip.py:835-842, output_conv deformable_detr_segm_vl.py:122-131)
//   maxpool2x2   : nn.MaxPool2d(2,2) of simfp_5 (vit_eva_clip.py:822-823)
//   gather_rows  : out[r] = in[idx[r]]  (LastLevelMaxPool = stride-2 subsample vit_eva_clip.py:907-912, query gathers)
#include "common.h"
#include "../../include/ape_hip.h"

// one thread per 8 output elements (one (c,ky) half-row of a patch: 8 consecutive kx)
template <typename TO>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int h, int w, const int* __restrict__ tok2raster,
                                                       int Wt, int ntok, float m0, float m1, float m2, float s0, float s1,
                                                       float s2, TO* __restrict__ out, int ldo) {
  const int gid = blockIdx.x * 256 + threadIdx.x;  // over ntok * 96 chunks
  const int t = gid / 96, ch = gid % 96;
  if (t >= ntok) return;
  const int c = ch / 32, rem = ch % 32, ky = rem >> 1, kx0 = (rem & 1) * 8;
  const int r = tok2raster ? tok2raster[t] : t;
  const int ty = r / Wt, tx = r % Wt;
  const int y = ty * 16 + ky, x0 = tx * 16 + kx0;
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
  const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  float v[8];
  const float* row = img + ((size_t)c * h + y) * w;
  if (y < h && x0 + 7 < w && (w & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0) {   // whole 8-pixel run inside the image, 16-byte aligned: two vector loads
    const float4 a = *reinterpret_cast<const float4*>(row + x0), b = *reinterpret_cast<const float4*>(row + x0 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (v[i] - mean) / sd;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int x = x0 + i;
      v[i] = (y < h && x < w) ? (row[x] - mean) / sd : 0.f;
    }
  }
  st8<TO>(out + (size_t)t * ldo + c * 256 + ky * 16 + kx0, v);
}

extern "C" int ape_hip_patchify(const float* img, int h, int w, const int* tok2raster, int Ht, int Wt, const float* mean3,
                                const float* std3, void* out, int ldo, int out_dt, void* stream) {
  APE_CHECK_ARG(img && out && mean3 && std3 && h > 0 && w > 0 && Ht > 0 && Wt > 0, "ape_hip_patchify: bad args");
  APE_CHECK_ARG(h <= Ht * 16 && w <= Wt * 16 && ldo % 8 == 0 && ((uintptr_t)out) % 16 == 0, "ape_hip_patchify: image larger than the token grid or unaligned out");
  const int ntok = Ht * Wt;
  const dim3 grid(ceil_div(ntok * 96, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (out_dt == APE_DT_F16) APE_LAUNCH(patchify_kernel<f16_t>, grid, block, 0, s, img, h, w, tok2raster, Wt, ntok, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (f16_t*)out, ldo);
  else if (out_dt == APE_DT_BF16) APE_LAUNCH(patchify_kernel<bf16_t>, grid, block, 0, s, img, h, w, tok2raster, Wt, ntok, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (bf16_t*)out, ldo);
  else
    APE_LAUNCH(patchify_kernel<float>, grid, block, 0, s, img, h, w, tok2raster, Wt, ntok, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (float*)out, ldo);
  APE_CHECK_LAUNCH("ape_hip_patchify");
  return 0;
}

// one thread per 8 channels of one (pixel, tap); C % 8 == 0
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* __restrict__ x, int ldx, const int* __restrict__ perm, int H,
                                                        int W, int C, T* __restrict__ out, int ldo) {
  const int cpr = C >> 3;  // chunks per tap
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)H * W * 9 * cpr;
  if (gid >= total) return;
  const int cc = (int)(gid % cpr);
  const int tap = (int)((gid / cpr) % 9);
  const int pix = (int)(gid / ((size_t)cpr * 9));
  const int y = pix / W + tap / 3 - 1, xx = pix % W + tap % 3 - 1;
  float v[8];
  if (y >= 0 && y < H && xx >= 0 && xx < W) {
    const int rs = y * W + xx;
    const int row = perm ? perm[rs] : rs;
    ld8<T>(x + (size_t)row * ldx + cc * 8, v);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  st8<T>(out + (size_t)pix * ldo + tap * C + cc * 8, v);
}

extern "C" int ape_hip_im2col3x3(const void* x, int ldx, const int* perm, int H, int W, int C, void* out, int ldo, int dt,
                                 void* stream) {
  APE_CHECK_ARG(x && out && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "ape_hip_im2col3x3: bad args");
  const size_t total = (size_t)H * W * 9 * (C / 8);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dt == APE_DT_F16) APE_LAUNCH(im2col3x3_kernel<f16_t>, grid, block, 0, s, (const f16_t*)x, ldx, perm, H, W, C, (f16_t*)out, ldo);
  else if (dt == APE_DT_BF16) APE_LAUNCH(im2col3x3_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, ldx, perm, H, W, C, (bf16_t*)out, ldo);
  else APE_LAUNCH(im2col3x3_kernel<float>, grid, block, 0, s, (const float*)x, ldx, perm, H, W, C, (float*)out, ldo);
  APE_CHECK_LAUNCH("ape_hip_im2col3x3");
  return 0;
}

// 2x2/2 max pool of a [H*W, C] map (source rows through perm) -> raster-ordered [(H/2)*(W/2), C]
template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const T* __restrict__ x, int ldx, const int* __restrict__ perm, int H,
                                                         int W, int C, T* __restrict__ out, int ldo) {
  const int cpr = C >> 3;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int Ho = H / 2, Wo = W / 2;
  if (gid >= (size_t)Ho * Wo * cpr) return;
  const int cc = (int)(gid % cpr);
  const int pix = (int)(gid / cpr);
  const int yo = pix / Wo, xo = pix % Wo;
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int rs = (yo * 2 + (d >> 1)) * W + xo * 2 + (d & 1);
    const int row = perm ? perm[rs] : rs;
    float v[8];
    ld8<T>(x + (size_t)row * ldx + cc * 8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], v[i]);
  }
  st8<T>(out + (size_t)pix * ldo + cc * 8, m);
}

extern "C" int ape_hip_maxpool2x2(const void* x, int ldx, const int* perm, int H, int W, int C, void* out, int ldo, int dt,
                                  void* stream) {
  APE_CHECK_ARG(x && out && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "ape_hip_maxpool2x2: bad args");
  const size_t total = (size_t)(H / 2) * (W / 2) * (C / 8);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dt == APE_DT_F16) APE_LAUNCH(maxpool2x2_kernel<f16_t>, grid, block, 0, s, (const f16_t*)x, ldx, perm, H, W, C, (f16_t*)out, ldo);
  else if (dt == APE_DT_BF16) APE_LAUNCH(maxpool2x2_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, ldx, perm, H, W, C, (bf16_t*)out, ldo);
  else APE_LAUNCH(maxpool2x2_kernel<float>, grid, block, 0, s, (const float*)x, ldx, perm, H, W, C, (float*)out, ldo);
  APE_CHECK_LAUNCH("ape_hip_maxpool2x2");
  return 0;
}

template <typename T, typename TI>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ x, int ldx, const TI* __restrict__ idx, int n,
                                                          int C, T* __restrict__ out, int ldo) {
  const int cpr = C >> 3;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)n * cpr) return;
  const int cc = (int)(gid % cpr), r = (int)(gid / cpr);
  float v[8];
  ld8<T>(x + (size_t)idx[r] * ldx + cc * 8, v);
  st8<T>(out + (size_t)r * ldo + cc * 8, v);
}

// rows of any width / stride (class-score rows [Q, K] with K = 133, 1203 ...: the panoptic branch gathers its kept queries' logits): one
// element per thread
template <typename T, typename TI>
__global__ __launch_bounds__(256) void gather_rows_any_kernel(const T* __restrict__ x, int ldx, const TI* __restrict__ idx, int n,
                                                              int C, T* __restrict__ out, int ldo) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)n * C) return;
  const int c = (int)(gid % C), r = (int)(gid / C);
  out[(size_t)r * ldo + c] = x[(size_t)idx[r] * ldx + c];
}

template <typename TI>
static int gather_rows_launch(const char* what, const void* x, int ldx, const TI* idx, int n, int C, void* out, int ldo, int dt, void* stream) {
  APE_CHECK_ARG(x && idx && out && n > 0 && C > 0 && ldx >= C && ldo >= C, "%s: bad args", what);
  hipStream_t s = (hipStream_t)stream;
  const int esz = dt == APE_DT_F32 ? 4 : 2;
  const bool vec = C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)x) % 16 == 0 && ((uintptr_t)out) % 16 == 0 && ((size_t)ldx * esz) % 16 == 0;
  if (vec) {
    const size_t total = (size_t)n * (C / 8);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (dt == APE_DT_F16) APE_LAUNCH((gather_rows_kernel<f16_t, TI>), grid, block, 0, s, (const f16_t*)x, ldx, idx, n, C, (f16_t*)out, ldo);
    else if (dt == APE_DT_BF16) APE_LAUNCH((gather_rows_kernel<bf16_t, TI>), grid, block, 0, s, (const bf16_t*)x, ldx, idx, n, C, (bf16_t*)out, ldo);
    else APE_LAUNCH((gather_rows_kernel<float, TI>), grid, block, 0, s, (const float*)x, ldx, idx, n, C, (float*)out, ldo);
  } else {
    const size_t total = (size_t)n * C;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (dt == APE_DT_F32) APE_LAUNCH((gather_rows_any_kernel<float, TI>), grid, block, 0, s, (const float*)x, ldx, idx, n, C, (float*)out, ldo);
    else APE_LAUNCH((gather_rows_any_kernel<uint16_t, TI>), grid, block, 0, s, (const uint16_t*)x, ldx, idx, n, C, (uint16_t*)out, ldo);   // either 16-bit type: a copy
  }
  return 0;
}

extern "C" int ape_hip_gather_rows(const void* x, int ldx, const int* idx, int n, int C, void* out, int ldo, int dt, void* stream) {
  const int rc = gather_rows_launch<int>("ape_hip_gather_rows", x, ldx, idx, n, C, out, ldo, dt, stream);
  if (rc != 0) return rc;
  APE_CHECK_LAUNCH("ape_hip_gather_rows");
  return 0;
}

// the same with int64 row indices (what the selection kernels hand out: torch's index dtype)
extern "C" int ape_hip_gather_rows_i64(const void* x, int ldx, const int64_t* idx, int n, int C, void* out, int ldo, int dt, void* stream) {
  const int rc = gather_rows_launch<int64_t>("ape_hip_gather_rows_i64", x, ldx, idx, n, C, out, ldo, dt, stream);
  if (rc != 0) return rc;
  APE_CHECK_LAUNCH("ape_hip_gather_rows_i64");
  return 0;
}

// ---- token embedding of the CLIP text tower (ape/modeling/text/eva02_clip/transformer.py:724-726,
//      clip_wrapper_eva02.py:136-138): out[b * Lp + t] = table[tokens[b, t]] + pos[t] for t < L, zero rows for L <= t < Lp
template <typename T>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int32_t* __restrict__ tokens, int ldt, const T* __restrict__ table,
                                                           int ldtab, const T* __restrict__ pos, int ldpos, float* __restrict__ out,
                                                           int ldo, int L, int Lp, int W, int vocab) {
  const int row = blockIdx.x, b = row / Lp, t = row % Lp;
  float* o = out + (size_t)row * ldo;
  if (t >= L) {
    for (int c = threadIdx.x * 4; c < W; c += 1024) *reinterpret_cast<float4*>(o + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  int tok = tokens[(size_t)b * ldt + t];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  const T* e = table + (size_t)tok * ldtab;
  const T* p = pos + (size_t)t * ldpos;
  for (int c = threadIdx.x * 4; c < W; c += 1024) {
    float a[4], q[4];
    ld4<T>(e + c, a);
    ld4<T>(p + c, q);
    // summed in fp32 (the residual stream is fp32 here; the reference adds in the parameter dtype)
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = a[i] + q[i];
    *reinterpret_cast<float4*>(o + c) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

extern "C" int ape_hip_embed_tokens(const int32_t* tokens, int ldt, const void* table, int ldtab, const void* pos, int ldpos, int dt,
                                    float* out, int ldo, int B, int L, int Lp, int W, int vocab, void* stream) {
  APE_CHECK_ARG(tokens && table && pos && out && B > 0 && L > 0 && Lp >= L && vocab > 0, "ape_hip_embed_tokens: bad arguments");
  APE_CHECK_ARG(W % 4 == 0 && ldtab % 4 == 0 && ldpos % 4 == 0 && ldo % 4 == 0, "ape_hip_embed_tokens: widths must be multiples of 4");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(B * Lp), block(256);
  if (dt == APE_DT_F16) APE_LAUNCH(embed_tokens_kernel<f16_t>, grid, block, 0, s, tokens, ldt, (const f16_t*)table, ldtab, (const f16_t*)pos,
                       ldpos, out, ldo, L, Lp, W, vocab);
  else if (dt == APE_DT_BF16) APE_LAUNCH(embed_tokens_kernel<bf16_t>, grid, block, 0, s, tokens, ldt, (const bf16_t*)table, ldtab, (const bf16_t*)pos,
                       ldpos, out, ldo, L, Lp, W, vocab);
  else
    APE_LAUNCH(embed_tokens_kernel<float>, grid, block, 0, s, tokens, ldt, (const float*)table, ldtab, (const float*)pos, ldpos,
                       out, ldo, L, Lp, W, vocab);
  APE_CHECK_LAUNCH("ape_hip_embed_tokens");
  return 0;
}
