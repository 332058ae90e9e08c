// LayerNorm / GroupNorm kernels (wave-shuffle reductions, fp32 statistics, two-pass variance).
//
// LayerNorm replaces nn.LayerNorm call sites on the hot path (vit_eva_clip.py:29-35,129,264,509,522;
// detectron2 channel-LN used by SimpleFeaturePyramid vit_eva_clip.py:829-842 -- in our NHWC layout a
// channel-LN is a row LayerNorm; fuse_helper.py:224-225; detrex BaseTransformerLayer norms;
// deformable_transformer_vl.py:366,635,644).
// GroupNorm(32) replaces the neck / mask-head norms (detrex ChannelMapper, config
// ape_deta_vitl_eva02_clip_vlf_lsj1024_cp_16x4_1080k.py:42-55; deformable_detr_segm_vl.py:115-135)
// on token-major [HW, C] tensors: statistics over HW x (C/32) per group.
#include "common.h"
#include "../../include/ape_hip.h"

__device__ __forceinline__ float ln_act(float x, int act) {
  if (act == APE_ACT_RELU) return fmaxf(x, 0.f);
  if (act == APE_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
  return x;
}

// One wave per row, the row cached in registers: lane owns float4 groups (i*64 + lane), i < NV4 (C <= NV4*256).
// Any C is accepted (ragged last group handled element-wise); rows must be 16-byte aligned (ld % 4 == 0).
template <typename TX, typename TY, typename TA, int NV4>
__global__ __launch_bounds__(256) void layernorm_kernel(const ApeLayerNormArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const TX* x = reinterpret_cast<const TX*>(p.x) + (size_t)row * p.ldx;
  TY* y = reinterpret_cast<TY*>(p.y) + (size_t)row * p.ldy;
  const TA* add = p.add ? reinterpret_cast<const TA*>(p.add) + (size_t)row * p.ldadd : nullptr;
  TY* y2 = p.y2 ? reinterpret_cast<TY*>(p.y2) + (size_t)row * p.ldy2 : nullptr;
  const int C = p.C;
  const float invC = 1.f / (float)C;
  float v[NV4][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c + 3 < C) {
      ld4<TX>(x + c, v[i]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[i][r] = (c + r < C) ? ldf<TX>(x + c + r) : 0.f;
    }
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) * invC;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int c = (i * 64 + lane) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = (c + r < C) ? v[i][r] - mean : 0.f;
      q = fmaf(d, d, q);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) * invC + p.eps);
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c >= p.Cpad) continue;
    float o[4], o2[4];
    if (c + 3 < C) {   // fast path: whole float4 group inside the row
      float w4[4], b4[4];
      ld4<float>(p.w + c, w4);
      ld4<float>(p.b + c, b4);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = ln_act((v[i][r] - mean) * rstd * w4[r] + b4[r], p.act);
      st4<TY>(y + c, o);
      if (y2 != nullptr) {
        ld4<TA>(add + c, o2);
#pragma unroll
        for (int r = 0; r < 4; ++r) o2[r] += o[r];
        st4<TY>(y2 + c, o2);
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (c + r < C) {
        o[r] = ln_act((v[i][r] - mean) * rstd * p.w[c + r] + p.b[c + r], p.act);
        o2[r] = (y2 != nullptr) ? o[r] + ldf<TA>(add + c + r) : 0.f;
      } else {
        o[r] = 0.f; o2[r] = 0.f;   // K padding columns C..Cpad-1
      }
    }
    if (c + 3 < p.Cpad) {
      st4<TY>(y + c, o);
      if (y2 != nullptr) st4<TY>(y2 + c, o2);
    } else {
      for (int r = 0; r < 4 && c + r < p.Cpad; ++r) { stf<TY>(y + c + r, o[r]); if (y2 != nullptr) stf<TY>(y2 + c + r, o2[r]); }
    }
  }
  // Cpad beyond the register window (never the case for NV4*256 >= Cpad)
  for (int c = NV4 * 256 + lane; c < p.Cpad; c += 64) { stf<TY>(y + c, 0.f); if (y2 != nullptr) stf<TY>(y2 + c, 0.f); }
}

// generic fallback: three passes over the row (unaligned rows or C > 3072)
template <typename TX, typename TY, typename TA>
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const ApeLayerNormArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const TX* x = reinterpret_cast<const TX*>(p.x) + (size_t)row * p.ldx;
  TY* y = reinterpret_cast<TY*>(p.y) + (size_t)row * p.ldy;
  const TA* add = p.add ? reinterpret_cast<const TA*>(p.add) + (size_t)row * p.ldadd : nullptr;
  TY* y2 = p.y2 ? reinterpret_cast<TY*>(p.y2) + (size_t)row * p.ldy2 : nullptr;
  const int C = p.C;
  const float invC = 1.f / (float)C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += ldf<TX>(x + c);
  const float mean = wave_sum(s) * invC;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = ldf<TX>(x + c) - mean; q = fmaf(d, d, q); }
  const float rstd = rsqrtf(wave_sum(q) * invC + p.eps);
  for (int c = lane; c < C; c += 64) {
    const float o = ln_act((ldf<TX>(x + c) - mean) * rstd * p.w[c] + p.b[c], p.act);
    stf<TY>(y + c, o);
    if (y2 != nullptr) stf<TY>(y2 + c, o + ldf<TA>(add + c));
  }
  for (int c = C + lane; c < p.Cpad; c += 64) {
    stf<TY>(y + c, 0.f);
    if (y2 != nullptr) stf<TY>(y2 + c, 0.f);
  }
}

template <typename TX, typename TY, typename TA>
static int launch_ln(const ApeLayerNormArgs& p, hipStream_t s) {
  const dim3 grid(ceil_div(p.M, 4)), block(256);
  const bool vec = (p.ldx % 4 == 0) && (p.ldy % 4 == 0) && (((uintptr_t)p.x) % 16 == 0) && (((uintptr_t)p.y) % 16 == 0) &&
                   (((uintptr_t)p.w) % 16 == 0) && (((uintptr_t)p.b) % 16 == 0) &&
                   (!p.y2 || ((p.ldadd % 4 == 0) && (p.ldy2 % 4 == 0) && (((uintptr_t)p.add) % 16 == 0) &&
                              (((uintptr_t)p.y2) % 16 == 0)));
  const int need = p.Cpad > p.C ? p.Cpad : p.C;
  if (vec && need <= 256) APE_LAUNCH((layernorm_kernel<TX, TY, TA, 1>), grid, block, 0, s, p);
  else if (vec && need <= 512) APE_LAUNCH((layernorm_kernel<TX, TY, TA, 2>), grid, block, 0, s, p);
  else if (vec && need <= 1024) APE_LAUNCH((layernorm_kernel<TX, TY, TA, 4>), grid, block, 0, s, p);
  else if (vec && need <= 2048) APE_LAUNCH((layernorm_kernel<TX, TY, TA, 8>), grid, block, 0, s, p);
  else if (vec && need <= 3072) APE_LAUNCH((layernorm_kernel<TX, TY, TA, 12>), grid, block, 0, s, p);
  else APE_LAUNCH((layernorm_generic_kernel<TX, TY, TA>), grid, block, 0, s, p);
  return 0;
}

extern "C" int ape_hip_layernorm(const ApeLayerNormArgs* a, void* stream) {
  APE_CHECK_ARG(a && a->x && a->y && a->w && a->b, "ape_hip_layernorm: null pointer");
  APE_CHECK_ARG(a->M > 0 && a->C > 0 && a->Cpad >= a->C, "ape_hip_layernorm: bad shape");
  APE_CHECK_ARG((a->add == nullptr) == (a->y2 == nullptr), "ape_hip_layernorm: add and y2 go together");
  const ApeLayerNormArgs p = *a;
  hipStream_t s = (hipStream_t)stream;
  // one 16-bit storage type per call (bf16 | f16); key bits: x, y, add are 16-bit
  const int add_dt = p.add ? p.add_dt : p.y_dt;
  const int hk = APE_H16_KIND(p.x_dt, p.y_dt, add_dt);
  if (hk < 0) { ape_set_error("ape_hip_layernorm: dtypes must be f32 or ONE 16-bit type (x %d, y %d, add %d)", p.x_dt, p.y_dt, add_dt); return -1; }
  const int key = (ape_is16(p.x_dt) ? 4 : 0) + (ape_is16(p.y_dt) ? 2 : 0) + (ape_is16(add_dt) ? 1 : 0);
#define APE_NORM_CASES(H_)                                      \
  switch (key) {                                                \
    case 0: launch_ln<float, float, float>(p, s); break;        \
    case 1: launch_ln<float, float, H_>(p, s); break;           \
    case 2: launch_ln<float, H_, float>(p, s); break;           \
    case 3: launch_ln<float, H_, H_>(p, s); break;              \
    case 4: launch_ln<H_, float, float>(p, s); break;           \
    case 5: launch_ln<H_, float, H_>(p, s); break;              \
    case 6: launch_ln<H_, H_, float>(p, s); break;              \
    default: launch_ln<H_, H_, H_>(p, s); break;                \
  }
  if (hk == APE_DT_F16) { APE_NORM_CASES(f16_t) } else { APE_NORM_CASES(bf16_t) }
#undef APE_NORM_CASES
  APE_CHECK_LAUNCH("ape_hip_layernorm");
  return 0;
}

// ------------------------------------------------------------------------------------------
// GroupNorm(G groups) over token-major x[HW, C]:
//   pass 1 (gn_partial): per row-chunk (mean, M2) for each group            -> partial[nchunk][G][2]
//   pass 2 (gn_apply)  : every block re-combines the partials (Chan et al.), normalises its rows,
//                        optional ReLU, optional residual add AFTER the norm/act is NOT needed by the
//                        reference; an optional `add` tensor is added BEFORE storing (lateral + memory).
// C <= 256, C % G == 0, channels per group cg = C/G (8 for the reference), block = 256 threads:
// thread t owns channel t (C == 256) so loads are fully coalesced (each row = 512 B bf16).
// ------------------------------------------------------------------------------------------
#define GN_ROWS_PER_BLOCK 128

// thread layout for C % 8 == 0: (C/8) column groups of 8 channels (one 16-byte access) x 256/(C/8) row lanes
template <typename TX>
__global__ __launch_bounds__(256) void gn_partial_kernel(const TX* __restrict__ x, int ldx, int HW, int C, int G,
                                                         float* __restrict__ partial) {
  __shared__ float sh[8 * 256 + 256];     // row-lane partials [<= 8][C], then per-column totals
  __shared__ float smean[64];
  const int t = threadIdx.x;
  const int cg = C / G;
  const int r0 = blockIdx.x * GN_ROWS_PER_BLOCK;
  const int r1 = min(r0 + GN_ROWS_PER_BLOCK, HW);
  const int nrows = r1 - r0;
  const bool vec = (C % 8 == 0) && (256 % (C / 8) == 0) && (256 / (C / 8) <= 8) && (ldx % 8 == 0) && (((uintptr_t)x) % 16 == 0);
  const int ncg = vec ? C / 8 : 0, nrl = vec ? 256 / ncg : 0;
  const int cgi = vec ? t % ncg : 0, rl = vec ? t / ncg : 0;
  float* tot = sh + 8 * 256;
  for (int pass = 0; pass < 2; ++pass) {
    if (vec) {
      float a[8];
      float mu[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { a[c] = 0.f; mu[c] = pass ? smean[(cgi * 8 + c) / cg] : 0.f; }
      for (int r = r0 + rl; r < r1; r += nrl) {
        float v[8];
        ld8<TX>(x + (size_t)r * ldx + cgi * 8, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float d = v[c] - mu[c]; a[c] += pass ? d * d : d; }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) sh[rl * 256 + cgi * 8 + c] = a[c];
      __syncthreads();
      if (t < C) {
        float z = 0.f;
        for (int k = 0; k < nrl; ++k) z += sh[k * 256 + t];
        tot[t] = z;
      }
    } else {
      float z = 0.f;
      if (t < C) {
        const float m = pass ? smean[t / cg] : 0.f;
        for (int r = r0; r < r1; ++r) { const float d = ldf<TX>(x + (size_t)r * ldx + t) - m; z += pass ? d * d : d; }
      }
      tot[t] = z;
    }
    __syncthreads();
    if (t < G) {
      float z = 0.f;
      for (int j = 0; j < cg; ++j) z += tot[t * cg + j];
      if (pass == 0) smean[t] = z / (float)(nrows * cg);
      else {
        partial[((size_t)blockIdx.x * G + t) * 2 + 0] = smean[t];
        partial[((size_t)blockIdx.x * G + t) * 2 + 1] = z;
      }
    }
    __syncthreads();
  }
}

// one 64-lane block per group: lanes stride over the row chunks, Chan-combine, then butterfly-combine
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, int nchunk, int HW, int cg,
                                                         int G, float eps, float* __restrict__ stats) {
  const int g = blockIdx.x, lane = threadIdx.x;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int k = lane; k < nchunk; k += 64) {
    const int rows = min(GN_ROWS_PER_BLOCK, HW - k * GN_ROWS_PER_BLOCK);
    const float nb = (float)(rows * cg);
    const float mb = partial[((size_t)k * G + g) * 2 + 0];
    const float qb = partial[((size_t)k * G + g) * 2 + 1];
    const float nt = n + nb;
    const float delta = mb - mean;
    mean += delta * (nb / nt);
    m2 += qb + delta * delta * (n * nb / nt);
    n = nt;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), qb = __shfl_xor(m2, o, 64);
    const float nt = n + nb;
    if (nt > 0.f) {
      const float delta = mb - mean;
      // symmetric form so both partners compute the same result
      mean = (n * mean + nb * mb) / nt;
      m2 = m2 + qb + delta * delta * (n * nb / nt);
    }
    n = nt;
  }
  if (lane == 0) {
    stats[g * 2 + 0] = mean;
    stats[g * 2 + 1] = rsqrtf(m2 / n + eps);
  }
}

template <typename TX, typename TY, typename TA>
__global__ __launch_bounds__(256) void gn_apply_kernel(const TX* __restrict__ x, int ldx, int HW, int C, int G,
                                                       const float* __restrict__ stats, const float* __restrict__ w,
                                                       const float* __restrict__ b, int act, const TA* __restrict__ add,
                                                       int ldadd, TY* __restrict__ y, int ldy) {
  const int t = threadIdx.x;
  const int cg = C / G;
  const int r0 = blockIdx.x * GN_ROWS_PER_BLOCK;
  const int r1 = min(r0 + GN_ROWS_PER_BLOCK, HW);
  const bool vec = (C % 8 == 0) && (256 % (C / 8) == 0) && (ldx % 8 == 0) && (ldy % 8 == 0) && (((uintptr_t)x) % 16 == 0) &&
                   (((uintptr_t)y) % 16 == 0) && (add == nullptr || (ldadd % 8 == 0 && ((uintptr_t)add) % 16 == 0));
  if (vec) {
    const int ncg = C / 8, nrl = 256 / ncg;
    const int cgi = t % ncg, rl = t / ncg;
    float wt[8], bt[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int ch = cgi * 8 + c;
      const float m = stats[(ch / cg) * 2 + 0], rs = stats[(ch / cg) * 2 + 1];
      wt[c] = w[ch] * rs; bt[c] = b[ch] - m * rs * w[ch];
    }
    for (int r = r0 + rl; r < r1; r += nrl) {
      float v[8];
      ld8<TX>(x + (size_t)r * ldx + cgi * 8, v);
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = fmaf(v[c], wt[c], bt[c]);
      if (add != nullptr) {
        float av[8];
        ld8<TA>(add + (size_t)r * ldadd + cgi * 8, av);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] += av[c];
      }
      if (act == APE_ACT_RELU) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
      }
      st8<TY>(y + (size_t)r * ldy + cgi * 8, v);
    }
    return;
  }
  if (t >= C) return;
  const float m = stats[(t / cg) * 2 + 0], rs = stats[(t / cg) * 2 + 1];
  const float wt = w[t] * rs, bt = b[t] - m * rs * w[t];
  for (int r = r0; r < r1; ++r) {
    float v = fmaf(ldf<TX>(x + (size_t)r * ldx + t), wt, bt);
    if (add != nullptr) v += ldf<TA>(add + (size_t)r * ldadd + t);
    if (act == APE_ACT_RELU) v = fmaxf(v, 0.f);
    stf<TY>(y + (size_t)r * ldy + t, v);
  }
}

template <typename TX, typename TY, typename TA>
static void launch_gn(const ApeGroupNormArgs& p, hipStream_t s) {
  const int nchunk = ceil_div(p.HW, GN_ROWS_PER_BLOCK);
  float* stats = p.workspace + (size_t)nchunk * p.G * 2;
  APE_LAUNCH((gn_partial_kernel<TX>), dim3(nchunk), dim3(256), 0, s, (const TX*)p.x, p.ldx, p.HW, p.C, p.G,
                     p.workspace);
  APE_LAUNCH(gn_finalize_kernel, dim3(p.G), dim3(64), 0, s, p.workspace, nchunk, p.HW, p.C / p.G, p.G, p.eps, stats);
  APE_LAUNCH((gn_apply_kernel<TX, TY, TA>), dim3(nchunk), dim3(256), 0, s, (const TX*)p.x, p.ldx, p.HW, p.C, p.G,
                     stats, p.w, p.b, p.act, (const TA*)p.add, p.ldadd, (TY*)p.y, p.ldy);
}

extern "C" int ape_hip_groupnorm_workspace_floats(int HW, int G) { return ceil_div(HW, GN_ROWS_PER_BLOCK) * G * 2 + G * 2; }

extern "C" int ape_hip_groupnorm(const ApeGroupNormArgs* a, void* stream) {
  APE_CHECK_ARG(a && a->x && a->y && a->w && a->b && a->workspace, "ape_hip_groupnorm: null pointer");
  APE_CHECK_ARG(a->C > 0 && a->C <= 256 && a->G > 0 && a->G <= 64 && a->C % a->G == 0 && a->HW > 0,
                "ape_hip_groupnorm: need C <= 256, G <= 64, C %% G == 0");
  const ApeGroupNormArgs p = *a;
  hipStream_t s = (hipStream_t)stream;
  // one 16-bit storage type per call (bf16 | f16); key bits: x, y, add are 16-bit
  const int add_dt = p.add ? p.add_dt : p.y_dt;
  const int hk = APE_H16_KIND(p.x_dt, p.y_dt, add_dt);
  if (hk < 0) { ape_set_error("ape_hip_groupnorm: dtypes must be f32 or ONE 16-bit type (x %d, y %d, add %d)", p.x_dt, p.y_dt, add_dt); return -1; }
  const int key = (ape_is16(p.x_dt) ? 4 : 0) + (ape_is16(p.y_dt) ? 2 : 0) + (ape_is16(add_dt) ? 1 : 0);
#define APE_NORM_CASES(H_)                                      \
  switch (key) {                                                \
    case 0: launch_gn<float, float, float>(p, s); break;        \
    case 1: launch_gn<float, float, H_>(p, s); break;           \
    case 2: launch_gn<float, H_, float>(p, s); break;           \
    case 3: launch_gn<float, H_, H_>(p, s); break;              \
    case 4: launch_gn<H_, float, float>(p, s); break;           \
    case 5: launch_gn<H_, float, H_>(p, s); break;              \
    case 6: launch_gn<H_, H_, float>(p, s); break;              \
    default: launch_gn<H_, H_, H_>(p, s); break;                \
  }
  if (hk == APE_DT_F16) { APE_NORM_CASES(f16_t) } else { APE_NORM_CASES(bf16_t) }
#undef APE_NORM_CASES
  APE_CHECK_LAUNCH("ape_hip_groupnorm");
  return 0;
}


// per-row LayerNorm statistics only (the normalisation itself is folded into the consuming GEMM, see ApeGemmArgs):
// one wave per row, two passes (the second one re-reads the 5 KB row from L1/L2)
template <typename TX>
__global__ __launch_bounds__(256) void row_stats_kernel(const TX* __restrict__ x, int ldx, int M, int C, float eps,
                                                        float* __restrict__ rowscale, float* __restrict__ rowshift) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const TX* xr = x + (size_t)row * ldx;
  const int C8 = (ldx % 8 == 0 && ((uintptr_t)x) % 16 == 0) ? (C & ~7) : 0;
  float s = 0.f;
  for (int c = lane * 8; c < C8; c += 512) {
    float v[8];
    ld8<TX>(xr + c, v);
    s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (int c = C8 + lane; c < C; c += 64) s += ldf<TX>(xr + c);
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane * 8; c < C8; c += 512) {
    float v[8];
    ld8<TX>(xr + c, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
  }
  for (int c = C8 + lane; c < C; c += 64) { const float d = ldf<TX>(xr + c) - mean; q = fmaf(d, d, q); }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { rowscale[row] = rstd; rowshift[row] = -mean * rstd; }
}

extern "C" int ape_hip_row_stats(const void* x, int ldx, int dt, int M, int C, float eps, float* rowscale, float* rowshift,
                                 void* stream) {
  APE_CHECK_ARG(x && rowscale && rowshift && M > 0 && C > 0, "ape_hip_row_stats: bad args");
  const dim3 grid(ceil_div(M, 4)), block(256);
  if (dt == APE_DT_F16) APE_LAUNCH(row_stats_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, (const f16_t*)x, ldx, M, C, eps, rowscale, rowshift);
  else if (dt == APE_DT_BF16) APE_LAUNCH(row_stats_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, ldx, M, C, eps, rowscale, rowshift);
  else APE_LAUNCH(row_stats_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, ldx, M, C, eps, rowscale, rowshift);
  APE_CHECK_LAUNCH("ape_hip_row_stats");
  return 0;
}

// ------------------------------------------------------------------------------------------
// Post-norm residual step of the EVA-CLIP "postnorm" blocks (ViT-e; ape/modeling/backbone/vit_eva_clip.py:505-523 with
// postnorm=True):  x <- x + LayerNorm(t)  on the fp32 residual stream, in place, plus the copy of the new stream in the GEMM
// operand type that the next linear reads (the stream is not normalised in this block flavour, so it stays fp32; the copy saves
// a cast launch).  t == NULL: only the copy (start of the stack).  One wave per row, the row of t in registers; C % 4 == 0,
// C <= 2048.
// ------------------------------------------------------------------------------------------
template <typename TT, typename TC, int NV4>
__global__ __launch_bounds__(256) void postnorm_residual_kernel(const TT* __restrict__ t, int ldt, const float* __restrict__ w,
                                                                const float* __restrict__ b, float eps, float* __restrict__ stream,
                                                                int lds, TC* __restrict__ copy, int ldc, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float* xs = stream + (size_t)row * lds;
  TC* cp = copy != nullptr ? copy + (size_t)row * ldc : nullptr;
  if (t == nullptr) {
    for (int c = lane * 4; c < C; c += 256) {
      float v[4];
      ld4<float>(xs + c, v);
      st4<TC>(cp + c, v);
    }
    return;
  }
  const TT* tr = t + (size_t)row * ldt;
  const float invC = 1.f / (float)C;
  float v[NV4][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) ld4<TT>(tr + c, v[i]);
    else { v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f; }
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) * invC;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = v[i][r] - mean; q = fmaf(d, d, q); }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) * invC + eps);
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c >= C) continue;
    float w4[4], b4[4], x4[4];
    ld4<float>(w + c, w4);
    ld4<float>(b + c, b4);
    ld4<float>(xs + c, x4);
#pragma unroll
    for (int r = 0; r < 4; ++r) x4[r] += (v[i][r] - mean) * rstd * w4[r] + b4[r];
    st4<float>(xs + c, x4);
    if (cp != nullptr) st4<TC>(cp + c, x4);
  }
}

template <typename TT, typename TC>
static void launch_postnorm(const void* t, int ldt, const float* w, const float* b, float eps, float* stream, int lds, void* copy, int ldc,
                            int M, int C, hipStream_t s) {
  const dim3 grid(ceil_div(M, 4)), block(256);
  if (C <= 256) APE_LAUNCH((postnorm_residual_kernel<TT, TC, 1>), grid, block, 0, s, (const TT*)t, ldt, w, b, eps, stream, lds, (TC*)copy, ldc, M, C);
  else if (C <= 512) APE_LAUNCH((postnorm_residual_kernel<TT, TC, 2>), grid, block, 0, s, (const TT*)t, ldt, w, b, eps, stream, lds, (TC*)copy, ldc, M, C);
  else if (C <= 1024) APE_LAUNCH((postnorm_residual_kernel<TT, TC, 4>), grid, block, 0, s, (const TT*)t, ldt, w, b, eps, stream, lds, (TC*)copy, ldc, M, C);
  else APE_LAUNCH((postnorm_residual_kernel<TT, TC, 8>), grid, block, 0, s, (const TT*)t, ldt, w, b, eps, stream, lds, (TC*)copy, ldc, M, C);
}

extern "C" int ape_hip_postnorm_residual(const void* t, int ldt, int t_dt, const float* w, const float* b, float eps, float* stream,
                                         int lds, void* copy, int ldc, int copy_dt, int M, int C, void* stream_) {
  APE_CHECK_ARG(stream && M > 0 && C > 0 && C % 4 == 0 && C <= 2048 && lds % 4 == 0, "ape_hip_postnorm_residual: C %% 4 == 0, C <= 2048");
  APE_CHECK_ARG(t != nullptr || copy != nullptr, "ape_hip_postnorm_residual: nothing to do");
  APE_CHECK_ARG(t == nullptr || (w && b && ldt % 4 == 0 && ((uintptr_t)t) % 16 == 0), "ape_hip_postnorm_residual: t / w / b");
  APE_CHECK_ARG(copy == nullptr || (ldc % 4 == 0 && ((uintptr_t)copy) % 8 == 0), "ape_hip_postnorm_residual: copy alignment");
  APE_CHECK_ARG(((uintptr_t)stream) % 16 == 0, "ape_hip_postnorm_residual: stream alignment");
  hipStream_t s = (hipStream_t)stream_;
  const int hk = APE_H16_KIND(t != nullptr ? t_dt : APE_DT_F32, copy != nullptr ? copy_dt : APE_DT_F32);
  APE_CHECK_ARG(hk >= 0, "ape_hip_postnorm_residual: t and the copy are f32 or ONE 16-bit type");
  const bool tb = t != nullptr && ape_is16(t_dt), cb = copy != nullptr && ape_is16(copy_dt);
#define APE_PN_CASES(H_)                                                                              \
  if (tb && cb) launch_postnorm<H_, H_>(t, ldt, w, b, eps, stream, lds, copy, ldc, M, C, s);          \
  else if (tb) launch_postnorm<H_, float>(t, ldt, w, b, eps, stream, lds, copy, ldc, M, C, s);        \
  else if (cb) launch_postnorm<float, H_>(t, ldt, w, b, eps, stream, lds, copy, ldc, M, C, s);        \
  else launch_postnorm<float, float>(t, ldt, w, b, eps, stream, lds, copy, ldc, M, C, s);
  if (hk == APE_DT_F16) { APE_PN_CASES(f16_t) } else { APE_PN_CASES(bf16_t) }
#undef APE_PN_CASES
  APE_CHECK_LAUNCH("ape_hip_postnorm_residual");
  return 0;
}
