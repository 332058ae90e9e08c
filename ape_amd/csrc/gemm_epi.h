// Epilogue arithmetic shared by the GEMM kernels (gemm.hip, gemm_p8.hip): alpha, folded LayerNorm terms, row masks, bias,
// RoPE, activation / SwiGLU, clamp, residual -- in the order documented at ApeGemmArgs (include/ape_hip.h).
#pragma once
#include "common.h"
#include "../../include/ape_hip.h"

typedef ApeGemmArgs GemmParams;

__device__ __forceinline__ float act_fn(float x, int act) {
  if (act == APE_ACT_RELU) return fmaxf(x, 0.f);
  if (act == APE_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
  if (act == APE_ACT_SILU) return x / (1.f + __expf(-x));
  return x;
}

template <typename T>
__device__ __forceinline__ void store_n(T* dst, const float* v, int cnt, bool vec) {
  if (vec && cnt == 4) {
    st4<T>(dst, v);
  } else {
    for (int r = 0; r < cnt; ++r) stf<T>(dst + r, v[r]);
  }
}

// Epilogue arithmetic for 4 consecutive output columns n0..n0+3 (n0 % 4 == 0) of row m (m < M, n0 < N).
// On return v[0..cnt) are the final values for output columns ocol..ocol+cnt (SwiGLU halves the column index).
// p.vec_ok bits (launcher): 1 = C / residual rows allow 16-byte accesses, 2 = bias is 16-byte aligned, 16 = bias indexed by row,
// 4 = RoPE tables 16-byte aligned, power-of-two head dim, rope_rows >= M or a power of two (masks replace the modulo).
// For short-K GEMMs (the ViT at M = 4096) this code is as long as the main loop, so it is written for few VALU ops:
// vector loads for bias / RoPE / residual, uniform conditions tested once per quad.
// H = the 16-bit storage type of the launch (bf16_t | f16_t): what a 2-byte residual / output holds.
template <typename H = bf16_t>
__device__ __forceinline__ void epi_n4_values(const GemmParams& p, int m, int n0, float v[4], int& ocol, int& cnt) {
  const bool masked = p.rowmask != nullptr && p.rowmask[m] != 0;
  const bool full = n0 + 3 < p.N;
  if (p.alpha != 1.f) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
  }
  if (p.rowscale != nullptr) {             // folded LayerNorm of the A operand: acc * rstd_m - rstd_m mean_m * rowsum(W')_n
    const float rs = p.rowscale[m], sh = p.rowshift[m];
    if ((p.vec_ok & 8) && full) {
      const float4 c = *reinterpret_cast<const float4*>(p.colvec + n0);
      v[0] = fmaf(v[0], rs, sh * c.x); v[1] = fmaf(v[1], rs, sh * c.y); v[2] = fmaf(v[2], rs, sh * c.z); v[3] = fmaf(v[3], rs, sh * c.w);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n0 + r < p.N) v[r] = fmaf(v[r], rs, sh * p.colvec[n0 + r]);
    }
  }
  if (masked && p.mask_mode == APE_MASK_ZERO_INPUT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = 0.f;
  }
  if (p.bias != nullptr) {
    if (p.vec_ok & 16) {                     // transposed problem (launcher exchanged the operands): bias follows the row
      const float b = p.bias[m];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += b;
    } else if ((p.vec_ok & 2) && full) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n0);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n0 + r < p.N) v[r] += p.bias[n0 + r];
    }
  }
  if (p.rope_cos != nullptr && n0 < p.rope_cols) {
    const int hd = p.rope_hd;
    float c[4], sn[4];
    if (p.vec_ok & 4) {
      const int rmask = p.rope_rows >= p.M ? 0x7fffffff : p.rope_rows - 1;   // rows >= M or a power of two
      const size_t trow = (size_t)(m & rmask) * hd + (n0 & (hd - 1));
      const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + trow), s4 = *reinterpret_cast<const float4*>(p.rope_sin + trow);
      c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w; sn[0] = s4.x; sn[1] = s4.y; sn[2] = s4.z; sn[3] = s4.w;
    } else {
      const size_t trow = (size_t)(m % p.rope_rows) * hd + n0 % hd;
#pragma unroll
      for (int r = 0; r < 4; ++r) { c[r] = p.rope_cos[trow + r]; sn[r] = p.rope_sin[trow + r]; }
    }
    const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
    // t*cos + rotate_half(t)*sin with rotate_half pairs (2i,2i+1) -> (-x[2i+1], x[2i])
    v[0] = x0 * c[0] - x1 * sn[0];
    v[1] = x1 * c[1] + x0 * sn[1];
    v[2] = x2 * c[2] - x3 * sn[2];
    v[3] = x3 * c[3] + x2 * sn[3];
  }
  if (p.act == APE_ACT_SWIGLU) {
    // interleaved (gate, up) pairs -> N/2 output columns
    const float o0 = (v[0] / (1.f + __expf(-v[0]))) * v[1];
    const float o1 = (v[2] / (1.f + __expf(-v[2]))) * v[3];
    v[0] = o0; v[1] = o1;
    ocol = n0 >> 1;
    cnt = (ocol + 1 < (p.N >> 1)) ? 2 : 1;
    return;
  }
  ocol = n0;
  cnt = full ? 4 : (p.N - n0);
  if (p.act == APE_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  } else if (p.act != APE_ACT_NONE) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = act_fn(v[r], p.act);
  }
  if (p.clamp > 0.f) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r], -p.clamp), p.clamp);
  }
  if (p.residual != nullptr) {
    const size_t roff = (size_t)m * p.ldr + n0;
    const bool vec = (p.vec_ok & 1) != 0;
    float rv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.res_dt == APE_DT_F32) {
      const float* rp = reinterpret_cast<const float*>(p.residual) + roff;
      if (vec && cnt == 4) ld4<float>(rp, rv); else for (int r = 0; r < cnt; ++r) rv[r] = rp[r];
    } else {
      const H* rp = reinterpret_cast<const H*>(p.residual) + roff;
      if (vec && cnt == 4) ld4<H>(rp, rv); else for (int r = 0; r < cnt; ++r) rv[r] = ldf<H>(rp + r);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += rv[r];
  }
  if (masked && p.mask_mode == APE_MASK_ZERO_OUTPUT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = 0.f;
  }
}

// direct (register -> global) epilogue: 4 consecutive output columns of row m
template <typename H = bf16_t>
__device__ __forceinline__ void epi_n4(const GemmParams& p, int m, int n0, float v[4]) {
  if (m >= p.M || n0 >= p.N) return;
  int ocol, cnt;
  epi_n4_values<H>(p, m, n0, v, ocol, cnt);
  const bool vec = (p.vec_ok & 1) != 0 && cnt == 4;
  const size_t off = (size_t)m * p.ldc + ocol;
  if (p.out_dt == APE_DT_F32) store_n<float>(reinterpret_cast<float*>(p.C) + off, v, cnt, vec);
  else if (cnt == 2 && (p.vec_ok & 1)) *reinterpret_cast<uint32_t*>(reinterpret_cast<H*>(p.C) + off) = h16<H>::pack2(v[0], v[1]);
  else store_n<H>(reinterpret_cast<H*>(p.C) + off, v, cnt, vec);
}

__device__ __forceinline__ void epi_m4_values(const GemmParams& p, int n, float v[4]) {
  const float b = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = act_fn(v[r] * p.alpha + b, p.act);
}

// transposed output C^T[n][m0..m0+3]  (bias by n, activation, no residual/rope/mask)
template <typename H = bf16_t>
__device__ __forceinline__ void epi_m4(const GemmParams& p, int m0, int n, float v[4]) {
  if (n >= p.N || m0 >= p.M) return;
  epi_m4_values(p, n, v);
  const int cnt = (p.M - m0) < 4 ? (p.M - m0) : 4;
  const size_t off = (size_t)n * p.ldc + m0;
  if (p.out_dt == APE_DT_F32) store_n<float>(reinterpret_cast<float*>(p.C) + off, v, cnt, (p.vec_ok & 1) != 0);
  else store_n<H>(reinterpret_cast<H*>(p.C) + off, v, cnt, (p.vec_ok & 1) != 0);
}

// ------------------------------------------------------------------------------------------
// Lean epilogue for kernels whose lanes own W (8 | 16) CONSECUTIVE output columns nb .. nb+W-1 of a row (gemm_p8.hip).
// Same arithmetic and order as epi_n4_values, restricted to the combinations the large GEMMs of the forward use and
// specialised at compile time (the generic quad code inlined 32x per lane was longer than the main loop and cost
// instruction-cache misses on every tile):  [folded-LayerNorm row/column terms] -> + bias -> [RoPE] -> [ReLU | SwiGLU]
// -> + residual.  Preconditions (checked once per launch by epi_fast_ok): alpha == 1, no row mask, no clamp, vector-
// aligned bias / colvec / RoPE tables / residual rows, nb % W == 0, nb + W <= N.
// ------------------------------------------------------------------------------------------
#define EPI_ACT_NONE 0
#define EPI_ACT_RELU 1
#define EPI_ACT_SWIGLU 2

__device__ __forceinline__ bool epi_fast_ok(const GemmParams& p) {
  if (p.alpha != 1.f || p.rowmask != nullptr || p.clamp > 0.f) return false;
  if (!(p.act == APE_ACT_NONE || p.act == APE_ACT_RELU || p.act == APE_ACT_SWIGLU)) return false;
  if (p.bias != nullptr && !(p.vec_ok & (2 | 16))) return false;
  if (p.rope_cos != nullptr && (!(p.vec_ok & 4) || p.rope_hd % 16 != 0 || p.rope_cols % 16 != 0 || p.act != APE_ACT_NONE)) return false;
  if (p.rowscale != nullptr && (!(p.vec_ok & 8) || p.act != APE_ACT_NONE)) return false;
  if (p.rope_cos != nullptr && p.rowscale != nullptr) return false;   // no RoPE + folded-LayerNorm specialisation: the generic epilogue applies both
  if (p.residual != nullptr && (!(p.vec_ok & 1) || p.ldr % 8 != 0 || p.act == APE_ACT_SWIGLU)) return false;
  return true;
}

template <int W> __device__ __forceinline__ void ldrow_f32(const float* src, float (&d)[W]) {
#pragma unroll
  for (int e = 0; e < W / 4; ++e) {
    const float4 t = *reinterpret_cast<const float4*>(src + e * 4);
    d[e * 4] = t.x; d[e * 4 + 1] = t.y; d[e * 4 + 2] = t.z; d[e * 4 + 3] = t.w;
  }
}

// silu(g) * u with the reciprocal instruction (v_rcp_f32, 1 ulp) instead of an IEEE division (a ~10-instruction expansion):
// the SwiGLU epilogue of the ViT up projection was VALU bound on it (ablation: 17 of the launch's 99 us were epilogue arithmetic)
__device__ __forceinline__ float silu_mul_fast(float g, float u) { return g * __builtin_amdgcn_rcpf(1.f + __expf(-g)) * u; }
// (Round 5 tried the two products as single-instruction inline asm, to keep hipcc's SLP vectoriser from pairing them into v_pk_mul_f32
// behind three v_mov per pair: the asm reads v_rcp_f32's result in the very next VALU slot, and the hazard recogniser does not see
// inside inline asm -- gfx950 needs a wait state between a transcendental and its consumer -- so every SwiGLU output was garbage at
// random (caught by the bf16 pipeline pins: p2 0.69 rms instead of 1.0e-2).  No measurable gain either: 88.2 vs 88.0 us.  Reverted.)

// row-invariant vectors of a lane's W columns, loaded ONCE per tile (the unrolled row loop re-read them per accumulator row:
// the stores in between keep the compiler from merging the loads)
template <int W>
struct EpiCols {
  float bias[W];
  float colvec[W];
};
template <int W, bool NORM>
__device__ __forceinline__ void epi_cols_load(const GemmParams& p, int nb, EpiCols<W>& c) {
  if (!NORM) return;              // without the folded LayerNorm the bias rides in the accumulators (BIAS_IN_ACC): nothing to load
  if (p.bias != nullptr && !(p.vec_ok & 16)) ldrow_f32<W>(p.bias + nb, c.bias);
  else {
#pragma unroll
    for (int e = 0; e < W; ++e) c.bias[e] = 0.f;
  }
  ldrow_f32<W>(p.colvec + nb, c.colvec);
}

// cs = W/2 (cos, sin) pairs of the lane's columns when the packed table came through LDS (ROPE_LDS), else unused.
// Without NORM the bias is already inside v: the tile kernel starts its accumulators from it (the bias is added before everything else
// in the epilogue order, and "+ bias" commutes with the accumulation up to fp32 rounding) -- no loads, no live registers.
template <int W, bool ROPE, bool NORM, int ACT, typename H = bf16_t, bool ROPE_LDS = false>
// skip_residual: the caller adds the residual itself, LAST, like this function does (gemm_p8.hip fetches the fp32 residual of its
// 256 x 128 tiles ahead of the epilogue and holds it in registers)
__device__ __forceinline__ void epi_row_fast(const GemmParams& p, int m, int nb, float (&v)[W], const EpiCols<W>& cols, const float* cs = nullptr,
                                             bool skip_residual = false) {
  if (NORM) {
    const float rs = p.rowscale[m], sh = p.rowshift[m];
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] = fmaf(v[e], rs, sh * cols.colvec[e]);
  }
  if (NORM && p.bias != nullptr) {
    if (p.vec_ok & 16) {
      const float b = p.bias[m];
#pragma unroll
      for (int e = 0; e < W; ++e) v[e] += b;
    } else {
#pragma unroll
      for (int e = 0; e < W; ++e) v[e] += cols.bias[e];
    }
  }
  if (ROPE) {
    if (nb < p.rope_cols) {
      if (ROPE_LDS) {
#pragma unroll
        for (int e = 0; e < W; e += 2) {          // pair (2i, 2i+1) shares (cos, sin) = cs[e], cs[e + 1]
          const float x0 = v[e], x1 = v[e + 1], c = cs[e], sn = cs[e + 1];
          v[e] = x0 * c - x1 * sn;
          v[e + 1] = x1 * c + x0 * sn;
        }
      } else {
        const int hd = p.rope_hd;
        const int rmask = p.rope_rows >= p.M ? 0x7fffffff : p.rope_rows - 1;
        const size_t trow = (size_t)(m & rmask) * hd + (nb & (hd - 1));
        float c[W], sn[W];
        ldrow_f32<W>(p.rope_cos + trow, c);
        ldrow_f32<W>(p.rope_sin + trow, sn);
#pragma unroll
        for (int e = 0; e < W; e += 2) {          // rotate_half pairs (2i, 2i+1) -> (-x[2i+1], x[2i])
          const float x0 = v[e], x1 = v[e + 1];
          v[e] = x0 * c[e] - x1 * sn[e];
          v[e + 1] = x1 * c[e + 1] + x0 * sn[e + 1];
        }
      }
    }
  }
  if (ACT == EPI_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (ACT == EPI_ACT_SWIGLU) {                   // interleaved (gate, up) pairs -> W/2 outputs in v[0 .. W/2)
#pragma unroll
    for (int e = 0; e < W / 2; ++e) v[e] = silu_mul_fast(v[2 * e], v[2 * e + 1]);
    return;
  }
  if (!skip_residual && p.residual != nullptr) {
    const size_t roff = (size_t)m * p.ldr + nb;
    if (p.res_dt == APE_DT_F32) {
      float r[W];
      ldrow_f32<W>(reinterpret_cast<const float*>(p.residual) + roff, r);
#pragma unroll
      for (int e = 0; e < W; ++e) v[e] += r[e];
    } else {
      const H* rp = reinterpret_cast<const H*>(p.residual) + roff;
#pragma unroll
      for (int e = 0; e < W / 8; ++e) {
        float r[8];
        ld8<H>(rp + e * 8, r);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[e * 8 + q] += r[q];
      }
    }
  }
}

// store NV values (NV % 4 == 0) starting at column ocol of row m; 16-byte pieces when the address allows
template <int NV, typename H = bf16_t>
__device__ __forceinline__ void store_row(const GemmParams& p, int m, int ocol, const float* o) {
  if (p.out_dt == APE_DT_F32) {
    float* dst = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + ocol;
#pragma unroll
    for (int e = 0; e < NV / 4; ++e) *reinterpret_cast<float4*>(dst + e * 4) = make_float4(o[e * 4], o[e * 4 + 1], o[e * 4 + 2], o[e * 4 + 3]);
  } else {
    H* dst = reinterpret_cast<H*>(p.C) + (size_t)m * p.ldc + ocol;
    if (NV % 8 == 0) {
#pragma unroll
      for (int e = 0; e < NV / 8; ++e)
        *reinterpret_cast<uint4*>(dst + e * 8) = make_uint4(h16<H>::pack2(o[e * 8], o[e * 8 + 1]), h16<H>::pack2(o[e * 8 + 2], o[e * 8 + 3]),
                                                            h16<H>::pack2(o[e * 8 + 4], o[e * 8 + 5]), h16<H>::pack2(o[e * 8 + 6], o[e * 8 + 7]));
    } else {
#pragma unroll
      for (int e = 0; e < NV / 4; ++e)
        *reinterpret_cast<uint2*>(dst + e * 4) = make_uint2(h16<H>::pack2(o[e * 4], o[e * 4 + 1]), h16<H>::pack2(o[e * 4 + 2], o[e * 4 + 3]));
    }
  }
}
