// Multi-scale deformable attention sampler (forward only) for gfx950.
//
// Replaces ms_deformable_im2col_gpu_kernel (ape/layers/csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:237-299,
// bilinear helper :33-84) -- which does not even build for ROCm in the reference (ms_deform_attn.h:32-39)
// -- and the grid_sample path multi_scale_deformable_attn_pytorch (ape/layers/multi_scale_deform_attn.py:84-124).
//
// Thread mapping (M = 8 heads x D = 32 channels, P = 4 points):
//   one 64-lane wave = 2 queries x 8 heads x 4 lanes; each lane owns 8 consecutive channels of one head,
//   so every bilinear corner is ONE 16-byte (bf16) load and a wave touches 4 full 64-byte head rows per
//   corner.  The value tensor (44.7 MB bf16 at 1024^2) is served from L2 / Infinity Cache: query chunks
//   are assigned to XCDs contiguously so one XCD's L2 only sees one band of every level.
// The fused variant also does the softmax over the L*P logits and the sampling-location arithmetic
// (multi_scale_deform_attn.py:278-311) in registers, so locations / weights never touch HBM.
#include <type_traits>

#include "common.h"
#include "../../include/ape_hip.h"

#define MS_HEADS 8
#define MS_D 32
#define MS_P 4

struct MsdaParams {
  const void* value; int ldv;
  const void* loc;      // plain: [Q, M, L, P, 2]
  const void* attn;     // plain: [Q, M, L, P]
  const void* offw; int ldoffw;    // fused: fp32, or f16 with offw_f16 (the offset / logit GEMM writes half the bytes)
  int offw_f16;
  const float* ref; int refdim;    // fused
  void* out; int ldout;
  int S, Q;             // per batch element
  int H[8], W[8], start[8];
  // fused quad kernel: 1 / W, 1 / H when EVERY level's extent is a power of two (then o * (1 / W) == o / W bit for bit and the location
  // arithmetic needs no IEEE division sequence: 10 of them per lane, ~110 VALU instructions); pow2 = 0 keeps the division
  float invW[8], invH[8];
  int pow2;
};

template <typename TV>
__device__ __forceinline__ void corner_acc(const TV* __restrict__ vbase, int ldv, int idx, float wgt, float acc[8]) {
  float v[8];
  ld8<TV>(vbase + (size_t)idx * ldv, v);
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = fmaf(wgt, v[c], acc[c]);
}

// bilinear sample at (x, y) in normalised [0,1] coords of an H x W level (align_corners=False, zero pad),
// weighted by aw and accumulated into acc
template <typename TV>
__device__ __forceinline__ void sample_acc(const TV* __restrict__ vlvl, int ldv, int H, int W, float lx, float ly,
                                           float aw, float acc[8]) {
  const float h_im = ly * (float)H - 0.5f;
  const float w_im = lx * (float)W - 0.5f;
  if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) return;
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = (int)hf, w_low = (int)wf;
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - hf, lw = w_im - wf;
  const float hh = 1.f - lh, hw = 1.f - lw;
  if (h_low >= 0 && w_low >= 0) corner_acc<TV>(vlvl, ldv, h_low * W + w_low, aw * hh * hw, acc);
  if (h_low >= 0 && w_high <= W - 1) corner_acc<TV>(vlvl, ldv, h_low * W + w_high, aw * hh * lw, acc);
  if (h_high <= H - 1 && w_low >= 0) corner_acc<TV>(vlvl, ldv, h_high * W + w_low, aw * lh * hw, acc);
  if (h_high <= H - 1 && w_high <= W - 1) corner_acc<TV>(vlvl, ldv, h_high * W + w_high, aw * lh * lw, acc);
}

template <typename TV, typename TO, int L, bool FUSED>
__global__ __launch_bounds__(256) void msda_kernel(const MsdaParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // XCD-aware chunk order: blocks of one XCD (blockIdx % 8) take a contiguous range of query chunks
  int chunk;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, j = b >> 3;
    chunk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int batch = blockIdx.y;
  const int g = lane >> 2, sub = lane & 3;
  const int qi = chunk * 8 + wave * 2 + (g >> 3);
  const int h = g & 7;
  if (qi >= p.Q) return;
  const size_t qg = (size_t)batch * p.Q + qi;  // global query row
  const TV* vb = reinterpret_cast<const TV*>(p.value) + (size_t)batch * p.S * p.ldv + h * MS_D + sub * 8;

  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;

  constexpr int LP = L * MS_P;
  if (FUSED) {
    // logits + softmax over the L*P samples of this (query, head)
    const float* lg = reinterpret_cast<const float*>(p.offw) + qg * p.ldoffw + MS_HEADS * LP * 2 + h * LP;
    float w[LP];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < LP; s += 4) {
      const float4 t = *reinterpret_cast<const float4*>(lg + s);
      w[s] = t.x; w[s + 1] = t.y; w[s + 2] = t.z; w[s + 3] = t.w;
      mx = fmaxf(mx, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < LP; ++s) { w[s] = expf(w[s] - mx); sum += w[s]; }
    const float inv = 1.f / sum;
    const float* of = reinterpret_cast<const float*>(p.offw) + qg * p.ldoffw + h * LP * 2;
    const float* rf = p.ref + qg * L * p.refdim;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int H = p.H[l], W = p.W[l];
      const TV* vl = vb + (size_t)p.start[l] * p.ldv;
      const float4 o01 = *reinterpret_cast<const float4*>(of + l * 8);
      const float4 o23 = *reinterpret_cast<const float4*>(of + l * 8 + 4);
      const float ox[4] = {o01.x, o01.z, o23.x, o23.z};
      const float oy[4] = {o01.y, o01.w, o23.y, o23.w};
      const float rx = rf[l * p.refdim], ry = rf[l * p.refdim + 1];
      float sx, sy;
      if (p.refdim == 2) {
        // loc = ref + off / (W, H)                       (multi_scale_deform_attn.py:298-303)
        sx = 1.f / (float)W; sy = 1.f / (float)H;
#pragma unroll
        for (int pt = 0; pt < MS_P; ++pt)
          sample_acc<TV>(vl, p.ldv, H, W, rx + ox[pt] / (float)W, ry + oy[pt] / (float)H, w[l * MS_P + pt] * inv, acc);
        (void)sx; (void)sy;
      } else {
        // loc = ref_xy + off / P * ref_wh * 0.5          (multi_scale_deform_attn.py:304-311)
        const float rw = rf[l * p.refdim + 2], rh = rf[l * p.refdim + 3];
#pragma unroll
        for (int pt = 0; pt < MS_P; ++pt)
          sample_acc<TV>(vl, p.ldv, H, W, rx + ox[pt] / (float)MS_P * rw * 0.5f, ry + oy[pt] / (float)MS_P * rh * 0.5f,
                         w[l * MS_P + pt] * inv, acc);
      }
    }
  } else {
    const TV* loc = reinterpret_cast<const TV*>(p.loc) + (qg * MS_HEADS + h) * LP * 2;
    const TV* aw = reinterpret_cast<const TV*>(p.attn) + (qg * MS_HEADS + h) * LP;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int H = p.H[l], W = p.W[l];
      const TV* vl = vb + (size_t)p.start[l] * p.ldv;
#pragma unroll
      for (int pt = 0; pt < MS_P; ++pt) {
        const float lx = ldf<TV>(loc + (l * MS_P + pt) * 2), ly = ldf<TV>(loc + (l * MS_P + pt) * 2 + 1);
        sample_acc<TV>(vl, p.ldv, H, W, lx, ly, ldf<TV>(aw + l * MS_P + pt), acc);
      }
    }
  }
  TO* o = reinterpret_cast<TO*>(p.out) + qg * p.ldout + h * MS_D + sub * 8;
  st8<TO>(o, acc);
}

// ---------------------------------------------------------------------------------------------------------------
// Fused variant with QUAD-SHARED sampling arithmetic.  The 4 lanes of a (query, head) used to repeat the softmax and
// the location / bilinear-corner arithmetic of all L*4 samples (the kernel is VALU bound: ~3000 instructions per wave).
// Here lane `sub` evaluates only the samples of point `sub` (one per level): logit -> softmax weight (max / sum over
// the quad with DPP), location, the 4 corner row indices and the 4 corner weights (attention weight folded in, 0 for
// corners outside the level).  The gather loop then broadcasts (index, weight) of every corner across the quad with
// DPP quad_perm moves -- 2 instructions per corner instead of ~15.  Memory access pattern is unchanged: per corner the
// quad reads one 64-byte head row.
// ---------------------------------------------------------------------------------------------------------------
template <int P> __device__ __forceinline__ float quad_bcast_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), P * 0x55, 0xf, 0xf, true));
}
template <int P> __device__ __forceinline__ int quad_bcast_i(int v) { return __builtin_amdgcn_mov_dpp(v, P * 0x55, 0xf, 0xf, true); }
__device__ __forceinline__ float quad_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)); }
__device__ __forceinline__ float quad_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)); }

// (weight, row index) of corner c of the sample owned by quad lane P, broadcast across the quad
__device__ __forceinline__ float quad_pick_f(int P, float v) {
  return P == 0 ? quad_bcast_f<0>(v) : (P == 1 ? quad_bcast_f<1>(v) : (P == 2 ? quad_bcast_f<2>(v) : quad_bcast_f<3>(v)));
}
__device__ __forceinline__ int quad_pick_i(int P, int v) {
  return P == 0 ? quad_bcast_i<0>(v) : (P == 1 ? quad_bcast_i<1>(v) : (P == 2 ? quad_bcast_i<2>(v) : quad_bcast_i<3>(v)));
}

// The four corner rows of one sample: 4 independent 16-byte loads.  No branch on the weight: a corner outside the level
// carries weight 0 and row index 0 (a valid row) -- a per-corner `if (w != 0)` serialised every load behind a branch.
template <typename TV>
struct CornerGroup {
  uint4 raw[4];      // bf16: 8 channels per corner
  float w[4];
};

// Round 6: a corner travels as a 32-bit BYTE OFFSET of its value row (computed once by the lane that owns the sample: row * row bytes,
// one full-rate v_mul_u32_u24), the receiving lane adds its own (head, channel group) byte offset -- the DPP broadcast and that add are
// ONE v_add_u32_dpp -- and the load takes the wave-uniform base pointer from SGPRs (global_load_dwordx4 v, v_off, s[base:base+1]).
// Before: v_mov_b32_dpp + v_mad_i64_i32 (quarter rate) + v_lshl_add_u64 per corner, 80 corners per lane -- the 64-bit address arithmetic
// was ~30 % of the kernel's VALU cycles, and the kernel is VALU bound.
template <typename TV, int L>
__device__ __forceinline__ void group_load(const unsigned char* __restrict__ vbase, uint32_t lane_off, const float (&cw)[L][4],
                                           const int (&ci)[L][4], int s, CornerGroup<TV>& g) {
  const int l = s >> 2, P = s & 3;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    g.w[c] = quad_pick_f(P, cw[l][c]);
    uint32_t off = (uint32_t)quad_pick_i(P, ci[l][c]) + lane_off;
    asm volatile("" : "+v"(off));            // the load may not be hoisted above this point (IR passes ignore sched_barrier)
    g.raw[c] = *reinterpret_cast<const uint4*>(vbase + (size_t)off);
  }
}

// IEEE-half values: v_fma_mix_f32 multiplies one half of a dword (op_sel picks it) by an fp32 weight into an fp32 accumulator --
// ONE 2-cycle instruction per channel and no unpacking.  The kernel is VALU bound (rocprofv3, profiles/r03_msda_counters_before.txt:
// 1779 VALU instructions per wave at one quad-cycle each = 77 % of the SIMDs' time; L2 hit rate 87 %): a bf16 corner costs 8
// shift / and unpack instructions + 4 v_pk_fma_f32 (4 cycles each), a half corner 8 v_fma_mix_f32.  hipcc does not form the
// instruction from (float)h * w + acc (it converts, then multiplies), hence the asm; the operands are plain VGPR values, so the
// compiler's own waitcnt bookkeeping for the loads that produced them still applies.
__device__ __forceinline__ void fma_mix_lo(float& acc, uint32_t d, float w) {
  asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(d), "v"(w));
}
__device__ __forceinline__ void fma_mix_hi(float& acc, uint32_t d, float w) {
  asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(d), "v"(w));
}
template <typename TV>
__device__ __forceinline__ void group_fma_half(const CornerGroup<TV>& g, float (&acc)[8]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t d[4] = {g.raw[c].x, g.raw[c].y, g.raw[c].z, g.raw[c].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      fma_mix_lo(acc[2 * e], d[e], g.w[c]);
      fma_mix_hi(acc[2 * e + 1], d[e], g.w[c]);
    }
  }
}

// acc += sum_c w[c] * row[c]: the kernel is VALU bound (each lane: 80 corners x 8 channels), so one v_pk_fma_f32 per bf16 pair
template <typename TV>
__device__ __forceinline__ void group_fma(const CornerGroup<TV>& g, f32x2_t (&a2)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x2_t ww = {g.w[c], g.w[c]};
    const uint32_t d[4] = {g.raw[c].x, g.raw[c].y, g.raw[c].z, g.raw[c].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x2_t v2 = {__uint_as_float(d[e] << 16), __uint_as_float(d[e] & 0xffff0000u)};
      a2[e] = __builtin_elementwise_fma(ww, v2, a2[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(a2[e]));   // ... and the FMAs may not sink below it
}

template <typename TV, typename TO, int L, typename TW = float>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void msda_fused_quad_kernel(const MsdaParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int chunk;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, j = b >> 3;
    chunk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int batch = blockIdx.y;
  const int g = lane >> 2, sub = lane & 3;
  int qi = chunk * 8 + wave * 2 + (g >> 3);
  const bool live = qi < p.Q;
  qi = live ? qi : p.Q - 1;               // keep whole quads active for the DPP exchanges; the store is predicated
  const int h = g & 7;
  const size_t qg = (size_t)batch * p.Q + qi;
  const TV* vb = reinterpret_cast<const TV*>(p.value) + (size_t)batch * p.S * p.ldv + h * MS_D + sub * 8;
  // the same rows as (wave-uniform base, 32-bit byte offset): the launcher checks S * ldv * sizeof(TV) < 2^32 and row bytes < 2^24
  const unsigned char* vbase = reinterpret_cast<const unsigned char*>(reinterpret_cast<const TV*>(p.value) + (size_t)batch * p.S * p.ldv);
  const uint32_t lane_off = (uint32_t)((h * MS_D + sub * 8) * sizeof(TV));
  const uint32_t rowb = (uint32_t)(p.ldv * (int)sizeof(TV));
  constexpr int LP = L * MS_P;

  // ---- phase 1: this lane's samples (point `sub` of every level)
  const TW* lg = reinterpret_cast<const TW*>(p.offw) + qg * p.ldoffw + MS_HEADS * LP * 2 + h * LP;
  float wv[L];
  float mx = -INFINITY;
#pragma unroll
  for (int l = 0; l < L; ++l) { wv[l] = ldf<TW>(lg + l * MS_P + sub); mx = fmaxf(mx, wv[l]); }
  mx = fmaxf(mx, quad_xor1(mx));
  mx = fmaxf(mx, quad_xor2(mx));
  float sum = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) { wv[l] = expf(wv[l] - mx); sum += wv[l]; }
  sum += quad_xor1(sum);
  sum += quad_xor2(sum);
  const float inv = 1.f / sum;
  const TW* of = reinterpret_cast<const TW*>(p.offw) + qg * p.ldoffw + h * LP * 2 + sub * 2;
  const float* rf = p.ref + qg * L * p.refdim;
  float cw[L][4];
  int ci[L][4];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = p.H[l], W = p.W[l];
    float2 o;
    if (sizeof(TW) == 4) {
      o = *reinterpret_cast<const float2*>(of + l * 8);
    } else {
      typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
      const h2_t t = *reinterpret_cast<const h2_t*>(of + l * 8);
      o = make_float2((float)t[0], (float)t[1]);
    }
    const float rx = rf[l * p.refdim], ry = rf[l * p.refdim + 1];
    float lx, ly;
    if (p.refdim == 2) {                       // loc = ref + off / (W, H)            (multi_scale_deform_attn.py:298-303)
      if (p.pow2) { lx = rx + o.x * p.invW[l]; ly = ry + o.y * p.invH[l]; }       // exact for power-of-two extents (uniform branch)
      else { lx = rx + o.x / (float)W; ly = ry + o.y / (float)H; }
    } else {                                   // loc = ref_xy + off / P * ref_wh / 2 (multi_scale_deform_attn.py:304-311)
      const float rw = rf[l * p.refdim + 2], rh = rf[l * p.refdim + 3];
      lx = rx + o.x / (float)MS_P * rw * 0.5f; ly = ry + o.y / (float)MS_P * rh * 0.5f;
    }
    const float aw = wv[l] * inv;
    const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
    const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h_low = (int)hf, w_low = (int)wf;
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - hf, lw = w_im - wf;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const bool t_ok = inside && h_low >= 0, b_ok = inside && h_high <= H - 1;
    const bool l_ok = w_low >= 0, r_ok = w_high <= W - 1;
    const int base = p.start[l];
    cw[l][0] = (t_ok && l_ok) ? aw * hh * hw : 0.f; ci[l][0] = (t_ok && l_ok) ? base + h_low * W + w_low : 0;
    cw[l][1] = (t_ok && r_ok) ? aw * hh * lw : 0.f; ci[l][1] = (t_ok && r_ok) ? base + h_low * W + w_high : 0;
    cw[l][2] = (b_ok && l_ok) ? aw * lh * hw : 0.f; ci[l][2] = (b_ok && l_ok) ? base + h_high * W + w_low : 0;
    cw[l][3] = (b_ok && r_ok) ? aw * lh * lw : 0.f; ci[l][3] = (b_ok && r_ok) ? base + h_high * W + w_high : 0;
    if (sizeof(TV) == 2) {                     // the 16-bit gather loops take BYTE offsets of the rows (group_load)
#pragma unroll
      for (int c = 0; c < 4; ++c) ci[l][c] = (int)__umul24((unsigned)ci[l][c], rowb);
    }
  }
  // ---- phase 2: gather; (index, weight) of each corner broadcast from the lane that owns the sample's point.
  // Software pipeline over the L*4 samples: the 4 loads of sample s+1 are issued before the FMAs of sample s; the
  // sched_barriers pin that order (left alone the compiler hoists all 80 loads and spills).
  float acc[8];
  if (std::is_same<TV, f16_t>::value) {
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    CornerGroup<TV> g0, g1;
    group_load<TV, L>(vbase, lane_off, cw, ci, 0, g0);
#pragma unroll
    for (int s = 0; s < L * 4; s += 2) {
      group_load<TV, L>(vbase, lane_off, cw, ci, s + 1, g1);
      __builtin_amdgcn_sched_barrier(0);
      group_fma_half<TV>(g0, acc);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < L * 4) group_load<TV, L>(vbase, lane_off, cw, ci, s + 2, g0);
      __builtin_amdgcn_sched_barrier(0);
      group_fma_half<TV>(g1, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (sizeof(TV) == 2) {
    f32x2_t a2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) a2[e] = (f32x2_t){0.f, 0.f};
    CornerGroup<TV> g0, g1;
    group_load<TV, L>(vbase, lane_off, cw, ci, 0, g0);
#pragma unroll
    for (int s = 0; s < L * 4; s += 2) {
      group_load<TV, L>(vbase, lane_off, cw, ci, s + 1, g1);
      __builtin_amdgcn_sched_barrier(0);
      group_fma<TV>(g0, a2);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < L * 4) group_load<TV, L>(vbase, lane_off, cw, ci, s + 2, g0);
      __builtin_amdgcn_sched_barrier(0);
      group_fma<TV>(g1, a2);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] = a2[e][0]; acc[2 * e + 1] = a2[e][1]; }
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
    for (int s = 0; s < L * 4; ++s) {
      const int l = s >> 2, P = s & 3;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float w = quad_pick_f(P, cw[l][c]);
        const int idx = quad_pick_i(P, ci[l][c]);
        if (w != 0.f) corner_acc<TV>(vb, p.ldv, idx, w, acc);    // fp32 validation path: one corner at a time (no spills)
      }
    }
  }
  if (live) {
    TO* o = reinterpret_cast<TO*>(p.out) + qg * p.ldout + h * MS_D + sub * 8;
    st8<TO>(o, acc);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-STAGED sampler for the encoder's level-0 queries (round 6; BASELINE.json north_star: "LDS-staged 4-level feature sampling with
// coalesced HBM reads"; replaces ape/layers/csrc/MsdaDeformAttn/ms_deform_im2col_cuda.cuh:237-299 for those queries).
//
// The quad kernel above gathers every bilinear corner as one 64-byte head row through the texture-address path: 80 scattered row
// reads per (query, head), 3.6 GB per encoder layer through 64 B / clk / CU with a working set (~670 KB per 16 x 16-query tile) that
// the 32 KB L1 cannot hold -- 189 us per layer, unchanged by a 25 % cut of its VALU instructions (it waits on L1 misses).
// An encoder query samples around ITS OWN position (multi_scale_deform_attn.py:298-303: loc = reference + offset / (W, H), offsets a
// few pixels), so the 256 queries of a 16 x 16 tile of the finest level read, in level l, a window of (16 / 2^l + 2 HALO + 2)^2 rows
// of ONE head: <= 900 rows x 64 B.  One workgroup = (tile, head): per level it STAGES that window with coalesced 16-byte loads
// (each row read once instead of ~9 times), then every thread samples its query's 4 points of the level out of LDS
// (ds_read_b128, chunk slots XOR-swizzled by the window column so that the lanes of one read spread over the banks) and keeps all
// 32 channels of its (query, head) in registers.  Two workgroups per CU (57.6 KB each): one stages while the other samples.
// A corner outside the staged window (an offset beyond HALO pixels, or reference points that are not the queries' own positions)
// is gathered from global memory like before -- the result never depends on the locality assumption, only the speed does.
// Queries of the coarser levels (25 %) go through the quad kernel (their windows on the finest level would not fit).
// ---------------------------------------------------------------------------------------------------------------
constexpr int ML_T = 16;                              // tile edge: 16 x 16 queries of level 0
constexpr int ML_HALO = 6;                            // staged margin around the tile's own positions, in pixels of each level
constexpr int ML_RMAX = ML_T + 2 * ML_HALO + 2;       // 30: widest window (level 0)
constexpr int ML_ROWS = ML_RMAX * ML_RMAX;            // 900 rows x 64 B = 57 600 B of LDS

template <typename TO, typename TW, int L>
__global__ __launch_bounds__(256) void msda_lds_kernel(const MsdaParams p) {
  __shared__ uint4 sV[ML_ROWS * 4];
  const int t = threadIdx.x;
  const int h = blockIdx.x & 7;                       // head = XCD (blockIdx round-robins over the 8 XCDs): an XCD's L2 holds one head's slices
  const int tile = blockIdx.x >> 3;
  const int batch = blockIdx.y;
  const int W0 = p.W[0];
  const int tiles_x = W0 / ML_T;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int qi = p.start[0] + (ty * ML_T + (t >> 4)) * W0 + tx * ML_T + (t & 15);
  const size_t qg = (size_t)batch * p.Q + qi;
  const size_t q_first = (size_t)batch * p.Q + p.start[0] + (size_t)(ty * ML_T) * W0 + tx * ML_T;
  const size_t q_last = q_first + (size_t)(ML_T - 1) * W0 + (ML_T - 1);
  const uint32_t rowb = (uint32_t)p.ldv * 2u;
  const unsigned char* vhead = reinterpret_cast<const unsigned char*>(p.value) + (size_t)batch * p.S * rowb + h * (MS_D * 2);
  constexpr int LP = L * MS_P;

  // ---- softmax over the L * 4 logits of this (query, head)
  float e[LP];
  {
    const TW* lg = reinterpret_cast<const TW*>(p.offw) + qg * p.ldoffw + MS_HEADS * LP * 2 + h * LP;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < LP; ++i) { e[i] = ldf<TW>(lg + i); mx = fmaxf(mx, e[i]); }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { e[i] = expf(e[i] - mx); sum += e[i]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < LP; ++i) e[i] *= inv;
  }
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;
  const TW* of = reinterpret_cast<const TW*>(p.offw) + qg * p.ldoffw + h * LP * 2;
  const float* rf = p.ref + qg * L * 2;
  const float* rfa = p.ref + q_first * L * 2;
  const float* rfb = p.ref + q_last * L * 2;

#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int H = p.H[l], W = p.W[l];
    // ---- the window of this level: the tile's own positions +- HALO (uniform: from the tile's first and last query)
    const float ax = rfa[l * 2] * (float)W - 0.5f, ay = rfa[l * 2 + 1] * (float)H - 0.5f;
    const float bx = rfb[l * 2] * (float)W - 0.5f, by = rfb[l * 2 + 1] * (float)H - 0.5f;
    int x_lo = max(0, (int)floorf(fminf(ax, bx)) - ML_HALO), x_hi = min(W - 1, (int)floorf(fmaxf(ax, bx)) + ML_HALO + 1);
    int y_lo = max(0, (int)floorf(fminf(ay, by)) - ML_HALO), y_hi = min(H - 1, (int)floorf(fmaxf(ay, by)) + ML_HALO + 1);
    x_lo = __builtin_amdgcn_readfirstlane(x_lo); x_hi = __builtin_amdgcn_readfirstlane(x_hi);
    y_lo = __builtin_amdgcn_readfirstlane(y_lo); y_hi = __builtin_amdgcn_readfirstlane(y_hi);
    const int rw = max(0, min(ML_RMAX, x_hi - x_lo + 1)), rh = max(0, min(ML_RMAX, y_hi - y_lo + 1));
    const unsigned char* vl = vhead + (size_t)p.start[l] * rowb;
    if (l > 0) __syncthreads();                        // every thread is done with the previous level's window
    // ---- stage: item i = (row r = i >> 2, chunk slot s = i & 3); slot s of window row (ry, rx) holds the row's chunk s ^ (rx & 3)
    {
      const int n = rw * rh * 4;
      if (n == 0 && t < 4) sV[t] = make_uint4(0u, 0u, 0u, 0u);     // empty window: corners read row 0 with weight 0 -- it must hold finite values
      const float inv_rw = 1.f / (float)max(rw, 1);
      for (int i0 = 0; i0 < n; i0 += 1024) {
        uint4 v[4];
        int ii[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * 256 + t;
          ii[u] = i;
          if (i < n) {
            const int r = i >> 2, sl = i & 3;
            const int ry = (int)(((float)r + 0.5f) * inv_rw);
            const int rx = r - ry * rw;
            v[u] = *reinterpret_cast<const uint4*>(vl + (size_t)((y_lo + ry) * W + x_lo + rx) * rowb + ((sl ^ (rx & 3)) << 4));
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ii[u] < n) sV[ii[u]] = v[u];
      }
    }
    __syncthreads();
    // ---- this query's 4 points of the level
    float off[8];
    if (sizeof(TW) == 4) {
      const float4 o0 = *reinterpret_cast<const float4*>(of + l * 8), o1 = *reinterpret_cast<const float4*>(of + l * 8 + 4);
      off[0] = o0.x; off[1] = o0.y; off[2] = o0.z; off[3] = o0.w; off[4] = o1.x; off[5] = o1.y; off[6] = o1.z; off[7] = o1.w;
    } else {
      typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
      const h8_t tq = *reinterpret_cast<const h8_t*>(of + l * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u) off[u] = (float)tq[u];
    }
    const float rx0 = rf[l * 2], ry0 = rf[l * 2 + 1];
    // this level's 4 attention weights: static indices only (a runtime index would move e[] out of the register file)
    float ew[MS_P];
#pragma unroll
    for (int pt = 0; pt < MS_P; ++pt) {
      ew[pt] = e[pt];
#pragma unroll
      for (int ll = 1; ll < L; ++ll) ew[pt] = (l == ll) ? e[ll * MS_P + pt] : ew[pt];
    }
#pragma unroll
    for (int pt = 0; pt < MS_P; ++pt) {
      float lx, ly;
      if (p.pow2) { lx = rx0 + off[2 * pt] * p.invW[l]; ly = ry0 + off[2 * pt + 1] * p.invH[l]; }
      else { lx = rx0 + off[2 * pt] / (float)W; ly = ry0 + off[2 * pt + 1] / (float)H; }
      const float aw = ew[pt];
      const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
      const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = (int)hf, w_low = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf;
      const float hh = 1.f - lh, hw = 1.f - lw;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cy = h_low + (c >> 1), cx = w_low + (c & 1);
        const bool ok = inside && cy >= 0 && cy <= H - 1 && cx >= 0 && cx <= W - 1;
        const float wgt = ok ? aw * ((c >> 1) ? lh : hh) * ((c & 1) ? lw : hw) : 0.f;
        const int ry = cy - y_lo, rx = cx - x_lo;
        const bool staged = ok && ry >= 0 && ry < rh && rx >= 0 && rx < rw;
        const int row = staged ? ry * rw + rx : 0;
        const float wl = staged ? wgt : 0.f;
        const int sw = rx & 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                      // logical chunk k (channels 8 k .. 8 k + 7) sits in slot k ^ (rx & 3)
          const uint4 d = sV[row * 4 + (k ^ sw)];
          fma_mix_lo(acc[8 * k + 0], d.x, wl); fma_mix_hi(acc[8 * k + 1], d.x, wl);
          fma_mix_lo(acc[8 * k + 2], d.y, wl); fma_mix_hi(acc[8 * k + 3], d.y, wl);
          fma_mix_lo(acc[8 * k + 4], d.z, wl); fma_mix_hi(acc[8 * k + 5], d.z, wl);
          fma_mix_lo(acc[8 * k + 6], d.w, wl); fma_mix_hi(acc[8 * k + 7], d.w, wl);
        }
        if (ok && !staged) {                               // outside the window: the row comes from global memory (rare)
          const unsigned char* g = vl + (size_t)(cy * W + cx) * rowb;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint4 d = *reinterpret_cast<const uint4*>(g + (k << 4));
            fma_mix_lo(acc[8 * k + 0], d.x, wgt); fma_mix_hi(acc[8 * k + 1], d.x, wgt);
            fma_mix_lo(acc[8 * k + 2], d.y, wgt); fma_mix_hi(acc[8 * k + 3], d.y, wgt);
            fma_mix_lo(acc[8 * k + 4], d.z, wgt); fma_mix_hi(acc[8 * k + 5], d.z, wgt);
            fma_mix_lo(acc[8 * k + 6], d.w, wgt); fma_mix_hi(acc[8 * k + 7], d.w, wgt);
          }
        }
      }
    }
  }
  TO* o = reinterpret_cast<TO*>(p.out) + qg * p.ldout + h * MS_D;
#pragma unroll
  for (int k = 0; k < 4; ++k) st8<TO>(o + 8 * k, acc + 8 * k);
}

template <typename TV, typename TO, bool FUSED>
static int launch_msda(const MsdaParams& p, int B, int L, hipStream_t s) {
  const dim3 grid(ceil_div(p.Q, 8), B), block(256);
  if constexpr (FUSED) {
    if (p.offw_f16) {
      switch (L) {
        case 1: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 1, f16_t>), grid, block, 0, s, p); return 0;
        case 2: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 2, f16_t>), grid, block, 0, s, p); return 0;
        case 3: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 3, f16_t>), grid, block, 0, s, p); return 0;
        case 4: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 4, f16_t>), grid, block, 0, s, p); return 0;
        case 5: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 5, f16_t>), grid, block, 0, s, p); return 0;
        default: ape_set_error("msda: num_levels %d not in 1..5", L); return -1;
      }
    }
    switch (L) {
      case 1: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 1>), grid, block, 0, s, p); return 0;
      case 2: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 2>), grid, block, 0, s, p); return 0;
      case 3: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 3>), grid, block, 0, s, p); return 0;
      case 4: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 4>), grid, block, 0, s, p); return 0;
      case 5: APE_LAUNCH((msda_fused_quad_kernel<TV, TO, 5>), grid, block, 0, s, p); return 0;
      default: ape_set_error("msda: num_levels %d not in 1..5", L); return -1;
    }
  } else {
  switch (L) {
    case 1: APE_LAUNCH((msda_kernel<TV, TO, 1, FUSED>), grid, block, 0, s, p); break;
    case 2: APE_LAUNCH((msda_kernel<TV, TO, 2, FUSED>), grid, block, 0, s, p); break;
    case 3: APE_LAUNCH((msda_kernel<TV, TO, 3, FUSED>), grid, block, 0, s, p); break;
    case 4: APE_LAUNCH((msda_kernel<TV, TO, 4, FUSED>), grid, block, 0, s, p); break;
    case 5: APE_LAUNCH((msda_kernel<TV, TO, 5, FUSED>), grid, block, 0, s, p); break;
    default: ape_set_error("msda: num_levels %d not in 1..5", L); return -1;
  }
  }
  return 0;
}

// spatial_shapes / level_start_index are DEVICE int64 tensors in the reference operator; they are tiny and
// constant per resolution, so the C-ABI takes HOST pointers (the Python shim keeps CPU copies) and bakes
// them into the kernel argument block -- no dependent device loads in the inner loop.
static int fill_levels(MsdaParams& p, const int64_t* shapes, const int64_t* starts, int L, int S) {
  if (L < 1 || L > 5) { ape_set_error("msda: num_levels %d not in 1..5", L); return -1; }
  int64_t tot = 0;
  for (int l = 0; l < L; ++l) {
    p.H[l] = (int)shapes[2 * l]; p.W[l] = (int)shapes[2 * l + 1]; p.start[l] = (int)starts[l];
    if (p.H[l] <= 0 || p.W[l] <= 0 || starts[l] != tot) { ape_set_error("msda: inconsistent level %d (h=%d w=%d start=%lld)", l, p.H[l], p.W[l], (long long)starts[l]); return -1; }
    tot += (int64_t)p.H[l] * p.W[l];
  }
  if (tot != S) { ape_set_error("msda: sum(h*w)=%lld != num_value=%d", (long long)tot, S); return -1; }
  return 0;
}

extern "C" int ape_hip_ms_deform_attn_forward(const void* value, int ldv, const int64_t* spatial_shapes,
                                              const int64_t* level_start_index, const void* sampling_loc,
                                              const void* attn_weight, void* out, int ldout, int B, int S, int Q, int L,
                                              int dt, void* stream) {
  APE_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out, "ms_deform_attn_forward: null pointer");
  APE_CHECK_ARG(B > 0 && S > 0 && Q > 0, "ms_deform_attn_forward: bad sizes");
  APE_CHECK_ARG(ldv % 8 == 0 && ldout % 8 == 0 && ((uintptr_t)value) % 16 == 0 && ((uintptr_t)out) % 16 == 0,
                "ms_deform_attn_forward: value/out must be 16-byte aligned with ld %% 8 == 0");
  MsdaParams p;
  memset(&p, 0, sizeof(p));
  p.value = value; p.ldv = ldv; p.loc = sampling_loc; p.attn = attn_weight; p.out = out; p.ldout = ldout; p.S = S; p.Q = Q;
  if (fill_levels(p, spatial_shapes, level_start_index, L, S)) return -1;
  int rc;
  APE_CHECK_ARG(dt == APE_DT_F32 || dt == APE_DT_BF16 || dt == APE_DT_F16, "ms_deform_attn_forward: dtype code %d (0 = f32, 1 = bf16, 2 = f16)", dt);
  if (dt == APE_DT_BF16) rc = launch_msda<bf16_t, bf16_t, false>(p, B, L, (hipStream_t)stream);
  else if (dt == APE_DT_F16) rc = launch_msda<f16_t, f16_t, false>(p, B, L, (hipStream_t)stream);   // the CUDA op's half dispatch (ms_deform_attn_cuda.cu:65)
  else rc = launch_msda<float, float, false>(p, B, L, (hipStream_t)stream);
  if (rc) return rc;
  APE_CHECK_LAUNCH("ape_hip_ms_deform_attn_forward");
  return 0;
}

static int msda_fused_launch(const void* value, int ldv, int v_dt, const int64_t* spatial_shapes, const int64_t* level_start_index,
                             const void* offw, int ldoffw, int offw_f16, const float* ref, int refdim, void* out, int ldout, int out_dt,
                             int B, int S, int Q, int L, void* stream) {
  APE_CHECK_ARG(value && spatial_shapes && level_start_index && offw && ref && out, "msda_fused: null pointer");
  APE_CHECK_ARG(refdim == 2 || refdim == 4, "msda_fused: reference points must have 2 or 4 coordinates, got %d", refdim);
  APE_CHECK_ARG(B > 0 && S > 0 && Q > 0, "msda_fused: bad sizes");
  APE_CHECK_ARG(ldv % 8 == 0 && ldout % 8 == 0 && ldoffw % 4 == 0 && ((uintptr_t)value) % 16 == 0 &&
                    ((uintptr_t)out) % 16 == 0 && ((uintptr_t)offw) % 16 == 0,
                "msda_fused: alignment (value/out 16 B, ld %% 8; offw 16 B, ld %% 4)");
  MsdaParams p;
  memset(&p, 0, sizeof(p));
  p.value = value; p.ldv = ldv; p.offw = offw; p.ldoffw = ldoffw; p.offw_f16 = offw_f16; p.ref = ref; p.refdim = refdim;
  p.out = out; p.ldout = ldout; p.S = S; p.Q = Q;
  if (fill_levels(p, spatial_shapes, level_start_index, L, S)) return -1;
  // 16-bit gather loops address a corner as (uniform base of the batch element) + 32-bit byte offset, the row offset a 24 x 24-bit product
  APE_CHECK_ARG(v_dt == APE_DT_F32 || ((uint64_t)S * (uint64_t)ldv * 2 < (1ull << 32) && S < (1 << 24) && ldv * 2 < (1 << 24)),
                "msda_fused: a batch element's value tensor must stay below 4 GiB (S = %d rows of %d elements)", S, ldv);
  p.pow2 = 1;
  for (int l = 0; l < L; ++l) {
    p.pow2 &= ((p.W[l] & (p.W[l] - 1)) == 0 && (p.H[l] & (p.H[l] - 1)) == 0) ? 1 : 0;
    p.invW[l] = 1.f / (float)p.W[l]; p.invH[l] = 1.f / (float)p.H[l];
  }
  int rc;
  hipStream_t s = (hipStream_t)stream;
  // ---- encoder calls (the queries ARE the tokens, 2-d reference points), OPT-IN (APE_MSDA_LDS=1): level-0 queries through the LDS-staged
  // kernel, the coarser levels' queries through the quad kernel on the remaining query range.  Measured on MI355X at 1024^2
  // (profiles/r06_msda_lds_staged.txt): 191 us for the level-0 queries + 58 us for the rest = 249 us against 189 us for the quad kernel
  // alone -- correct in every case of tests/test_ops_gpu.py::test_msda_lds_staged_encoder_path, but SLOWER: with one thread per (query,
  // head) the location / corner arithmetic of all 20 samples is no longer shared by a quad (8.1 k VALU instructions per (query, head)
  // against 4.4 k), and 57.6 KB of LDS per workgroup leaves 2 waves per SIMD to hide LDS latency with.  So the default stays the quad
  // kernel; the staged kernel is kept as the measured answer to "why not LDS".
  {
    const char* le = getenv("APE_MSDA_LDS");
    const int n0 = p.H[0] * p.W[0];
    if ((le != nullptr && atoi(le) == 1) && L == 5 && refdim == 2 && Q == S && v_dt == APE_DT_F16 && p.start[0] == 0 && p.W[0] % ML_T == 0 &&
        p.H[0] % ML_T == 0 && n0 < Q && (out_dt == APE_DT_BF16 || out_dt == APE_DT_F16)) {
      const dim3 grid((p.H[0] / ML_T) * (p.W[0] / ML_T) * 8, B), block(256);
      if (out_dt == APE_DT_BF16) {
        if (offw_f16) APE_LAUNCH((msda_lds_kernel<bf16_t, f16_t, 5>), grid, block, 0, s, p);
        else APE_LAUNCH((msda_lds_kernel<bf16_t, float, 5>), grid, block, 0, s, p);
      } else {
        if (offw_f16) APE_LAUNCH((msda_lds_kernel<f16_t, f16_t, 5>), grid, block, 0, s, p);
        else APE_LAUNCH((msda_lds_kernel<f16_t, float, 5>), grid, block, 0, s, p);
      }
      APE_CHECK_LAUNCH("ape_hip_msda_fused (LDS-staged level-0 queries)");
      // the rest of the queries: [n0, Q) of every batch element -- the quad kernel indexes queries from its pointers' origin
      APE_CHECK_ARG(B == 1, "msda_fused: the LDS-staged encoder path takes one batch element per call");
      MsdaParams r = p;
      r.Q = Q - n0;
      r.offw = reinterpret_cast<const unsigned char*>(offw) + (size_t)n0 * ldoffw * (offw_f16 ? 2 : 4);
      r.ref = ref + (size_t)n0 * L * refdim;
      r.out = reinterpret_cast<unsigned char*>(out) + (size_t)n0 * ldout * 2;
      if (out_dt == APE_DT_BF16) rc = launch_msda<f16_t, bf16_t, true>(r, B, L, s);
      else rc = launch_msda<f16_t, f16_t, true>(r, B, L, s);
      if (rc) return rc;
      APE_CHECK_LAUNCH("ape_hip_msda_fused");
      return 0;
    }
  }
  if (v_dt == APE_DT_BF16 && out_dt == APE_DT_BF16) rc = launch_msda<bf16_t, bf16_t, true>(p, B, L, s);
  else if (v_dt == APE_DT_BF16 && out_dt == APE_DT_F32) rc = launch_msda<bf16_t, float, true>(p, B, L, s);
  else if (v_dt == APE_DT_F16 && out_dt == APE_DT_BF16) rc = launch_msda<f16_t, bf16_t, true>(p, B, L, s);    // half values (production)
  else if (v_dt == APE_DT_F16 && out_dt == APE_DT_F32) rc = launch_msda<f16_t, float, true>(p, B, L, s);
  else if (v_dt == APE_DT_F16 && out_dt == APE_DT_F16) rc = launch_msda<f16_t, f16_t, true>(p, B, L, s);      // the f16 flavour of the pipeline
  else if (v_dt == APE_DT_F32 && out_dt == APE_DT_F32) rc = launch_msda<float, float, true>(p, B, L, s);
  else { ape_set_error("msda_fused: unsupported dtype combination v=%d out=%d", v_dt, out_dt); return -1; }
  if (rc) return rc;
  APE_CHECK_LAUNCH("ape_hip_msda_fused");
  return 0;
}

extern "C" int ape_hip_msda_fused(const void* value, int ldv, int v_dt, const int64_t* spatial_shapes,
                                  const int64_t* level_start_index, const float* offw, int ldoffw, const float* ref,
                                  int refdim, void* out, int ldout, int out_dt, int B, int S, int Q, int L, void* stream) {
  return msda_fused_launch(value, ldv, v_dt, spatial_shapes, level_start_index, offw, ldoffw, 0, ref, refdim, out, ldout, out_dt, B, S, Q, L,
                           stream);
}

// offsets | logits stored as IEEE half (the K = 256 GEMM that produces them is bound by the bytes it writes)
extern "C" int ape_hip_msda_fused_h(const void* value, int ldv, int v_dt, const int64_t* spatial_shapes,
                                    const int64_t* level_start_index, const void* offw_f16, int ldoffw, const float* ref,
                                    int refdim, void* out, int ldout, int out_dt, int B, int S, int Q, int L, void* stream) {
  APE_CHECK_ARG(v_dt == APE_DT_BF16 || v_dt == APE_DT_F16, "msda_fused_h: bf16 or f16 values (the production modes)");
  return msda_fused_launch(value, ldv, v_dt, spatial_shapes, level_start_index, offw_f16, ldoffw, 1, ref, refdim, out, ldout, out_dt, B, S, Q,
                           L, stream);
}
