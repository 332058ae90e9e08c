// Flash-style scaled-dot-product attention for gfx950 (no dropout; optional causal mask for the CLIP text tower,
// ape/modeling/text/eva02_clip/transformer.py:714-720 -- key tiles above the diagonal of a workgroup are never loaded).
//
// Replaces F.scaled_dot_product_attention at ape/modeling/backbone/vit_eva_clip.py:261-263 (16 heads x 64;
// 4 windows x 1024 tokens or 1 x 4096 tokens) and the attention core of nn.MultiheadAttention used by the
// decoder self-attention (detrex MultiheadAttention built at deformable_transformer_vl.py:141-146; 8 x 32).
//
// bf16 kernel: one workgroup = 4 waves = 64 query rows of one (window, head); K and V^T tiles of 64 keys are
// register-staged into XOR-swizzled LDS (double buffered, one barrier per tile).  The score MFMA is issued
// "swapped" (S^T = K.Q^T) so that after it every lane holds, for ONE query (column lane&15), 4 consecutive
// keys per 16-key tile; those registers are exactly the B operand of the P.V MFMA (O^T = V^T.P^T) under a
// permuted k-order that is applied identically to the V^T operand -> no LDS round trip and no cross-lane
// traffic for P.  Row max / row sum need only two xor-shuffles (lanes l, l^16, l^32, l^48 share a query).
// V is consumed TRANSPOSED ([head*HD + d][token]); the producing GEMM writes it that way (trans_out).
// f32 kernel: one query per lane, K / V tiles broadcast from LDS -- exact-math validation mode.
#include <stdlib.h>

#include "common.h"
#include "../../include/ape_hip.h"

struct AttnParams {
  const void* Q; const void* K; const void* Vt; void* O;
  int ldq, ldk, ldvt, ldo;
  int N, H;
  int bstride;       // rows (tokens) between consecutive batch items / windows, >= N
  float scale_log2;  // scale * log2(e)
  float scale;
  int causal;        // f32 kernel: keys > query are masked (the bf16 kernel takes it as a template argument)
};

__device__ __forceinline__ int swz_rows(int row, int c, int chunks_per_row) {
  // element offset of 16-byte chunk c of `row`; rows are chunks_per_row*8 bf16 wide
  if (chunks_per_row == 8) return row * 64 + ((c ^ ((row >> 1) & 7)) << 3);
  /* 4 chunks (64-byte rows) */ return row * 32 + ((c ^ (((row >> 3) & 1) << 1)) << 3);
}

// ---- key order of the score tiles (PERM kernels).  Tile i, MFMA row rho of S^T = K.Q^T is key
//      kappa(i, rho) = 32 (i >> 1) + 8 (rho >> 2) + 4 (i & 1) + (rho & 3)
// instead of 16 i + rho: a lane (fq = lane >> 4) then holds, in the accumulators of tiles 2s and 2s+1, the EIGHT CONSECUTIVE keys
// 32 s + 8 fq .. + 7 -- the plain k order of the P.V MFMA's B operand -- so the V^T fragment is ONE ds_read_b128 (the natural
// order needed two ds_read_b64 halves per fragment, which hipcc merges into the half-rate ds_read2st64_b64).  Only the ROW the K
// fragment read addresses changes; the 128-byte K rows use the chunk swizzle bits (1, 3, 4) of the row, which is conflict-free
// for the permuted rows (bank-group check: 16 distinct 16-byte slots per ds_read_b128 lane group).
__device__ __forceinline__ int kperm_row(int i, int rho) { return 32 * (i >> 1) + 8 * (rho >> 2) + 4 * (i & 1) + (rho & 3); }
__device__ __forceinline__ int kperm_swz(int row) { return ((row >> 1) & 1) | ((row >> 2) & 6); }
// head dimension 128 (ViT-e: 112 padded): K rows are 256 bytes = 16 chunks = one full row of LDS banks, so the 16 rows a
// ds_read_b128 lane group touches (kappa(i, 0..15): bits 0, 1, 3, 4 of the row vary) must land in 16 different chunk columns
__device__ __forceinline__ int kperm_swz16(int row) { return (((row >> 3) & 3) << 2) | (row & 3); }
// max over the 4 lanes l, l^16, l^32, l^48 that share a query: two VALU lane swaps (gfx950) instead of two LDS round trips
__device__ __forceinline__ float quad_rows_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned w = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// QT = query tiles (16 rows each) per wave: a workgroup covers 64*QT queries.  QT = 2 halves both the K/V bytes every
// workgroup streams from L2 (each (window, head) re-reads its K/V once per workgroup) and the LDS fragment reads per MFMA.
// H = bf16_t | f16_t: storage of q / k / v^T / o and of the probabilities handed to the P.V MFMA.
// HDV = head width of V / the output when it differs from the q.k width HD: the EVA-01 MIM ViT (vit_eva.py) adds decomposed relative
// positions to the scores, which this kernel takes as EXTRA q / k channels (q_ext = [scale q | q.Rh | q.Rw], k_ext = [k | one-hot row |
// one-hot column]: modeling/backbone/vit_eva.py) -- HD = 256 / 288 / 320 against HDV = 128.
// chunk swizzle of K rows with >= 16 chunks: XOR of the low four chunk bits (whole groups of 16) -- a trailing group of 4 or 8 chunks
// (HD = 288 / 320) swizzles inside itself
template <int KC> __device__ __forceinline__ int kswz_wide(int cp, int row) {
  constexpr int REM = KC & 15;
  static_assert(REM == 0 || REM == 4 || REM == 8, "K rows of 16 g (+ 4 | 8) chunks");
  const int s = kperm_swz16(row);
  if (REM != 0 && cp >= KC - REM) return (cp & ~15) + ((cp & 15) ^ (s & (REM - 1)));
  return cp ^ s;
}

template <int HD, int QT, bool CAUSAL = false, bool PERM = true, int OCC = 1, typename H = bf16_t, int HDV = HD>
__global__ __launch_bounds__(256, OCC) void attn_bf16_kernel(const AttnParams p) {
  static_assert(HD == 32 || HD == 64 || (HD % 32 == 0 && HD >= 128 && HD <= 320 && PERM), "q.k width 32 / 64 / 128 ... 320 (>= 128: permuted key order only)");
  static_assert(HDV == HD || (HDV == 128 && HD > 128), "V width = q.k width, or 128 under a wider (extended) q.k");
  constexpr int KC = HD / 8;        // 16-byte chunks per K row
  constexpr int KSTEPS = HD / 32;   // MFMA k-steps over d for S
  constexpr int DT = HDV / 16;      // output d tiles
  // two SEPARATE arrays (not one [2][...]) and a tile loop unrolled by two: with a run-time buffer index hipcc cannot tell the
  // LDS-DMA writes of the NEXT tile from the ds_reads of the current one and drains the DMA queue (s_waitcnt vmcnt(0)) in front of
  // the V^T reads of every tile -- the prefetch then overlaps nothing.  Distinct objects are provably disjoint.
  __shared__ __attribute__((aligned(16))) bf16_t smemA[64 * HD + HDV * 64];   // [K tile | Vt tile], even tiles
  __shared__ __attribute__((aligned(16))) bf16_t smemB[64 * HD + HDV * 64];   // odd tiles

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int N = p.N;
  const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q);
  const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K);
  const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.Vt);

  // Q fragments (B operand of S^T = K.Q^T): query = frow of tile u, d = ks*32 + fq*8 .. +8
  int qrow[QT];
  bf16x8_t qf[QT][KSTEPS];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    qrow[u] = qblk * (64 * QT) + (wave * QT + u) * 16 + frow;
    const int qc = qrow[u] < N ? qrow[u] : N - 1;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      qf[u][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(Qp + ((size_t)b * p.bstride + qc) * p.ldq + h * HD + ks * 32 + fq * 8));
  }

  // staging: K tile (64 keys x HD) and Vt tile (HD x 64 keys) go global -> LDS with global_load_lds (no VGPR round trip,
  // nothing for the compiler to spill: the register-staged version kept the prefetched tile in scratch and waited for
  // every global load right after issuing it).  The LDS destination of an instruction is lane-linear (64 x 16 B), so the
  // XOR swizzle of swz_rows() is applied to the per-lane SOURCE chunk instead.
  constexpr int PER = HD * 8 / 256;    // K-tile instructions per wave: HD / 32
  constexpr int PERV = HDV * 8 / 256;  // V^T-tile instructions per wave
  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef const __attribute__((address_space(1))) void gbl_void_t;
  // per-lane source positions as 32-bit element offsets from wave-uniform bases (64-bit per-lane pointers cost twice the registers)
  const bf16_t* kbase = Kp + (size_t)b * p.bstride * p.ldk + h * HD;
  const bf16_t* vbase = Vp + (size_t)(h * HDV) * p.ldvt + (size_t)b * p.bstride;
  int kcol[PER], voff[PERV], krow[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int slot = (wave * PER + i) * 64 + lane;
    const int row = slot / KC, cp = slot % KC;
    const int c = KC >= 16 ? kswz_wide<KC>(cp, row) : KC == 8 ? (cp ^ (PERM ? kperm_swz(row) : ((row >> 1) & 7))) : (cp ^ (((row >> 3) & 1) << 1));
    krow[i] = row;
    kcol[i] = c * 8;
  }
#pragma unroll
  for (int i = 0; i < PERV; ++i) {
    const int slot = (wave * PERV + i) * 64 + lane;
    {
      const int row = slot >> 3, cp = slot & 7;
      const int c = cp ^ ((row >> 1) & 7);
      // PERM: LDS row 16 d + 4 g + r of the V^T tile holds channel 32 (d >> 1) + 8 g + 4 (d & 1) + r, so that after O^T = V^T P^T a
      // lane owns 8 consecutive channels per PAIR of output tiles and one epilogue store has the four lanes of a query write 64
      // contiguous bytes (channel 16 d + 4 g + r gave 8-byte stores, 32 contiguous bytes per instruction)
      const int ch = PERM ? (row >> 5) * 32 + ((row >> 2) & 3) * 8 + ((row >> 4) & 1) * 4 + (row & 3) : row;
      voff[i] = ch * p.ldvt + c * 8;           // < HD * ldvt: 32 bits suffice (ldvt < 2^24)
    }
  }
  auto issue = [&](int t, bf16_t* dst) __attribute__((always_inline)) {
    const int key0 = t * 64;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int key = key0 + krow[i]; key = key < N ? key : N - 1;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(kbase + ((size_t)key * p.ldk + kcol[i])), (lds_void_t*)(dst + (wave * PER + i) * 512), 16, 0, 0);
      if (i < PERV)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(vbase + ((size_t)voff[i] + key0)), (lds_void_t*)(dst + 64 * HD + (wave * PERV + i) * 512), 16, 0, 0);
    }
  };

  // lacc: the softmax denominator, accumulated by the matrix core as one more output tile of O^T = [V^T ; 1] P^T (an
  // all-ones A operand needs no LDS read): sums the SAME bf16-rounded probabilities the numerator uses and takes 16 adds
  // per key tile off the VALU, which bounds this kernel.
  f32x4_t oacc[QT][DT], lacc[QT];
  float m_run[QT];
  const uint32_t one2 = h16<H>::dt == APE_DT_F16 ? 0x3c003c00u : 0x3f803f80u;   // (1.0, 1.0) in H
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(one2, one2, one2, one2));
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    m_run[u] = -INFINITY; lacc[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DT; ++d) oacc[u][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  int nt = (N + 63) / 64;
  if (CAUSAL) {                          // key tiles beyond the workgroup's last query are fully masked: skip them
    const int last_q = min(N - 1, qblk * (64 * QT) + 64 * QT - 1);
    nt = min(nt, last_q / 64 + 1);
  }
  issue(0, smemA);
  __syncthreads();                       // s_waitcnt vmcnt(0) + barrier: tile 0 has landed for every wave
  auto tile = [&](int t, const bf16_t* cur, bf16_t* nxt) __attribute__((always_inline)) {
    if (t + 1 < nt) issue(t + 1, nxt);       // lands while this tile is consumed; the buffer was released by the last barrier
    const bf16_t* sK = cur;
    const bf16_t* sV = cur + 64 * HD;

    // S^T[key][query] for the 64 keys of this tile: 4 key tiles x KSTEPS, every K fragment feeds QT MFMAs
    f32x4_t sacc[QT][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int u = 0; u < QT; ++u) sacc[u][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int krow_ = PERM ? kperm_row(i, frow) : i * 16 + frow;
        const int koff = KC >= 16 ? krow_ * HD + (kswz_wide<KC>(ks * 4 + fq, krow_) << 3)
                       : (PERM && KC == 8) ? krow_ * 64 + (((ks * 4 + fq) ^ kperm_swz(krow_)) << 3) : swz_rows(krow_, ks * 4 + fq, KC);
        const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(&sK[koff]));
#pragma unroll
        for (int u = 0; u < QT; ++u) sacc[u][i] = h16<H>::mfma(kf, qf[u][ks], sacc[u][i]);
      }
    }
    // keys >= N only exist in the last tile
    if (t == nt - 1 && (N & 63) != 0) {
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * 64 + (PERM ? kperm_row(i, fq * 4 + r) : i * 16 + fq * 4 + r) >= N) sacc[u][i][r] = -INFINITY;
    }
    if (CAUSAL) {
      // key > query -> -inf.  Key 0 is visible to every query, so the running maximum is finite from tile 0 on and a fully
      // masked later tile contributes exact zeros.
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * 64 + (PERM ? kperm_row(i, fq * 4 + r) : i * 16 + fq * 4 + r) > qrow[u]) sacc[u][i][r] = -INFINITY;
    }
    uint4 pk[QT][2];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      // running max on the RAW scores (scale > 0 commutes with max)
      float mx = fmaxf(fmaxf(sacc[u][0][0], sacc[u][0][1]), fmaxf(sacc[u][0][2], sacc[u][0][3]));
#pragma unroll
      for (int i = 1; i < 4; ++i) mx = fmaxf(mx, fmaxf(fmaxf(sacc[u][i][0], sacc[u][i][1]), fmaxf(sacc[u][i][2], sacc[u][i][3])));
      if (PERM) {
        mx = quad_rows_max(mx);
      } else {
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      }
      const float m_new = fmaxf(m_run[u], mx * p.scale_log2);
      const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
      // p = 2^(s * scale_log2 - m): one fma + one v_exp_f32 per score
      float pv[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[i][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[u][i][r], p.scale_log2, -m_new));
      m_run[u] = m_new;
      // rescale only when some query of the wave raised its maximum (alpha == 1 exactly otherwise): after the first few
      // key tiles this branch is rarely taken
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
        lacc[u][0] *= alpha; lacc[u][1] *= alpha; lacc[u][2] *= alpha; lacc[u][3] *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          oacc[u][d][0] *= alpha; oacc[u][d][1] *= alpha; oacc[u][d][2] *= alpha; oacc[u][d][3] *= alpha;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        pk[u][s2].x = h16<H>::pack2(pv[2 * s2][0], pv[2 * s2][1]);
        pk[u][s2].y = h16<H>::pack2(pv[2 * s2][2], pv[2 * s2][3]);
        pk[u][s2].z = h16<H>::pack2(pv[2 * s2 + 1][0], pv[2 * s2 + 1][1]);
        pk[u][s2].w = h16<H>::pack2(pv[2 * s2 + 1][2], pv[2 * s2 + 1][3]);
      }
    }
    // O^T[d][query] += Vt[d][key] P^T[key][query]; k-step s covers score tiles 2s, 2s+1.  PERM: their accumulators are keys
    // 32 s + 8 fq + e in order (kperm_row); natural order: e<4 -> key (2s)*16 + fq*4 + e ; e>=4 -> key (2s+1)*16 + fq*4 + (e-4),
    // matched by two 8-byte halves of the V^T row.  Every V fragment feeds QT MFMAs
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int u = 0; u < QT; ++u)
        lacc[u] = h16<H>::mfma(ones, __builtin_bit_cast(bf16x8_t, pk[u][s2]), lacc[u]);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const int row = d * 16 + frow;
        // keys (2s)*16 + fq*4 .. +4 : byte offset in the 128-byte row = s*64 + fq*8 -> chunk s*4 + (fq>>1)
        const int o0 = swz_rows(row, s2 * 4 + (fq >> 1), 8) + (fq & 1) * 4;
        const int o1 = swz_rows(row, s2 * 4 + 2 + (fq >> 1), 8) + (fq & 1) * 4;
        uint4 vk;
        if (PERM) {                       // keys 32 s + 8 fq .. + 7: chunk s*4 + fq of the 128-byte V^T row
          vk = *reinterpret_cast<const uint4*>(&sV[swz_rows(row, s2 * 4 + fq, 8)]);
        } else {
          const uint2 a0 = *reinterpret_cast<const uint2*>(&sV[o0]);
          const uint2 a1 = *reinterpret_cast<const uint2*>(&sV[o1]);
          vk.x = a0.x; vk.y = a0.y; vk.z = a1.x; vk.w = a1.y;
        }
#pragma unroll
        for (int u = 0; u < QT; ++u)
          oacc[u][d] = h16<H>::mfma(__builtin_bit_cast(bf16x8_t, vk), __builtin_bit_cast(bf16x8_t, pk[u][s2]), oacc[u][d]);
      }
    }
    __syncthreads();                       // drains this wave's DMA of tile t + 1 (issued a whole tile ago) and releases `cur`
  };
  for (int t = 0; t < nt; t += 2) {
    tile(t, smemA, smemB);
    if (t + 1 < nt) tile(t + 1, smemB, smemA);
  }
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const float inv = 1.f / lacc[u][0];    // every row of the ones-tile holds the full sum over keys for query frow
    if (qrow[u] < N) {
      if (PERM) {
        bf16_t* o = reinterpret_cast<bf16_t*>(p.O) + ((size_t)b * p.bstride + qrow[u]) * p.ldo + h * HDV + fq * 8;
#pragma unroll
        for (int dp = 0; dp < DT / 2; ++dp) {
          const float v[8] = {oacc[u][2 * dp][0] * inv, oacc[u][2 * dp][1] * inv, oacc[u][2 * dp][2] * inv, oacc[u][2 * dp][3] * inv,
                              oacc[u][2 * dp + 1][0] * inv, oacc[u][2 * dp + 1][1] * inv, oacc[u][2 * dp + 1][2] * inv, oacc[u][2 * dp + 1][3] * inv};
          st8<H>(reinterpret_cast<H*>(o + dp * 32), v);
        }
      } else {
        bf16_t* o = reinterpret_cast<bf16_t*>(p.O) + ((size_t)b * p.bstride + qrow[u]) * p.ldo + h * HDV + fq * 4;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const float v[4] = {oacc[u][d][0] * inv, oacc[u][d][1] * inv, oacc[u][d][2] * inv, oacc[u][d][3] * inv};
          st4<H>(reinterpret_cast<H*>(o + d * 16), v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// f32 validation kernel: block = 64 lanes = 64 queries of one (window, head)
// ------------------------------------------------------------------------------------------
template <int HD, int HDV = HD>
__global__ __launch_bounds__(64) void attn_f32_kernel(const AttnParams p) {
  __shared__ float sK[64][HD];
  __shared__ float sV[64][HDV];
  const int lane = threadIdx.x;
  const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int N = p.N;
  const float* Qp = reinterpret_cast<const float*>(p.Q);
  const float* Kp = reinterpret_cast<const float*>(p.K);
  const float* Vp = reinterpret_cast<const float*>(p.Vt);
  const int qrow = qblk * 64 + lane;
  const int qrow_c = qrow < N ? qrow : N - 1;
  float q[HD], o[HDV];
#pragma unroll
  for (int d = 0; d < HD; ++d) q[d] = Qp[((size_t)b * p.bstride + qrow_c) * p.ldq + h * HD + d] * p.scale;
#pragma unroll
  for (int d = 0; d < HDV; ++d) o[d] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int nt = (N + 63) / 64;
  for (int t = 0; t < nt; ++t) {
    const int key0 = t * 64;
    __syncthreads();
    for (int idx = lane; idx < 64 * HD; idx += 64) {
      const int row = idx / HD, d = idx % HD;
      int key = key0 + row; key = key < N ? key : N - 1;
      sK[row][d] = Kp[((size_t)b * p.bstride + key) * p.ldk + h * HD + d];
    }
    for (int idx = lane; idx < 64 * HDV; idx += 64) {
      const int d = idx / 64, kk = idx % 64;
      const int key = key0 + kk;
      sV[kk][d] = key < N ? Vp[(size_t)(h * HDV + d) * p.ldvt + (size_t)b * p.bstride + key] : 0.f;
    }
    __syncthreads();
    for (int c0 = 0; c0 < 64; c0 += 16) {
      float s[16];
      float mx = -INFINITY;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) a = fmaf(q[d], sK[c0 + kk][d], a);
        if (key0 + c0 + kk >= N || (p.causal && key0 + c0 + kk > qrow)) a = -INFINITY;
        s[kk] = a;
        mx = fmaxf(mx, a);
      }
      const float m_new = fmaxf(m_run, mx);
      if (m_new == -INFINITY) continue;
      const float alpha = expf(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < HDV; ++d) o[d] *= alpha;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float e = expf(s[kk] - m_new);
        l_run += e;
#pragma unroll
        for (int d = 0; d < HDV; ++d) o[d] = fmaf(e, sV[c0 + kk][d], o[d]);
      }
      m_run = m_new;
    }
  }
  if (qrow < N) {
    float* op = reinterpret_cast<float*>(p.O) + ((size_t)b * p.bstride + qrow) * p.ldo + h * HDV;
    const float inv = 1.f / l_run;
#pragma unroll
    for (int d = 0; d < HDV; ++d) op[d] = o[d] * inv;
  }
}

template <typename H>
static int attention_launch_h16(const AttnParams& p, dim3 grid, int B, int N, int bstride, int H_, int HD, int HDV, int causal, hipStream_t s) {
  const void* Q = p.Q; const void* K = p.K; const void* Vt = p.Vt; void* O = p.O;
  const int ldq = p.ldq, ldk = p.ldk, ldvt = p.ldvt, ldo = p.ldo;
  const int nheads = H_;
  APE_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 8 == 0, "ape_hip_attention(bf16): ld alignment");
  APE_CHECK_ARG(((uintptr_t)Q) % 16 == 0 && ((uintptr_t)K) % 16 == 0 && ((uintptr_t)Vt) % 16 == 0 && ((uintptr_t)O) % 16 == 0,
                "ape_hip_attention(bf16): pointer alignment");
  APE_CHECK_ARG(B == 1 || bstride % 8 == 0, "ape_hip_attention(bf16): batched windows need a batch stride %% 8 == 0");
  // 128 queries per workgroup when that still leaves >= 2 workgroups per CU; 64 otherwise (decoder: 900 queries x 8 heads)
  const bool big = (size_t)ceil_div(N, 128) * nheads * B >= 512;
  if (big) grid.x = ceil_div(N, 128);
  static const bool natural = getenv("APE_ATTN_NATURAL_KEY_ORDER") != nullptr;     // A/B: the round-2 kernel (two b64 V halves, LDS shuffles)
  if (HDV != HD) {                  // extended q.k (relative positions as extra channels): 256 / 288 / 320 against V of width 128
    APE_CHECK_ARG(!causal && HDV == 128 && (HD == 256 || HD == 288 || HD == 320), "ape_hip_attention_ext(16-bit): q.k width 256 / 288 / 320 with V width 128");
#define ATT_WIDE(HD_)                                                                                                         \
    do {                                                                                                                      \
      if (big) APE_LAUNCH((attn_bf16_kernel<HD_, 2, false, true, 1, H, 128>), grid, dim3(256), 0, s, p);              \
      else APE_LAUNCH((attn_bf16_kernel<HD_, 1, false, true, 1, H, 128>), grid, dim3(256), 0, s, p);                  \
    } while (0)
    if (HD == 256) ATT_WIDE(256); else if (HD == 288) ATT_WIDE(288); else ATT_WIDE(320);
#undef ATT_WIDE
    return 0;
  }
  if (causal) {
    APE_CHECK_ARG(HD == 64, "ape_hip_attention_causal(bf16): head dimension 64 (every CLIP text tower of the reference)");
    if (big) APE_LAUNCH((attn_bf16_kernel<64, 2, true, true, 1, H>), grid, dim3(256), 0, s, p);
    else APE_LAUNCH((attn_bf16_kernel<64, 1, true, true, 1, H>), grid, dim3(256), 0, s, p);
  } else if (HD == 128) {           // ViT-e (head width 112 zero-padded to 128 by the packing)
    if (big) APE_LAUNCH((attn_bf16_kernel<128, 2, false, true, 1, H>), grid, dim3(256), 0, s, p);
    else APE_LAUNCH((attn_bf16_kernel<128, 1, false, true, 1, H>), grid, dim3(256), 0, s, p);
  } else if (natural) {
    if (HD == 64) { if (big) APE_LAUNCH((attn_bf16_kernel<64, 2, false, false, 1, H>), grid, dim3(256), 0, s, p); else APE_LAUNCH((attn_bf16_kernel<64, 1, false, false, 1, H>), grid, dim3(256), 0, s, p); }
    else { if (big) APE_LAUNCH((attn_bf16_kernel<32, 2, false, false, 1, H>), grid, dim3(256), 0, s, p); else APE_LAUNCH((attn_bf16_kernel<32, 1, false, false, 1, H>), grid, dim3(256), 0, s, p); }
  } else if (HD == 64) {
    static const bool occ4 = getenv("APE_ATTN_OCC4") != nullptr;       // A/B: cap the 128-query kernel at 128 registers (4 waves per SIMD)
    // 256 queries per workgroup (4 query tiles per wave: every K / V^T fragment read from LDS feeds four MFMAs) when that still
    // leaves two workgroups per CU -- the 2-image ViT pass: 2 x 16 heads x 16 blocks (global), 8 windows x 16 heads x 4 blocks
    const char* qt4_env = getenv("APE_ATTN_QT4");
    // round 6: ON by default (APE_ATTN_QT4=0 restores the 128-query kernel): bit-identical outputs (test_attention_four_query_tiles), the
    // kernel alone - 5 % on the 2-image ViT launches (round 3), the driver command + 0.6-0.8 % on one box (profiles/r06_qt4_ab.txt)
    const bool qt4 = (qt4_env ? atoi(qt4_env) != 0 : true) && (size_t)ceil_div(N, 256) * nheads * B >= 512;
    if (qt4) {
      grid.x = ceil_div(N, 256);
      APE_LAUNCH((attn_bf16_kernel<64, 4, false, true, 1, H>), grid, dim3(256), 0, s, p);
    } else if (big && occ4) APE_LAUNCH((attn_bf16_kernel<64, 2, false, true, 4, H>), grid, dim3(256), 0, s, p);
    else if (big) APE_LAUNCH((attn_bf16_kernel<64, 2, false, true, 1, H>), grid, dim3(256), 0, s, p);
    else APE_LAUNCH((attn_bf16_kernel<64, 1, false, true, 1, H>), grid, dim3(256), 0, s, p);
  }
  else { if (big) APE_LAUNCH((attn_bf16_kernel<32, 2, false, true, 1, H>), grid, dim3(256), 0, s, p); else APE_LAUNCH((attn_bf16_kernel<32, 1, false, true, 1, H>), grid, dim3(256), 0, s, p); }
  return 0;
}

static int attention_launch(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo, int B, int N,
                            int bstride, int H, int HD, float scale, int dt, int causal, void* stream, int HDV = 0) {
  if (HDV == 0) HDV = HD;
  APE_CHECK_ARG(Q && K && Vt && O, "ape_hip_attention: null pointer");
  APE_CHECK_ARG(B > 0 && N > 0 && H > 0 && ((HDV == HD && (HD == 32 || HD == 64 || HD == 128)) || (HDV == 128 && (HD == 256 || HD == 288 || HD == 320))),
                "ape_hip_attention: head width 32 / 64 / 128, or an extended q.k width of 256 / 288 / 320 over a V width of 128 (got %d / %d)", HD, HDV);
  APE_CHECK_ARG(bstride >= N, "ape_hip_attention: batch stride %d < N %d", bstride, N);
  AttnParams p;
  p.Q = Q; p.K = K; p.Vt = Vt; p.O = O; p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo; p.N = N; p.H = H; p.bstride = bstride;
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
  p.causal = causal;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ceil_div(N, 64), H, B);
  if (ape_is16(dt)) {
    const int rc = dt == APE_DT_F16 ? attention_launch_h16<f16_t>(p, grid, B, N, bstride, H, HD, HDV, causal, s)
                                    : attention_launch_h16<bf16_t>(p, grid, B, N, bstride, H, HD, HDV, causal, s);
    if (rc != 0) return rc;
  } else {
    if (HD == 256) APE_LAUNCH((attn_f32_kernel<256, 128>), grid, dim3(64), 0, s, p);
    else if (HD == 288) APE_LAUNCH((attn_f32_kernel<288, 128>), grid, dim3(64), 0, s, p);
    else if (HD == 320) APE_LAUNCH((attn_f32_kernel<320, 128>), grid, dim3(64), 0, s, p);
    else if (HD == 128) APE_LAUNCH(attn_f32_kernel<128>, grid, dim3(64), 0, s, p);
    else if (HD == 64) APE_LAUNCH(attn_f32_kernel<64>, grid, dim3(64), 0, s, p);
    else APE_LAUNCH(attn_f32_kernel<32>, grid, dim3(64), 0, s, p);
  }
  APE_CHECK_LAUNCH("ape_hip_attention");
  return 0;
}

extern "C" int ape_hip_attention_strided(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                                         int B, int N, int bstride, int H, int HD, float scale, int dt, void* stream) {
  return attention_launch(Q, ldq, K, ldk, Vt, ldvt, O, ldo, B, N, bstride, H, HD, scale, dt, 0, stream);
}

// q.k width HDQ != V width HDV (relative-position channels appended to q / k): see attn_bf16_kernel
extern "C" int ape_hip_attention_ext(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                                     int B, int N, int bstride, int H, int HDQ, int HDV, float scale, int dt, void* stream) {
  return attention_launch(Q, ldq, K, ldk, Vt, ldvt, O, ldo, B, N, bstride, H, HDQ, scale, dt, 0, stream, HDV);
}

extern "C" int ape_hip_attention_causal(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                                        int B, int N, int bstride, int H, int HD, float scale, int dt, void* stream) {
  return attention_launch(Q, ldq, K, ldk, Vt, ldvt, O, ldo, B, N, bstride, H, HD, scale, dt, 1, stream);
}

extern "C" int ape_hip_attention(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                                 int B, int N, int H, int HD, float scale, int dt, void* stream) {
  return ape_hip_attention_strided(Q, ldq, K, ldk, Vt, ldvt, O, ldo, B, N, N, H, HD, scale, dt, stream);
}
