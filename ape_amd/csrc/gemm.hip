// GEMM family for the APE forward pass on gfx950:  C[M,N] = epi(alpha * A[M,K] . W[N,K]^T)
//
// Replaces every nn.Linear / 1x1-conv / patch-embed / deconv contraction on the hot path
// (reference call sites: ape/modeling/backbone/vit_eva_clip.py:225-232,264-268,125-132;
//  ape/layers/multi_scale_deform_attn.py:268-277,353; detrex FFN/MLP via
//  ape/modeling/ape_deta/deformable_transformer_vl.py:45-54,140-167).
//
// bf16 path: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles,
//   register-staged global->LDS double buffering (one barrier per K tile), XOR-swizzled LDS rows
//   (conflict-free ds_read_b128 fragments), XCD-aware tile order.  The MFMA operands are swapped
//   (D = W.A^T) so each lane owns 4 consecutive output columns -> 8/16-byte epilogue stores.
// f32 path: plain LDS-tiled FMA kernel with the same epilogue (exact-math validation mode).
#include <stdlib.h>

#include <type_traits>
#include "common.h"
#include "../../include/ape_hip.h"

#include "gemm_epi.h"

const char* ape_gemm_p8_launch(ApeGemmArgs p, int bn, int stagger, hipStream_t s);   // gemm_p8.hip

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------
#define GB_M 128
#define GB_N 128
#define GB_K 64

__device__ __forceinline__ int swz128(int row, int c) {  // 128-byte rows (8 chunks of 16 B)
  return row * 64 + ((c ^ ((row >> 1) & 7)) << 3);
}

template <bool TRANS, typename H = bf16_t>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][2][GB_M * GB_K];  // [buf][A|W] 64 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap: each XCD (blockIdx % 8) walks a contiguous range of tiles so that
  // the A rows it streams stay in its private L2.
  const int tiles_n = (p.N + GB_N - 1) / GB_N;
  const int nblk = gridDim.x;
  int id;
  {
    const int b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, j = b >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tm = id / tiles_n, tn = id % tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;

  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);

  // staging: 1024 16-byte chunks per operand tile, 4 per thread
  const bf16_t* ga[4];
  const bf16_t* gw[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cid = tid + 256 * i;
    const int row = cid >> 3, c = cid & 7;
    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
    ga[i] = A + (size_t)gm * p.lda + c * 8;
    gw[i] = W + (size_t)gn * p.ldw + c * 8;
    soff[i] = swz128(row, c);
  }
  uint4 ra[4], rw[4];
  const int kchunk = (tid & 7) * 8;  // same 16-byte chunk column for all 4 staged rows of this thread
  auto gload = [&](int kt) {
    const bool kin = kt * GB_K + kchunk < p.K;  // K tail (K % 8 == 0): chunks past K read as zeros
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = kin ? *reinterpret_cast<const uint4*>(ga[i] + kt * GB_K) : make_uint4(0u, 0u, 0u, 0u);
      rw[i] = kin ? *reinterpret_cast<const uint4*>(gw[i] + kt * GB_K) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(&smem[buf][0][soff[i]]) = ra[i];
      *reinterpret_cast<uint4*>(&smem[buf][1][soff[i]]) = rw[i];
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + frow;
        af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(&smem[buf][0][swz128(row, ks * 4 + fq)]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + frow;
        wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(&smem[buf][1][swz128(row, ks * 4 + fq)]));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (TRANS) acc[i][j] = h16<H>::mfma(af[i], wf[j], acc[i][j]);
          else acc[i][j] = h16<H>::mfma(wf[j], af[i], acc[i][j]);
        }
    }
  };

  const int nk = (p.K + GB_K - 1) / GB_K;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    compute(buf);
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (TRANS) {
        // D[i = m_local = fq*4 + r][j = n_local = frow]
        epi_m4<H>(p, m0 + wm * 64 + i * 16 + fq * 4, n0 + wn * 64 + j * 16 + frow, v);
      } else {
        // D[i = n_local = fq*4 + r][j = m_local = frow]
        epi_n4<H>(p, m0 + wm * 64 + i * 16 + frow, n0 + wn * 64 + j * 16 + fq * 4, v);
      }
    }
}


// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel v2.  Same 128x128x64 tiling / swizzled LDS image as above, but
//  * GLDS: the operand tiles go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write
//    pass: the ds_write path, ~80 B/clk/CU, was the LDS bottleneck of the register-staged loop).  The XOR
//    swizzle is applied to the per-lane SOURCE address; the LDS destination stays lane-linear.
//  * the epilogue goes through LDS so that every global store is a 16-byte chunk of a full output row
//    (the MFMA accumulator layout only yields 8-byte pieces 32 B apart, which made the K=256 GEMMs of the
//    deformable encoder store-bound).
// ------------------------------------------------------------------------------------------
#define GEMM_T64_LDS (4 * 128 * 32 * 2)  /* 64x64 ring: 4 stages x (64+64) rows x 64 B = 32 KiB >= the 64 x (64*4+16) B fp32 epilogue tile */
#define GEMM_T128x64_LDS (4 * (128 + 64) * 32 * 2)  /* 48 KiB of stages >= the 128 x (64*4+16) B epilogue tile */
#define GEMM_V2_LDS (GB_M * (GB_N * 4 + 16))  /* 67584 B: fp32 epilogue tile; >= the 64 KiB operand buffers */
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// LDS-staged epilogue shared by the MFMA kernels: tile [128 rows][tcols], pitch + 16 B, then 16-byte row chunks
template <bool TRANS, int TM = 4, int TN = 4, typename H = bf16_t>
__device__ __forceinline__ void epilogue_via_lds(const GemmParams& p, f32x4_t (&acc)[TM][TN], unsigned char* smem_raw, int m0, int n0,
                                                 int tid, int wm, int wn, int frow, int fq) {
  constexpr int BM = 32 * TM, BN = 32 * TN;   // block tile; each of the 2x2 waves owns (16*TM) x (16*TN)
  const int esz = p.out_dt == APE_DT_F32 ? 4 : 2;
  const bool swiglu = p.act == APE_ACT_SWIGLU;
  const int tcols = TRANS ? BM : (swiglu ? BN / 2 : BN);                 // output columns held by the tile
  const int pitch = tcols * esz + 16;                         // bytes
  unsigned char* tile = reinterpret_cast<unsigned char*>(smem_raw);
  const int out_rows = TRANS ? p.N : p.M, out_cols = TRANS ? p.M : (swiglu ? (p.N >> 1) : p.N);
  const int orow0 = TRANS ? n0 : m0;
  const int ocol0 = TRANS ? m0 : (swiglu ? (n0 >> 1) : n0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      int lrow, lcol, cnt;
      if (TRANS) {
        const int n = n0 + wn * (16 * TN) + j * 16 + frow, mm = m0 + wm * (16 * TM) + i * 16 + fq * 4;
        if (n >= p.N || mm >= p.M) continue;
        epi_m4_values(p, n, v);
        lrow = wn * (16 * TN) + j * 16 + frow; lcol = wm * (16 * TM) + i * 16 + fq * 4; cnt = 4;
      } else {
        const int m = m0 + wm * (16 * TM) + i * 16 + frow, nn = n0 + wn * (16 * TN) + j * 16 + fq * 4;
        if (m >= p.M || nn >= p.N) continue;
        int ocol;
        epi_n4_values<H>(p, m, nn, v, ocol, cnt);
        lrow = wm * (16 * TM) + i * 16 + frow; lcol = ocol - ocol0; cnt = swiglu ? 2 : 4;
      }
      unsigned char* dst = tile + lrow * pitch + lcol * esz;
      if (esz == 4) {
        if (cnt == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
      } else {
        if (cnt == 4) *reinterpret_cast<uint2*>(dst) = make_uint2(h16<H>::pack2(v[0], v[1]), h16<H>::pack2(v[2], v[3]));
        else *reinterpret_cast<uint32_t*>(dst) = h16<H>::pack2(v[0], v[1]);
      }
    }
  __syncthreads();
  // coalesced copy-out: 128 rows x (tcols*esz/16) 16-byte chunks; consecutive lanes -> consecutive chunks of a row
  const int cpr = tcols * esz / 16;
  const int per = 16 / esz;  // elements per chunk
  const int trows = TRANS ? BN : BM;
  const int cpr_sh = __ffs(cpr) - 1;   // tile widths are 32/64/128 columns of 2 or 4 bytes: cpr is a power of two
  for (int cid = tid; cid < trows * cpr; cid += 256) {
    const int lrow = cid >> cpr_sh, cc = cid & (cpr - 1);
    const int grow = orow0 + lrow;
    const int gcol = ocol0 + cc * per;
    if (grow >= out_rows || gcol >= out_cols) continue;
    const unsigned char* src = tile + lrow * pitch + cc * 16;
    unsigned char* gdst = reinterpret_cast<unsigned char*>(p.C) + ((size_t)grow * p.ldc + gcol) * esz;
    if (gcol + per <= out_cols) {
      *reinterpret_cast<uint4*>(gdst) = *reinterpret_cast<const uint4*>(src);
    } else {
      const int rem = out_cols - gcol;
      if (esz == 4) for (int e = 0; e < rem; ++e) reinterpret_cast<float*>(gdst)[e] = reinterpret_cast<const float*>(src)[e];
      else for (int e = 0; e < rem; ++e) reinterpret_cast<bf16_t*>(gdst)[e] = reinterpret_cast<const bf16_t*>(src)[e];
    }
  }
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel v3 ("ring"): 128x128 tile, BK = 32, FOUR LDS stages filled by global_load_lds and consumed
// behind counted `s_waitcnt vmcnt(N)` + raw `s_barrier`, so three K-tiles stay in flight across every barrier
// (the v2 loop drained its single prefetched tile with vmcnt(0) each iteration and sat on HBM/L2 latency).
// LDS image per stage: A [128][32] | W [128][32] bf16 (64-byte rows, chunk ^= ((row>>3)&1)<<1 keeps the 16-row
// ds_read_b128 fragments conflict-free).  Needs K % 32 == 0.
// ------------------------------------------------------------------------------------------
#define GR_K 32
#define GR_STAGES 4
__device__ __forceinline__ int swz64(int row, int c) { return row * 32 + ((c ^ (((row >> 3) & 1) << 1)) << 3); }

template <bool TRANS, int TM, int TN, typename H = bf16_t>
__global__ __launch_bounds__(256, 2) void gemm_bf16_ring_kernel(const GemmParams p) {
  constexpr int BM = 32 * TM, BN = 32 * TN;           // 128x128 (TM=TN=4) or 64x64 (TM=TN=2) block tile
  constexpr int STAGE = (BM + BN) * GR_K;              // elements per LDS stage: A [BM][32] | W [BN][32]
  constexpr int GA = TM / 2, GW = TN / 2;              // 16-row glds groups per wave for A and for W
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // GEMM_V2_LDS bytes; stages use the first 64 KiB
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);                        // stage s: A at s*STAGE, W at s*STAGE + BM*32 (elements)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nblk = gridDim.x;
  int id;
  {
    const int b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, j = b >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tm = id / tiles_n, tn = id % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);

  // wave w fills row groups 2w, 2w+1 (16 rows x 64 B = 1 KiB per instruction) of A and of W
  const bf16_t* ga[GA];
  const bf16_t* gw[GW];
  int soa[GA], sow[GW];
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    const int rg = wave * GA + i;
    const int row = rg * 16 + (lane >> 2);
    const int c = (lane & 3) ^ (((row >> 3) & 1) << 1);
    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
    ga[i] = A + (size_t)gm * p.lda + c * 8;
    soa[i] = rg * 512;
  }
#pragma unroll
  for (int i = 0; i < GW; ++i) {
    const int rg = wave * GW + i;
    const int row = rg * 16 + (lane >> 2);
    const int c = (lane & 3) ^ (((row >> 3) & 1) << 1);
    int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
    gw[i] = W + (size_t)gn * p.ldw + c * 8;
    sow[i] = rg * 512;
  }
  constexpr int LPT = GA + GW;   // glds instructions per wave per K-tile
  auto issue = [&](int kt) {
    bf16_t* st = smem + (kt & (GR_STAGES - 1)) * STAGE;
#pragma unroll
    for (int i = 0; i < GA; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(ga[i] + kt * GR_K), (lds_void_t*)(st + soa[i]), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < GW; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gw[i] + kt * GR_K), (lds_void_t*)(st + BM * GR_K + sow[i]), 16, 0, 0);
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;

  // split-K: blockIdx.y owns k-tiles [kbeg, kbeg + nk)
  const int nk_total = p.K / GR_K;
  const int splits = gridDim.y;
  const int per_split = (nk_total + splits - 1) / splits;
  const int kbeg = blockIdx.y * per_split;
  const int nk = (nk_total - kbeg) < per_split ? (nk_total - kbeg) : per_split;
#pragma unroll
  for (int i = 0; i < GA; ++i) ga[i] += (size_t)kbeg * GR_K;
#pragma unroll
  for (int i = 0; i < GW; ++i) gw[i] += (size_t)kbeg * GR_K;
  auto load_frags = [&](int kt, bf16x8_t (&af)[TM], bf16x8_t (&wf)[TN]) {
    const bf16_t* sa = smem + (kt & (GR_STAGES - 1)) * STAGE;
    const bf16_t* sw = sa + BM * GR_K;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(sa + swz64(wm * (16 * TM) + i * 16 + frow, fq)));
#pragma unroll
    for (int j = 0; j < TN; ++j)
      wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(sw + swz64(wn * (16 * TN) + j * 16 + frow, fq)));
  };
  auto mma = [&](bf16x8_t (&af)[TM], bf16x8_t (&wf)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (TRANS) acc[i][j] = h16<H>::mfma(af[i], wf[j], acc[i][j]);
        else acc[i][j] = h16<H>::mfma(wf[j], af[i], acc[i][j]);
      }
  };
  // Opaque "use" of a fragment set: hipcc places the lgkmcnt wait for these registers HERE (before the barrier and
  // before the next set's ds_reads are issued) instead of in front of the first MFMA, where it would also wait for the
  // freshly issued prefetch reads and serialise LDS latency with the matrix pipe.
  auto touch = [&](bf16x8_t (&af)[TM], bf16x8_t (&wf)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(wf[j]));
  };
  // wait until this wave's loads of tile `t` have landed, given that tiles up to min(t+2, nk-1) were issued after it
  auto wait_n_tiles_in_flight = [&](int tiles) {   // allow `tiles` younger K-tiles (LPT loads each) to stay outstanding
    if (tiles >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
    else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto wait_tile = [&](int t) { wait_n_tiles_in_flight(t + 2 < nk ? 2 : (t + 1 < nk ? 1 : 0)); };
  // Software pipeline: fragments of tile t+1 are read from LDS while the 16 MFMAs of tile t run from registers.
  //   iteration t: [tile t+1 landed] -> barrier -> issue tile t+4 into the stage tile t just vacated ->
  //                ds_read fragments(t+1) || mfma(fragments(t))
  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  if (nk > 2) issue(2);
  wait_tile(0);
  __builtin_amdgcn_s_barrier();
  if (nk > 3) issue(3);
  bf16x8_t fa0[TM], fw0[TN], fa1[TM], fw1[TN];
  if (nk > 0) load_frags(0, fa0, fw0);
  for (int kt = 0; kt < nk; kt += 2) {
    // ---- even tile kt from (fa0, fw0); prefetch kt+1 into (fa1, fw1)
    if (kt + 1 < nk) wait_tile(kt + 1);   // tiles issued so far: up to min(kt+3, nk-1)
    touch(fa0, fw0);   // our own fragment reads of tile kt are complete (see touch())
    __builtin_amdgcn_s_barrier();
    if (kt + 4 < nk) issue(kt + 4);
    if (kt + 1 < nk) load_frags(kt + 1, fa1, fw1);
    mma(fa0, fw0);
    if (kt + 1 >= nk) break;
    // ---- odd tile kt+1 from (fa1, fw1); prefetch kt+2 into (fa0, fw0)
    if (kt + 2 < nk) wait_tile(kt + 2);
    touch(fa1, fw1);
    __builtin_amdgcn_s_barrier();
    if (kt + 5 < nk) issue(kt + 5);
    if (kt + 2 < nk) load_frags(kt + 2, fa0, fw0);
    mma(fa1, fw1);
  }
  __syncthreads();   // all operand reads done before the epilogue reuses the LDS
  if (splits > 1) {
    // raw fp32 partial tile -> workspace plane of this split; gemm_splitk_reduce_kernel finishes the job
    GemmParams q = p;
    q.C = p.workspace + (size_t)blockIdx.y * p.M * p.N;
    q.ldc = p.N; q.out_dt = APE_DT_F32; q.bias = nullptr; q.residual = nullptr; q.rowmask = nullptr; q.rope_cos = nullptr; q.rowscale = nullptr;
    q.act = APE_ACT_NONE; q.alpha = 1.f; q.clamp = 0.f;
    if (nk <= 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    epilogue_via_lds<TRANS, TM, TN, H>(q, acc, smem_raw, m0, n0, tid, wm, wn, frow, fq);
    return;
  }
  epilogue_via_lds<TRANS, TM, TN, H>(p, acc, smem_raw, m0, n0, tid, wm, wn, frow, fq);
}

// sum the split-K partial planes and apply the full epilogue; one thread per 4 consecutive columns
template <typename H = bf16_t>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GemmParams p) {
  const int ngrp = (p.N + 3) / 4;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)p.M * ngrp) return;
  const int m = (int)(gid / ngrp), n0 = (int)(gid % ngrp) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const size_t plane = (size_t)p.M * p.N;
  const float* src = p.workspace + (size_t)m * p.N + n0;
  const bool full = (n0 + 3 < p.N) && (p.N % 4 == 0);
  for (int s = 0; s < p.splitk; ++s) {
    if (full) {
      const float4 t = *reinterpret_cast<const float4*>(src + s * plane);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    } else {
      for (int r = 0; r < 4 && n0 + r < p.N; ++r) v[r] += src[s * plane + r];
    }
  }
  epi_n4<H>(p, m, n0, v);
}

template <bool TRANS, bool GLDS, typename H = bf16_t>
__global__ __launch_bounds__(256, 2) void gemm_bf16_v2_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // GEMM_V2_LDS bytes
  bf16_t (*smem)[2][GB_M * GB_K] = reinterpret_cast<bf16_t (*)[2][GB_M * GB_K]>(smem_raw);  // [buf][A|W] 64 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + GB_N - 1) / GB_N;
  const int nblk = gridDim.x;
  int id;
  {
    const int b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, j = b >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tm = id / tiles_n, tn = id % tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);

  // ---- staging addresses
  // GLDS: wave w fills row groups rg = 4w..4w+3 (8 rows x 128 B = 1 KiB per instruction); lane -> (row rg*8 + lane/8,
  //       LDS chunk lane%8) and fetches global chunk (lane%8) ^ f(row) so that reads use the usual swz128().
  // !GLDS: register staging exactly as in v1 (supports a K tail).
  const bf16_t* ga[4];
  const bf16_t* gw[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row, c;
    if (GLDS) {
      row = (wave * 4 + i) * 8 + (lane >> 3);
      c = (lane & 7) ^ ((row >> 1) & 7);
      soff[i] = (wave * 4 + i) * 512;  // wave-uniform LDS base (elements) of this row group
    } else {
      const int cid = tid + 256 * i;
      row = cid >> 3; c = cid & 7;
      soff[i] = swz128(row, c);
    }
    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
    ga[i] = A + (size_t)gm * p.lda + c * 8;
    gw[i] = W + (size_t)gn * p.ldw + c * 8;
  }
  uint4 ra[4], rw[4];
  const int kchunk = (tid & 7) * 8;
  auto stage_issue = [&](int kt, int buf) {
    if (GLDS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(ga[i] + kt * GB_K), (lds_void_t*)(&smem[buf][0][soff[i]]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(gw[i] + kt * GB_K), (lds_void_t*)(&smem[buf][1][soff[i]]), 16, 0, 0);
      }
    } else {
      const bool kin = kt * GB_K + kchunk < p.K;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = kin ? *reinterpret_cast<const uint4*>(ga[i] + kt * GB_K) : make_uint4(0u, 0u, 0u, 0u);
        rw[i] = kin ? *reinterpret_cast<const uint4*>(gw[i] + kt * GB_K) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto stage_commit = [&](int buf) {
    if (GLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<uint4*>(&smem[buf][0][soff[i]]) = ra[i];
        *reinterpret_cast<uint4*>(&smem[buf][1][soff[i]]) = rw[i];
      }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + frow;
        af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(&smem[buf][0][swz128(row, ks * 4 + fq)]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + frow;
        wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(&smem[buf][1][swz128(row, ks * 4 + fq)]));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (TRANS) acc[i][j] = h16<H>::mfma(af[i], wf[j], acc[i][j]);
          else acc[i][j] = h16<H>::mfma(wf[j], af[i], acc[i][j]);
        }
    }
  };

  const int nk = (p.K + GB_K - 1) / GB_K;
  stage_issue(0, 0);
  stage_commit(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) stage_issue(kt + 1, buf ^ 1);
    compute(buf);
    if (kt + 1 < nk) stage_commit(buf ^ 1);
    __syncthreads();
  }

  epilogue_via_lds<TRANS, 4, 4, H>(p, acc, smem_raw, m0, n0, tid, wm, wn, frow, fq);
}

// ------------------------------------------------------------------------------------------
// f32 kernel (validation mode): 64x64x16 tile, 256 threads, 4x4 outputs / thread
// ------------------------------------------------------------------------------------------
template <bool TRANS, typename H = bf16_t>   // H: what a 2-byte output / residual of this fp32-operand launch holds
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
  __shared__ float sA[16][64 + 4];
  __shared__ float sW[16][64 + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int tiles_n = (p.N + 63) / 64;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  const float* __restrict__ A = reinterpret_cast<const float*>(p.A);
  const float* __restrict__ W = reinterpret_cast<const float*>(p.W);
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 4, kk = idx & 15;
      const int gm = m0 + row, gn = n0 + row, gk = k0 + kk;
      sA[kk][row] = (gm < p.M && gk < p.K) ? A[(size_t)gm * p.lda + gk] : 0.f;
      sW[kk][row] = (gn < p.N && gk < p.K) ? W[(size_t)gn * p.ldw + gk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sA[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = sW[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
    }
    __syncthreads();
  }
  if (TRANS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
      epi_m4<H>(p, m0 + ty * 4, n0 + tx * 4 + j, v);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) epi_n4<H>(p, m0 + ty * 4 + i, n0 + tx * 4, acc[i]);
  }
}

// ------------------------------------------------------------------------------------------
// GEMV (M small): out[m][n] = scale[n] * (alpha * sum_k x[m][k] * W[n][k] + bias[n]); one wave per n.
// Optional second output out2[m][n] = add[m][n] + out[m][n] (the single-token language side's residual / folded-bias updates,
// layers/fuse_helper.py: gamma_v * delta_v next to LN bias + gamma_v * delta_v, l + gamma_l * delta_l).
// ------------------------------------------------------------------------------------------
template <typename TW>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ x, int ldx, const TW* __restrict__ W,
                                                   int ldw, const float* __restrict__ bias, float* __restrict__ out,
                                                   int ldo, int M, int N, int K, float alpha, const float* __restrict__ scale,
                                                   const float* __restrict__ add, int ldadd, float* __restrict__ out2, int ldo2) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const TW* w = W + (size_t)n * ldw;
  // 4 elements per lane per step when the rows allow 16-byte (fp32) / 8-byte (bf16) accesses
  const bool vec = (K % 4 == 0) && (ldw % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)W) % 16 == 0) && (((uintptr_t)x) % 16 == 0);
  for (int m = 0; m < M; ++m) {
    const float* xr = x + (size_t)m * ldx;
    float s = 0.f;
    if (vec) {
      for (int k = lane * 4; k < K; k += 256) {
        float wv[4], xv[4];
        ld4<TW>(w + k, wv);
        ld4<float>(xr + k, xv);
        s = fmaf(xv[0], wv[0], s); s = fmaf(xv[1], wv[1], s); s = fmaf(xv[2], wv[2], s); s = fmaf(xv[3], wv[3], s);
      }
    } else {
      for (int k = lane; k < K; k += 64) s = fmaf(xr[k], ldf<TW>(w + k), s);
    }
    s = wave_sum(s);
    if (lane == 0) {
      float r = s * alpha + (bias != nullptr ? bias[n] : 0.f);
      if (scale != nullptr) r *= scale[n];
      out[(size_t)m * ldo + n] = r;
      if (out2 != nullptr) out2[(size_t)m * ldo2 + n] = add[(size_t)m * ldadd + n] + r;
    }
  }
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel "kres" for K == 256 (the 256-wide deformable encoder / decoder linears over 87 k tokens).
// The whole K extent of the activation operand stays in REGISTERS: every wave owns 32 rows of A as 2 x 8 MFMA
// B-operand fragments (64 VGPRs, loaded once, straight from global memory), and a workgroup (4 waves = 128 rows) walks
// over the output columns in 128-wide chunks.  Only W streams through LDS (ring of four [128 n][64 k] stages filled
// by global_load_lds behind counted vmcnt waits), so the L2->LDS fill per flop is half that of a 128x128 tile and a
// quarter of the 64x64 tile these shapes used before, and A is read from HBM exactly once.
//  * W rows are PERMUTED on their way into LDS: ring row rho = 16 j + 4 g + r holds chunk column 32 g + 4 j + r, so
//    that after D = W A^T lane (m = lane & 15, g = lane >> 4) owns 32 CONSECUTIVE output columns of its row: the
//    epilogue stores 16-byte pieces from registers (no LDS round trip) and 4 lanes cover a 256-byte row segment.
//  * bias lives in LDS (lgkmcnt domain); the residual rows of a chunk are fetched BEFORE the last W stage of the chunk
//    is requested, so consuming them never drains the W prefetch queue (VMEM returns in issue order).
//  * the wait immediates count the residual loads / output stores issued between W stages; a chunk or row block that
//    needs bounds checks takes a predicated epilogue followed by a full drain instead.
// ------------------------------------------------------------------------------------------
#define KR_BM 128
#define KR_CH 128
#define KR_MAXN 2048
#define KR_LDS (4 * 16384 + KR_MAXN * 4)

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int RES /*0 none, 1 = 16-bit H (2 = f32: compiles, spills, not instantiated)*/, bool OUT_F32, bool OUT_F16 = false /* 2-byte output is IEEE half */,
          typename H = bf16_t /* operand / residual storage */>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kres_kernel(const GemmParams p) {
  typedef typename std::conditional<OUT_F16, f16_t, H>::type HO;             // what a 2-byte output holds
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* ring = reinterpret_cast<bf16_t*>(smem_raw);                     // 4 stages x [128][64] bf16 (swz128 image)
  float* sbias = reinterpret_cast<float*>(smem_raw + 4 * 16384);          // bias of this block's column range
  constexpr int NR = RES == 0 ? 0 : (RES == 1 ? 8 : 16);                  // residual loads per lane per chunk
  constexpr int NE = OUT_F32 ? 16 : 8;                                    // output stores per lane per chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int nch_total = (p.N + KR_CH - 1) / KR_CH;
  // TAIL SPLIT (launcher: p.reserved0 = row blocks that take all columns | column parts of the rest << 16; 0 = off).  Two workgroups
  // fit a CU, so 87 296 rows = 682 row blocks run as one full round of 512 workgroups and a second round of 170 -- a third of the
  // slots -- which costs as much time as the first.  The launcher therefore cuts the LAST round's row blocks into S column parts each
  // (S x 170 <= 512 workgroups, all resident at once, each with 1 / S of the columns behind its own copy of the A rows).
  int row_block = blockIdx.x, cbeg, nch;
  const int nfull = p.reserved0 & 0xffff, tparts = p.reserved0 >> 16;
  if (tparts > 0 && (int)blockIdx.x >= nfull) {
    const int t = blockIdx.x - nfull;
    row_block = nfull + t / tparts;
    const int part = t % tparts;
    cbeg = part * nch_total / tparts;                              // balanced: every part gets floor or ceil of nch_total / tparts
    nch = (part + 1) * nch_total / tparts - cbeg;
  } else {
    const int per = (nch_total + gridDim.y - 1) / gridDim.y;
    cbeg = blockIdx.y * per;
    nch = min(per, nch_total - cbeg);
  }
  if (nch <= 0) return;
  const int nbeg = cbeg * KR_CH;
  for (int i = tid; i < nch * KR_CH; i += 256) sbias[i] = (p.bias != nullptr && nbeg + i < p.N) ? p.bias[nbeg + i] : 0.f;
  __syncthreads();

  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);
  const int m_wave = row_block * KR_BM + wave * 32;
  const bool rows_full = (row_block + 1) * KR_BM <= p.M;

  // ---- A: 2 row tiles x 8 k-steps of B-operand fragments (lane: row = frow, k = s*32 + fq*8 .. +8)
  bf16x8_t af[2][8];
  int mrow[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m_wave + mi * 16 + frow;
    mrow[mi] = m;
    const bf16_t* ap = A + (size_t)(m < p.M ? m : p.M - 1) * p.lda + fq * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) af[mi][s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(ap + s * 32));
  }
  bool masked[2] = {false, false};
  if (p.rowmask != nullptr) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) masked[mi] = mrow[mi] < p.M && p.rowmask[mrow[mi]] != 0;
  }

  // ---- W stage loads: wave w issues instructions q = 4w .. 4w+3, each 64 lanes x 16 B = ring rows 8q .. 8q+7
  int wcol[4], wk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rho = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((rho >> 1) & 7);
    const int idx = rho & 15;
    // chunk column held by ring row rho = 16 j + 4 g + r: 32 (j >> 1) + 8 g + 4 (j & 1) + r -- after D = W A^T lane (m, g) then
    // owns, for every PAIR of column tiles jp = j >> 1, the 8 consecutive columns 32 jp + 8 g .. + 7, so that one store
    // instruction (fixed jp) has the four lanes g = 0..3 of a row write 64 (f32: 128) CONTIGUOUS bytes.  (Round 2 gave a lane 32
    // consecutive columns: every store instruction then touched 64 sectors with one isolated 16-byte piece each, and the
    // write-bound launches of this kernel sat at ~2 TB/s.)
    wcol[i] = (rho >> 5) * 32 + (idx >> 2) * 8 + ((rho >> 4) & 1) * 4 + (idx & 3);
    wk[i] = c * 8;
  }
  const int T = nch * 4;
  auto issue = [&](int t) {
    t = t < T ? t : T - 1;                                           // past the end: harmless reload, keeps the counts uniform
    const int cc = t >> 2, ks = t & 3;
    bf16_t* st = ring + (t & 3) * 8192;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int n = nbeg + cc * KR_CH + wcol[i];
      n = n < p.N ? n : p.N - 1;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(W + (size_t)n * p.ldw + ks * 64 + wk[i]), (lds_void_t*)(st + (wave * 4 + i) * 512), 16, 0, 0);
    }
  };

  f32x4_t acc[2][8];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[mi][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int t, int ks) {
    const bf16_t* st = ring + (t & 3) * 8192;
    // W fragments in half-sets of 4 column tiles (16 VGPRs): set h+1 is read from LDS while the 8 MFMAs of set h run.
    // The sched_barriers pin that order (left alone, hipcc hoists all 16 reads of a stage and the kernel spills, and
    // scratch traffic would break the vmcnt bookkeeping).
    auto rd = [&](int h, bf16x8_t (&wf)[4]) {
      const int kk = h >> 1, jb = (h & 1) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rho = (jb + j) * 16 + frow;
        wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(st + rho * 64 + (((kk * 4 + fq) ^ ((rho >> 1) & 7)) << 3)));
      }
    };
    auto mm = [&](int h, bf16x8_t (&wf)[4]) {
      const int kk = h >> 1, jb = (h & 1) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi][jb + j] = h16<H>::mfma(wf[j], af[mi][ks * 2 + kk], acc[mi][jb + j]);
    };
    bf16x8_t w0[4], w1[4];
    rd(0, w0);
    rd(1, w1);
    __builtin_amdgcn_sched_barrier(0);
    mm(0, w0);
    __builtin_amdgcn_sched_barrier(0);
    rd(2, w0);
    mm(1, w1);
    __builtin_amdgcn_sched_barrier(0);
    rd(3, w1);
    mm(2, w0);
    __builtin_amdgcn_sched_barrier(0);
    mm(3, w1);
  };

  // residual registers of the current chunk: lane owns row mrow[mi], columns n_lane + 32 q .. + 7 (q = 0..3), n_lane = chunk + 8 fq
  static_assert(RES != 2, "the f32-residual flavour was never instantiated (register budget); its piece layout is not maintained");
  uint4 rres[RES == 0 ? 1 : 2][4];
  auto load_res = [&](int n_lane) {
    if (RES == 0) return;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const unsigned char* rp = reinterpret_cast<const unsigned char*>(p.residual) + ((size_t)mrow[mi] * p.ldr + n_lane) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) rres[RES == 0 ? 0 : mi][q] = *reinterpret_cast<const uint4*>(rp + q * 64);
    }
  };
  // epilogue arithmetic on one accumulator quad (same order as epi_n4_values); rv = residual values or zeros.
  // MODE 0: bias (+ masks, residual) only; 1: + ReLU; 2: anything (alpha, clamp, GELU/SiLU) -- chosen once per launch.
  const bool min_[2] = {masked[0] && p.mask_mode == APE_MASK_ZERO_INPUT, masked[1] && p.mask_mode == APE_MASK_ZERO_INPUT};
  const bool mout_[2] = {masked[0] && p.mask_mode == APE_MASK_ZERO_OUTPUT, masked[1] && p.mask_mode == APE_MASK_ZERO_OUTPUT};
  const int mode = (p.alpha == 1.f && p.clamp <= 0.f && (p.act == APE_ACT_NONE || p.act == APE_ACT_RELU)) ? (p.act == APE_ACT_RELU ? 1 : 0) : 2;
  auto finish = [&](auto mode_tag, f32x4_t a, const float4 b, const float rv[4], int mi, float v[4]) {
    constexpr int MODE = decltype(mode_tag)::value;
    const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = a[r];
      if (MODE == 2) x *= p.alpha;
      x = min_[mi] ? 0.f : x;
      x += bb[r];
      if (MODE == 1) x = fmaxf(x, 0.f);
      if (MODE == 2) {
        x = act_fn(x, p.act);
        if (p.clamp > 0.f) x = fminf(fmaxf(x, -p.clamp), p.clamp);
      }
      if (RES != 0) x += rv[r];
      x = mout_[mi] ? 0.f : x;
      v[r] = x;
    }
  };
  auto epilogue_fast = [&](auto mode_tag, int cc) {        // whole chunk in bounds: unconditional 16-byte stores, NE per lane
    const int ncol = cc * KR_CH + fq * 8;   // relative to nbeg; piece jp covers columns ncol + 32 jp .. + 7
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      unsigned char* cp = reinterpret_cast<unsigned char*>(p.C) + ((size_t)mrow[mi] * p.ldc + nbeg + ncol) * (OUT_F32 ? 4 : 2);
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        float v[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = jp * 2 + h;
          const float4 b = *reinterpret_cast<const float4*>(sbias + ncol + jp * 32 + h * 4);
          float rv[4] = {0.f, 0.f, 0.f, 0.f};
          if (RES == 1) {
            const uint4 u = rres[RES == 0 ? 0 : mi][jp];
            const uint32_t lo = h == 0 ? u.x : u.z, hi = h == 0 ? u.y : u.w;
            unpack2<H>(lo, rv[0], rv[1]);
            unpack2<H>(hi, rv[2], rv[3]);
          }
          finish(mode_tag, acc[mi][j], b, rv, mi, v[h]);
        }
        if (OUT_F32) {
          *reinterpret_cast<float4*>(cp + jp * 128) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
          *reinterpret_cast<float4*>(cp + jp * 128 + 16) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
        } else {
          *reinterpret_cast<uint4*>(cp + jp * 64) = make_uint4(h16<HO>::pack2(v[0][0], v[0][1]), h16<HO>::pack2(v[0][2], v[0][3]),
                                                               h16<HO>::pack2(v[1][0], v[1][1]), h16<HO>::pack2(v[1][2], v[1][3]));
        }
        __builtin_amdgcn_sched_barrier(0);    // one 8-column piece at a time: keeps the live set small (no spills)
      }
    }
  };
  auto epilogue_edge = [&](int cc) {        // bounds-checked (last row block / partial last chunk), 4 columns at a time
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int m = mrow[mi], ncol = cc * KR_CH + (j >> 1) * 32 + fq * 8 + (j & 1) * 4, n = nbeg + ncol;
        if (m >= p.M || n >= p.N) continue;          // N % 4 == 0 (launcher), so a quad is all in or all out
        const float4 b = *reinterpret_cast<const float4*>(sbias + ncol);
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (RES == 1) ld4<H>(reinterpret_cast<const H*>(p.residual) + (size_t)m * p.ldr + n, rv);
        float v[4];
        finish(std::integral_constant<int, 2>{}, acc[mi][j], b, rv, mi, v);
        if (OUT_F32) st4<float>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, v);
        else st4<HO>(reinterpret_cast<HO*>(p.C) + (size_t)m * p.ldc + n, v);
      }
  };

  issue(0); issue(1); issue(2);
  bool prev_fast = false;                    // previous chunk took the fast epilogue (its NR + NE ops are in the queue)
  for (int cc = 0; cc < nch; ++cc) {
    const int t0 = cc * 4;
    const bool fast = rows_full && nbeg + (cc + 1) * KR_CH <= p.N;
    // ks = 0: ops issued after stage t0's request: [t0+1] [R] [t0+2] [E]
    if (prev_fast) wait_vmcnt<8 + NR + NE>(); else wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    issue(t0 + 3);
    compute(t0, 0);
    // ks = 1: after stage t0+1's request: [R] [t0+2] [E] [t0+3]
    if (prev_fast) wait_vmcnt<8 + NR + NE>(); else wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    issue(t0 + 4);
    compute(t0 + 1, 1);
    // ks = 2: after stage t0+2's request: [E] [t0+3] [t0+4]
    if (prev_fast) wait_vmcnt<8 + NE>(); else wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    issue(t0 + 5);
    compute(t0 + 2, 2);
    // ks = 3: after stage t0+3's request: [t0+4] [t0+5]
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    if (fast) load_res(nbeg + cc * KR_CH + fq * 8);
    issue(t0 + 6);
    compute(t0 + 3, 3);
    if (fast) {
      if (mode == 0) epilogue_fast(std::integral_constant<int, 0>{}, cc);
      else if (mode == 1) epilogue_fast(std::integral_constant<int, 1>{}, cc);
      else epilogue_fast(std::integral_constant<int, 2>{}, cc);
    } else {
      epilogue_edge(cc);
      wait_vmcnt<0>();                        // unknown number of predicated ops: drain, the next chunk counts from zero
    }
    prev_fast = fast;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[mi][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
}

// ------------------------------------------------------------------------------------------
// "kres" with the LayerNorm that follows the linear in its epilogue:  y = LN(x W^T + b + residual) * gamma + beta  for K = N = 256
// (detrex BaseTransformerLayer: the deformable attention's output projection + identity, then "norm";
// ape/layers/multi_scale_deform_attn.py:353-358 + the layer's norms[0]).  A workgroup still owns 128 rows; a wave keeps the
// accumulators of BOTH 128-column chunks (2 x 64 registers), so a row's 256 channels sit in the 4 lanes frow + 16 fq and the
// statistics are two lane swaps (as in ffn_fused.hip) -- on the fp32 sums, without the 16-bit rounding a separate LayerNorm launch
// would read back, and without its 89 MB round trip per encoder layer.  The residual rows are fetched after the last MFMA (the
// A fragments are dead by then: no registers to hold them earlier); the second workgroup of the CU covers that latency.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float kr_rows_sum(float v) {      // sum over the lanes l, l^16, l^32, l^48
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned w = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <typename H>
__global__ __launch_bounds__(256, 2) void gemm_kres_ln_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* ring = reinterpret_cast<bf16_t*>(smem_raw);                     // 4 stages x [128][64] (swz128 image)
  float* sbias = reinterpret_cast<float*>(smem_raw + 4 * 16384);          // bias | gamma | beta, 256 floats each
  float* sgam = sbias + 256;
  float* sbet = sgam + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  sbias[tid] = p.bias != nullptr ? p.bias[tid] : 0.f;
  sgam[tid] = p.ln_w[tid];
  sbet[tid] = p.ln_b[tid];
  __syncthreads();

  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);
  const int m_wave = blockIdx.x * KR_BM + wave * 32;
  bf16x8_t af[2][8];
  int mrow[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m_wave + mi * 16 + frow;
    mrow[mi] = m;
    const bf16_t* ap = A + (size_t)(m < p.M ? m : p.M - 1) * p.lda + fq * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) af[mi][s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(ap + s * 32));
  }
  uint32_t woff[4];                                                      // element offset of this lane's 16-byte piece inside a W chunk
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rho = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((rho >> 1) & 7);
    const int idx = rho & 15;
    const int wcol = (rho >> 5) * 32 + (idx >> 2) * 8 + ((rho >> 4) & 1) * 4 + (idx & 3);     // same row permutation as gemm_bf16_kres_kernel
    woff[i] = (uint32_t)wcol * (uint32_t)p.ldw + (uint32_t)c * 8u;
  }
  constexpr int T = 8;                                                   // 2 chunks x 4 k steps
  auto issue = [&](int t) {
    t = t < T ? t : T - 1;
    const int cc = t >> 2, ks = t & 3;
    bf16_t* st = ring + (t & 3) * 8192;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(W + (size_t)cc * KR_CH * p.ldw + ks * 64 + woff[i]), (lds_void_t*)(st + (wave * 4 + i) * 512), 16, 0, 0);
  };
  f32x4_t acc[2][2][8];                                                  // [chunk][row tile][column tile]
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[cc][mi][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int t, int cc, int ks) __attribute__((always_inline)) {
    const bf16_t* st = ring + (t & 3) * 8192;
    auto rd = [&](int h, bf16x8_t (&wf)[4]) {
      const int kk = h >> 1, jb = (h & 1) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rho = (jb + j) * 16 + frow;
        wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(st + rho * 64 + (((kk * 4 + fq) ^ ((rho >> 1) & 7)) << 3)));
      }
    };
    auto mm = [&](int h, bf16x8_t (&wf)[4]) {
      const int kk = h >> 1, jb = (h & 1) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[cc][mi][jb + j] = h16<H>::mfma(wf[j], af[mi][ks * 2 + kk], acc[cc][mi][jb + j]);
    };
    bf16x8_t w0[4], w1[4];
    rd(0, w0);
    rd(1, w1);
    __builtin_amdgcn_sched_barrier(0);
    mm(0, w0);
    __builtin_amdgcn_sched_barrier(0);
    rd(2, w0);
    mm(1, w1);
    __builtin_amdgcn_sched_barrier(0);
    rd(3, w1);
    mm(2, w0);
    __builtin_amdgcn_sched_barrier(0);
    mm(3, w1);
  };
  issue(0); issue(1); issue(2);
#pragma unroll
  for (int t = 0; t < T; ++t) {          // stage t's request is followed by the requests of t+1, t+2 (8 LDS-DMAs) when it is awaited
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    issue(t + 3);
    compute(t, t >> 2, t & 3);
  }
  // ---- epilogue: + bias + residual, LayerNorm over the row's 256 channels, 16-byte stores (64 contiguous bytes per row and instruction)
  // ONE wave-uniform branch on the residual (round 6).  Tested per load, each of the 16 residual pieces of a wave sat in a basic block
  // of its own, and the register allocator carried three finished accumulators through that chain in scratch: the 56-76 B per lane
  // this kernel had since round 4 (a dispatch with any scratch is throttled on this chip).  Same arithmetic, 250 / 235 registers, 0 B.
  auto epilogue = [&](auto res_t) __attribute__((always_inline)) {
    constexpr bool RES = decltype(res_t)::value;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = mrow[mi] < p.M ? mrow[mi] : p.M - 1;                   // clamped: the lane swaps need every lane
      float v[64];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          const int col = cc * KR_CH + jp * 32 + fq * 8;
          float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (RES) ld8<H>(reinterpret_cast<const H*>(p.residual) + (size_t)m * p.ldr + col, rv);
          const float4 b0 = *reinterpret_cast<const float4*>(sbias + col), b1 = *reinterpret_cast<const float4*>(sbias + col + 4);
          const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[cc * 32 + jp * 8 + h * 4 + r] = acc[cc][mi][jp * 2 + h][r] + bb[h * 4 + r] + rv[h * 4 + r];
        }
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 64; ++e) sum += v[e];
      const float mean = kr_rows_sum(sum) * (1.f / 256.f);
      float d2 = 0.f;
#pragma unroll
      for (int e = 0; e < 64; ++e) { v[e] -= mean; d2 = fmaf(v[e], v[e], d2); }
      const float rstd = rsqrtf(kr_rows_sum(d2) * (1.f / 256.f) + p.ln_eps);
      if (mrow[mi] < p.M) {
        H* yp = reinterpret_cast<H*>(p.C) + (size_t)m * p.ldc;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int jp = 0; jp < 4; ++jp) {
            const int col = cc * KR_CH + jp * 32 + fq * 8;
            const float4 g0 = *reinterpret_cast<const float4*>(sgam + col), g1 = *reinterpret_cast<const float4*>(sgam + col + 4);
            const float4 e0 = *reinterpret_cast<const float4*>(sbet + col), e1 = *reinterpret_cast<const float4*>(sbet + col + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, ee[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = fmaf(v[cc * 32 + jp * 8 + q] * rstd, gg[q], ee[q]);
            st8<H>(yp + col, o);
          }
      }
      __builtin_amdgcn_sched_barrier(0);      // one row tile at a time (both interleaved do not fit the register file)
    }
  };
  if (p.residual != nullptr) epilogue(std::true_type{}); else epilogue(std::false_type{});
}

// ------------------------------------------------------------------------------------------
// C-ABI launchers
// ------------------------------------------------------------------------------------------
static thread_local const char* g_last_gemm_kernel = "";
extern "C" const char* ape_hip_gemm_last_kernel(void) { return g_last_gemm_kernel; }
#define LAUNCH_GEMM(NAME, KERNEL, GRID, LDS) do { g_last_gemm_kernel = NAME; APE_LAUNCH(KERNEL, GRID, dim3(256), LDS, s, p); } while (0)

// 16-bit operand launches (H = bf16_t | f16_t): tile selection is the same for both flavours
template <typename H>
static int gemm_launch_h16(ApeGemmArgs& p, hipStream_t s) {
  APE_CHECK_ARG(p.K % 8 == 0, "ape_hip_gemm(bf16): K=%d must be a multiple of 8 (pad the operands)", p.K);
  APE_CHECK_ARG(p.lda % 8 == 0 && p.ldw % 8 == 0, "ape_hip_gemm(bf16): lda/ldw must be multiples of 8");
  APE_CHECK_ARG(((uintptr_t)p.A) % 16 == 0 && ((uintptr_t)p.W) % 16 == 0, "ape_hip_gemm(bf16): A/W must be 16-byte aligned");
  const int nblk = ceil_div(p.M, GB_M) * ceil_div(p.N, GB_N);
  const int nblk64 = ceil_div(p.M, 64) * ceil_div(p.N, 64);
  static const int force_v1 = getenv("APE_GEMM_V1") ? atoi(getenv("APE_GEMM_V1")) : 0;
  static const int no_glds = getenv("APE_GEMM_NOGLDS") ? atoi(getenv("APE_GEMM_NOGLDS")) : 0;
  const int esz = p.out_dt == APE_DT_F32 ? 4 : 2;
  // v2's LDS-staged epilogue writes 16-byte chunks of output rows
  const bool v2_ok = !force_v1 && ((size_t)p.ldc * esz) % 16 == 0 && ((uintptr_t)p.C) % 16 == 0 &&
                     (p.act != APE_ACT_SWIGLU || p.N % 4 == 0);
  const bool glds = v2_ok && !no_glds && p.K % GB_K == 0;
  static const int no_ring = getenv("APE_GEMM_NORING") ? atoi(getenv("APE_GEMM_NORING")) : 0;
  const char* ring_env = getenv("APE_GEMM_RING");     // read per call so a probe can flip it
  const int use_ring_always = ring_env ? atoi(ring_env) : 0;
  const bool ring = v2_ok && !no_glds && !no_ring && p.K % GR_K == 0;
  if (p.conv_h > 0) {
    // implicit-GEMM 3 x 3 convolution: the eight-wave tile kernel only (callers fall back to im2col + gemm when this is refused): 256 x 256
    // tiles when they fill the chip, else 256 x 128 tiles when there are at least 100 of them (the 128 x 128-pixel p3 map: 128)
    const int t256 = ceil_div(p.M, 256) * ceil_div(p.N, 256), t128 = ceil_div(p.M, 256) * ceil_div(p.N, 128);
    APE_CHECK_ARG(p.conv_w > 0 && p.splitk <= 1 && !p.trans_out && (t256 >= 200 || t128 >= 100),
                  "ape_hip_gemm(conv3x3): needs >= 200 tiles of 256 x 256 or >= 100 of 256 x 128 (M = %d, N = %d), no split-K / transposed output", p.M, p.N);
    const char* name = ape_gemm_p8_launch(p, t256 >= 200 ? 256 : 128, 1, s);
    APE_CHECK_ARG(name != nullptr, "ape_hip_gemm(conv3x3): K = 9 * 256 channels of 16-bit operands, lda >= 256, M == conv_h * conv_w, a 16-byte aligned zero row");
    g_last_gemm_kernel = name;
    return 0;
  }
  if (p.tile64 == 3 || p.tile64 == 4) {
    // 256 x 256 / 256 x 128 eight-wave tiles with the counted-wait pipeline (gemm_p8.hip); falls back when unsupported
    const char* st_env = getenv("APE_GEMM_P8_STAGGER");
    const int stagger = st_env ? atoi(st_env) : 1;
    const char* name = ape_gemm_p8_launch(p, p.tile64 == 3 ? 256 : 128, stagger, s);
    if (name != nullptr) {
      g_last_gemm_kernel = name;
      return 0;
    }
    APE_CHECK_ARG(p.rowstat_cols == 0, "ape_hip_gemm: rowstat_cols needs the 256 x 128 tile kernel's plain epilogue (K %% 64 == 0, N %% 128 == 0, alpha 1, "
                                       "no mask / clamp / activation / RoPE, 16-byte aligned bias / colvec / residual rows)");
    p.tile64 = 0;
  }
  constexpr bool HF = h16<H>::dt == APE_DT_F16;
  APE_CHECK_ARG(HF || v2_ok || p.out_dt != APE_DT_F16, "ape_hip_gemm: f16 output needs 16-byte aligned output rows");
  if (v2_ok) {
    static ApeOncePerDevice attr_done;
    if (attr_done.first()) {  // > 64 KiB of dynamic LDS needs the opt-in attribute
      (void)hipFuncSetAttribute((const void*)gemm_bf16_v2_kernel<true, true, H>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_V2_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_bf16_v2_kernel<true, false, H>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_V2_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_bf16_v2_kernel<false, true, H>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_V2_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_bf16_v2_kernel<false, false, H>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_V2_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_bf16_ring_kernel<true, 4, 4, H>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_V2_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_bf16_ring_kernel<false, 4, 4, H>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_V2_LDS);
    }
    if (p.ln_w != nullptr) {
      // LayerNorm in the epilogue: the K = N = 256 register-resident kernel only (the callers check the same conditions and
      // launch a separate LayerNorm otherwise)
      APE_CHECK_ARG(p.ln_b != nullptr && p.K == 256 && p.N == 256 && p.M >= 2048 && !p.trans_out && p.act == APE_ACT_NONE && p.rope_cos == nullptr &&
                        p.rowscale == nullptr && p.rowmask == nullptr && p.splitk <= 1 && p.alpha == 1.f && p.clamp <= 0.f && p.out_dt == h16<H>::dt &&
                        p.ldc % 8 == 0 && ((uintptr_t)p.C) % 16 == 0 && p.K % GB_K == 0 && ((uintptr_t)p.ln_w) % 16 == 0 && ((uintptr_t)p.ln_b) % 16 == 0 &&
                        (p.residual == nullptr || (p.res_dt == h16<H>::dt && p.ldr % 8 == 0 && ((uintptr_t)p.residual) % 16 == 0)),
                    "ape_hip_gemm: an epilogue LayerNorm (ln_w) needs K == N == 256, M >= 2048, 16-bit operands / residual / output of one type, a plain epilogue");
      static ApeOncePerDevice lattr;
      if (lattr.first()) { (void)hipFuncSetAttribute((const void*)gemm_kres_ln_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, KR_LDS); }
      LAUNCH_GEMM("gemm_kres_ln_kernel", (gemm_kres_ln_kernel<H>), dim3(ceil_div(p.M, KR_BM)), KR_LDS);
      return 0;
    }
    const char* nk_env = getenv("APE_GEMM_NOKRES");     // read per call so a probe can flip it
    const int no_kres = nk_env ? atoi(nk_env) : 0;
    const bool kres = !no_kres && !no_glds && p.K == 256 && p.rowscale == nullptr && p.M >= 2048 && p.N >= 64 && p.N % 8 == 0 && !p.trans_out &&
                      p.act != APE_ACT_SWIGLU && p.rope_cos == nullptr && p.splitk <= 1 && p.ldc % 8 == 0 &&
                      (p.residual == nullptr || (p.res_dt == h16<H>::dt && p.ldr % 8 == 0 && ((uintptr_t)p.residual) % 16 == 0));
    APE_CHECK_ARG(HF || kres || p.out_dt != APE_DT_F16, "ape_hip_gemm: f16 output is only produced by the K == 256 kernel (disabled by the environment?)");
    if (kres) {
      static ApeOncePerDevice kattr;
      if (kattr.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kres_kernel<0, false, false, H>, hipFuncAttributeMaxDynamicSharedMemorySize, KR_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kres_kernel<1, false, false, H>, hipFuncAttributeMaxDynamicSharedMemorySize, KR_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kres_kernel<0, true, false, H>, hipFuncAttributeMaxDynamicSharedMemorySize, KR_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kres_kernel<1, true, false, H>, hipFuncAttributeMaxDynamicSharedMemorySize, KR_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kres_kernel<0, false, true, H>, hipFuncAttributeMaxDynamicSharedMemorySize, KR_LDS);
      }
      const int mblk = ceil_div(p.M, KR_BM);
      const int nch = ceil_div(p.N, KR_CH);
      int ysplit = ceil_div(nch, KR_MAXN / KR_CH);                  // bias slab in LDS holds KR_MAXN columns
      while (mblk * ysplit < 512 && ysplit * 2 <= nch) ysplit *= 2;   // few row blocks: spread the column chunks as well
      dim3 grid(mblk, ysplit);
      // tail split (see the kernel): the row blocks of a last round that would fill less than half of the 512 resident workgroups are
      // cut into column parts so that the round is full and short.  APE_KRES_TAILSPLIT=0 restores the plain grid.
      p.reserved0 = 0;
      {
        const char* te = getenv("APE_KRES_TAILSPLIT");
        const int slots = 2 * ape_cu_count();                       // 2 workgroups per CU (512 on the 256 CUs of an MI355X)
        const int nfull = (mblk / slots) * slots, tail = mblk - nfull;
        // from 8 column chunks on (measured, profiles/r05_kres_probe.log: 1536 columns 120 -> 109 us, 2048: 154 -> 142; with 2 - 4 chunks the
        // extra copies of the A rows cost more than the shorter last round gains: 480 columns 52.7 -> 56.1 us, 256: 31.8 -> 32.9)
        if (!(te != nullptr && atoi(te) == 0) && ysplit == 1 && nfull > 0 && nfull < 65536 && tail > 0 && tail * 2 <= slots && nch >= 8) {
          const int parts = min(nch, slots / tail);
          if (parts >= 2) {
            p.reserved0 = nfull | (parts << 16);
            grid = dim3(nfull + tail * parts, 1);
          }
        }
      }
      const bool res = p.residual != nullptr;          // bf16 residual only (an fp32 one does not fit the register budget)
      const bool of32 = p.out_dt == APE_DT_F32;
      if (!HF && p.out_dt == APE_DT_F16) LAUNCH_GEMM("gemm_bf16_kres_kernel<0, false, true>", (gemm_bf16_kres_kernel<0, false, true, H>), grid, KR_LDS);
      else if (!res && !of32) LAUNCH_GEMM("gemm_bf16_kres_kernel<0, false>", (gemm_bf16_kres_kernel<0, false, false, H>), grid, KR_LDS);
      else if (res && !of32) LAUNCH_GEMM("gemm_bf16_kres_kernel<1, false>", (gemm_bf16_kres_kernel<1, false, false, H>), grid, KR_LDS);
      else if (!res) LAUNCH_GEMM("gemm_bf16_kres_kernel<0, true>", (gemm_bf16_kres_kernel<0, true, false, H>), grid, KR_LDS);
      else LAUNCH_GEMM("gemm_bf16_kres_kernel<1, true>", (gemm_bf16_kres_kernel<1, true, false, H>), grid, KR_LDS);
    } else if (p.splitk > 1) {
      APE_CHECK_ARG(ring && !p.trans_out && p.workspace != nullptr && p.act != APE_ACT_SWIGLU,
                    "ape_hip_gemm: split-K needs bf16, K %% 32 == 0, no trans_out / SwiGLU, and a workspace");
      APE_CHECK_ARG(p.N % 4 == 0 && ((uintptr_t)p.workspace) % 16 == 0, "ape_hip_gemm: split-K needs N %% 4 == 0 and an aligned workspace");
      if (p.tile64 == 2) LAUNCH_GEMM("gemm_bf16_ring_kernel<false, 4, 2>", (gemm_bf16_ring_kernel<false, 4, 2, H>), dim3(ceil_div(p.M, 128) * ceil_div(p.N, 64), p.splitk), GEMM_T128x64_LDS);
      else if (p.tile64) LAUNCH_GEMM("gemm_bf16_ring_kernel<false, 2, 2>", (gemm_bf16_ring_kernel<false, 2, 2, H>), dim3(nblk64, p.splitk), GEMM_T64_LDS);
      else LAUNCH_GEMM("gemm_bf16_ring_kernel<false, 4, 4>", (gemm_bf16_ring_kernel<false, 4, 4, H>), dim3(nblk, p.splitk), GEMM_V2_LDS);
      const size_t groups = (size_t)p.M * ((p.N + 3) / 4);
      APE_LAUNCH(gemm_splitk_reduce_kernel<H>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, p);
    } else if (ring && p.tile64 == 2 && !p.trans_out) {
      // 128 x 64 tiles: twice the workgroups of a 128 x 128 tiling at 3/4 of its operand traffic per flop
      const int nblk_mn = ceil_div(p.M, 128) * ceil_div(p.N, 64);
      LAUNCH_GEMM("gemm_bf16_ring_kernel<false, 4, 2>", (gemm_bf16_ring_kernel<false, 4, 2, H>), dim3(nblk_mn), GEMM_T128x64_LDS);
    } else if (ring && p.tile64) {
      if (p.trans_out) LAUNCH_GEMM("gemm_bf16_ring_kernel<true, 2, 2>", (gemm_bf16_ring_kernel<true, 2, 2, H>), dim3(nblk64), GEMM_T64_LDS);
      else LAUNCH_GEMM("gemm_bf16_ring_kernel<false, 2, 2>", (gemm_bf16_ring_kernel<false, 2, 2, H>), dim3(nblk64), GEMM_T64_LDS);
    } else if (ring && (use_ring_always || nblk < 64)) {
      if (p.trans_out) LAUNCH_GEMM("gemm_bf16_ring_kernel<true, 4, 4>", (gemm_bf16_ring_kernel<true, 4, 4, H>), dim3(nblk), GEMM_V2_LDS);
      else LAUNCH_GEMM("gemm_bf16_ring_kernel<false, 4, 4>", (gemm_bf16_ring_kernel<false, 4, 4, H>), dim3(nblk), GEMM_V2_LDS);
    } else if (p.trans_out) {
      if (glds) LAUNCH_GEMM("gemm_bf16_v2_kernel<true, true>", (gemm_bf16_v2_kernel<true, true, H>), dim3(nblk), GEMM_V2_LDS);
      else LAUNCH_GEMM("gemm_bf16_v2_kernel<true, false>", (gemm_bf16_v2_kernel<true, false, H>), dim3(nblk), GEMM_V2_LDS);
    } else {
      if (glds) LAUNCH_GEMM("gemm_bf16_v2_kernel<false, true>", (gemm_bf16_v2_kernel<false, true, H>), dim3(nblk), GEMM_V2_LDS);
      else LAUNCH_GEMM("gemm_bf16_v2_kernel<false, false>", (gemm_bf16_v2_kernel<false, false, H>), dim3(nblk), GEMM_V2_LDS);
    }
  } else if (p.trans_out) LAUNCH_GEMM("gemm_bf16_kernel<true>", (gemm_bf16_kernel<true, H>), dim3(nblk), 0);
  else LAUNCH_GEMM("gemm_bf16_kernel<false>", (gemm_bf16_kernel<false, H>), dim3(nblk), 0);
  return 0;
}

extern "C" int ape_hip_gemm(const ApeGemmArgs* a, void* stream) {
  APE_CHECK_ARG(a != nullptr, "ape_hip_gemm: null args");
  ApeGemmArgs p = *a;
  // internal hint field of the library's OWN copy (bit 30: p8 residual prefetch off; bits 0-15 | 16-29: kres tail split `nfull | parts << 16`);
  // whatever a caller left there is dropped here, once, for every kernel behind this entry
  p.reserved0 = 0;
  APE_CHECK_ARG(p.A && p.W && p.C, "ape_hip_gemm: null pointer");
  APE_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "ape_hip_gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
  APE_CHECK_ARG(p.in_dt == APE_DT_F32 || ape_is16(p.in_dt), "ape_hip_gemm: bad in_dt %d", p.in_dt);
  // one 16-bit storage type per launch (operands, residual, output); the single exception is the bf16 flavour's IEEE-half
  // output of the K = 256 kernel checked below
  APE_CHECK_ARG(p.residual == nullptr || p.res_dt == APE_DT_F32 || (ape_is16(p.res_dt) && (p.in_dt == APE_DT_F32 || p.res_dt == p.in_dt)),
                "ape_hip_gemm: a 16-bit residual must have the operands' 16-bit type (in_dt %d, res_dt %d)", p.in_dt, p.res_dt);
  APE_CHECK_ARG(p.in_dt != APE_DT_F16 || p.out_dt != APE_DT_BF16, "ape_hip_gemm: f16 operands produce f32 or f16 outputs");
  APE_CHECK_ARG(p.in_dt != APE_DT_F32 || p.residual == nullptr || !ape_is16(p.res_dt) || !ape_is16(p.out_dt) || p.res_dt == p.out_dt,
                "ape_hip_gemm: 16-bit residual and output of an f32 launch must share a type");
  APE_CHECK_ARG(p.out_dt == APE_DT_F32 || p.out_dt == APE_DT_BF16 || p.out_dt == APE_DT_F16, "ape_hip_gemm: bad out_dt %d", p.out_dt);
  // IEEE-half output exists for one kernel: the K = 256 register-resident GEMM without residual (the deformable attention's
  // offset / logit projection over the encoder tokens, which is bound by the bytes it writes)
  APE_CHECK_ARG(p.out_dt != APE_DT_F16 || p.in_dt != APE_DT_BF16 || (p.K == 256 && p.M >= 2048 && p.N >= 64 && p.N % 8 == 0 && !p.trans_out &&
                                           p.residual == nullptr && p.rowscale == nullptr && p.act != APE_ACT_SWIGLU && p.rope_cos == nullptr &&
                                           p.splitk <= 1 && p.ldc % 8 == 0 && ((uintptr_t)p.C) % 16 == 0 && p.tile64 != 3 && p.tile64 != 4),
                "ape_hip_gemm: f16 output needs bf16 inputs, K == 256, M >= 2048, N %% 8 == 0, no residual / transpose / rope / split-K");
  APE_CHECK_ARG(p.act != APE_ACT_SWIGLU || (p.N % 4 == 0 && !p.trans_out), "ape_hip_gemm: swiglu needs N%%4==0, no transpose");
  APE_CHECK_ARG(!p.trans_out || (p.residual == nullptr && p.rope_cos == nullptr && p.rowmask == nullptr),
                "ape_hip_gemm: transposed output supports bias/act only");
  APE_CHECK_ARG(p.splitk <= 1 || ape_is16(p.in_dt), "ape_hip_gemm: split-K is implemented for the 16-bit kernels only");
  APE_CHECK_ARG(p.rope_cos == nullptr || (p.rope_sin != nullptr && p.rope_rows > 0 && p.rope_hd > 0 && p.rope_hd % 4 == 0),
                "ape_hip_gemm: bad rope args");
  APE_CHECK_ARG((p.rowscale == nullptr) == (p.rowshift == nullptr) && ((p.rowscale == nullptr) == (p.colvec == nullptr) || (p.rowstat_cols > 0 && p.rowscale == nullptr)),
                "ape_hip_gemm: rowscale / rowshift / colvec go together (colvec alone with rowstat_cols > 0)");
  APE_CHECK_ARG(p.rowscale == nullptr || (!p.trans_out && p.act != APE_ACT_SWIGLU), "ape_hip_gemm: folded LayerNorm needs a plain (non-transposed, non-SwiGLU) epilogue");
  APE_CHECK_ARG(p.rowstat_cols >= 0 && (p.rowstat_cols == 0 || (p.rowscale == nullptr && p.colvec != nullptr && ape_is16(p.in_dt) && p.tile64 == 4 && !p.trans_out &&
                                                                  p.splitk <= 1 && p.ln_w == nullptr && p.conv_h <= 0)),
                "ape_hip_gemm: rowstat_cols (in-launch row statistics) needs colvec without rowscale / rowshift, 16-bit operands and tile64 == 4");
  const int esz_out = p.out_dt == APE_DT_F32 ? 4 : 2;
  const int esz_res = p.res_dt == APE_DT_F32 ? 4 : 2;
  int vec = (p.ldc % 4 == 0) && (((uintptr_t)p.C) % 16 == 0);
  if (p.residual) vec = vec && (p.ldr % 4 == 0) && (((uintptr_t)p.residual) % 16 == 0);
  (void)esz_out; (void)esz_res;
  if (p.bias != nullptr && ((uintptr_t)p.bias) % 16 == 0) vec |= 2;
  if (p.colvec != nullptr && ((uintptr_t)p.colvec) % 16 == 0) vec |= 8;
  if (p.rope_cos != nullptr && ((uintptr_t)p.rope_cos) % 16 == 0 && ((uintptr_t)p.rope_sin) % 16 == 0 &&
      (p.rope_hd & (p.rope_hd - 1)) == 0 && (p.rope_rows >= p.M || (p.rope_rows & (p.rope_rows - 1)) == 0)) vec |= 4;
  p.vec_ok = vec;
  hipStream_t s = (hipStream_t)stream;
  APE_CHECK_ARG(p.ln_w == nullptr || ape_is16(p.in_dt), "ape_hip_gemm: the epilogue LayerNorm exists for the 16-bit K = N = 256 kernel only");
  APE_CHECK_ARG(p.conv_h <= 0 || ape_is16(p.in_dt), "ape_hip_gemm(conv3x3): implicit convolution exists for the 16-bit tile kernel only (use ape_hip_im2col3x3 in fp32)");
  if (ape_is16(p.in_dt)) {
    if (p.ln_w != nullptr && (p.tile64 == 3 || p.tile64 == 4)) p.tile64 = 0;
    const int rc = p.in_dt == APE_DT_F16 ? gemm_launch_h16<f16_t>(p, s) : gemm_launch_h16<bf16_t>(p, s);
    if (rc != 0) return rc;
    if (p.in_dt == APE_DT_F16) {             // the f16 instantiations report as gemm_f16_*
      static thread_local char nm[96];
      const char* src = g_last_gemm_kernel;
      if (strncmp(src, "gemm_bf16_", 10) == 0) { snprintf(nm, sizeof(nm), "gemm_f16_%s", src + 10); g_last_gemm_kernel = nm; }
    }
  } else {
    const int nblk = ceil_div(p.M, 64) * ceil_div(p.N, 64);
    const bool hf = p.out_dt == APE_DT_F16 || (p.residual != nullptr && p.res_dt == APE_DT_F16);
    if (p.trans_out) { if (hf) LAUNCH_GEMM("gemm_f32_kernel<true>", (gemm_f32_kernel<true, f16_t>), dim3(nblk), 0); else LAUNCH_GEMM("gemm_f32_kernel<true>", (gemm_f32_kernel<true, bf16_t>), dim3(nblk), 0); }
    else { if (hf) LAUNCH_GEMM("gemm_f32_kernel<false>", (gemm_f32_kernel<false, f16_t>), dim3(nblk), 0); else LAUNCH_GEMM("gemm_f32_kernel<false>", (gemm_f32_kernel<false, bf16_t>), dim3(nblk), 0); }
  }
  APE_CHECK_LAUNCH("ape_hip_gemm");
  return 0;
}

static int launch_gemv(const float* x, int ldx, const void* W, int ldw, int w_dt, const float* bias, float* out, int ldo, int M,
                       int N, int K, float alpha, const float* scale, const float* add, int ldadd, float* out2, int ldo2,
                       void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nblk = ceil_div(N, 4);
  if (w_dt == APE_DT_F16)
    APE_LAUNCH(gemv_kernel<f16_t>, dim3(nblk), dim3(256), 0, s, x, ldx, (const f16_t*)W, ldw, bias, out, ldo, M, N, K, alpha,
                       scale, add, ldadd, out2, ldo2);
  else if (w_dt == APE_DT_BF16)
    APE_LAUNCH(gemv_kernel<bf16_t>, dim3(nblk), dim3(256), 0, s, x, ldx, (const bf16_t*)W, ldw, bias, out, ldo, M, N, K, alpha,
                       scale, add, ldadd, out2, ldo2);
  else
    APE_LAUNCH(gemv_kernel<float>, dim3(nblk), dim3(256), 0, s, x, ldx, (const float*)W, ldw, bias, out, ldo, M, N, K, alpha,
                       scale, add, ldadd, out2, ldo2);
  return 0;
}

extern "C" int ape_hip_gemv(const float* x, int ldx, const void* W, int ldw, int w_dt, const float* bias, float* out,
                            int ldo, int M, int N, int K, float alpha, void* stream) {
  APE_CHECK_ARG(x && W && out && M > 0 && N > 0 && K > 0, "ape_hip_gemv: bad args");
  launch_gemv(x, ldx, W, ldw, w_dt, bias, out, ldo, M, N, K, alpha, nullptr, nullptr, 0, nullptr, 0, stream);
  APE_CHECK_LAUNCH("ape_hip_gemv");
  return 0;
}

extern "C" int ape_hip_gemv_affine(const float* x, int ldx, const void* W, int ldw, int w_dt, const float* bias, float* out,
                                   int ldo, int M, int N, int K, float alpha, const float* scale, const float* add, int ldadd,
                                   float* out2, int ldo2, void* stream) {
  APE_CHECK_ARG(x && W && out && M > 0 && N > 0 && K > 0 && ((add == nullptr) == (out2 == nullptr)), "ape_hip_gemv_affine: bad args");
  launch_gemv(x, ldx, W, ldw, w_dt, bias, out, ldo, M, N, K, alpha, scale, add, ldadd, out2, ldo2, stream);
  APE_CHECK_LAUNCH("ape_hip_gemv_affine");
  return 0;
}
