// ffn_fused.hip -- EXPERIMENTAL (opt-in: APE_FFN_FUSED=1; written at the end of round 2 without GPU time left to validate it;
// the layout design is validated at index level by tests/test_ffn_fused_layout.py, the kernel by
// tests/test_ops_gpu.py::test_ffn_fused under APE_TEST_EXPERIMENTAL=1).
//
// Encoder FFN of the deformable transformer (detrex FFN, ape/modeling/ape_deta/deformable_transformer_vl.py:45-54) in ONE kernel:
//     y = x + relu(x W1^T + b1) W2^T + b2         x [M, 256] bf16, W1 [HID, 256], W2 [256, HID], HID = 2048, M = 87 296
// The two-GEMM form writes the [M, HID] hidden activations (357 MB per layer) and reads them back: FFN1 is bound by that write
// (182 us), FFN2 by reading it (135 us).  Here the hidden tile never leaves the registers -- the flash-attention pattern of
// attention.hip with relu() in the place of softmax:
//   * a workgroup = 4 waves = 128 token rows; a wave keeps its 32 rows of x as 2 x 8 MFMA B-operand fragments (64 VGPRs, loaded
//     once) and its 32 x 256 output tile as 16 x 2 accumulators (128 VGPRs);
//   * loop over the hidden dimension in chunks of 64: W1[chunk, :] (32 KB) and W2[:, chunk] (32 KB) stream HBM/L2 -> LDS with
//     global_load_lds (two stages; the next chunk lands while this one is consumed);
//   * first MFMA  H^T[hidden][token] = W1c . x^T : after bias + ReLU, a lane's accumulator registers (token lane & 15, hidden
//     4 g .. 4 g + 3 of each 16-row tile, g = lane >> 4) ARE the B operand of the second MFMA  Y^T[channel][token] += W2c . H^T
//     under a permuted k order (e < 4 -> hidden (2kk) 16 + 4g + e, e >= 4 -> (2kk + 1) 16 + 4g + e - 4), which is applied
//     identically to the W2 fragment reads (two 8-byte halves) -- no LDS round trip, no cross-lane traffic for H; H is rounded to
//     bf16 exactly where the two-GEMM form rounds it;
//   * the W2 rows are permuted on their way into LDS (row ot*16 + m holds channel (m >> 2) * 64 + ot * 4 + (m & 3)) so that a lane
//     ends up with 64 CONSECUTIVE output channels of its token: bias, residual and the bf16 store work on 16-byte pieces.
//   * LDS images: W1 rows are 512 B with the 16-byte chunk XOR-ed by (row & 31), W2 rows 128 B with chunk ^ ((row >> 1) & 7); the
//     XOR is applied to the LDS-DMA source address (its destination is lane-linear) and to the fragment read address.
// Per layer: 183 GFLOP on the MFMA pipe, 44.7 MB read + 44.7 MB written instead of 2 x 357 MB more.
#include <stdlib.h>

#include "common.h"
#include "../../include/ape_hip.h"

typedef __attribute__((address_space(3))) void ff_lds_void_t;
typedef const __attribute__((address_space(1))) void ff_gbl_void_t;

#define FF_K 256
#define FF_HC 64
#define FF_N 256
#define FF_BM 128
#define FF_STAGE 65536   // W1 chunk (32 KB) + W2 chunk (32 KB)

struct FfnParams {
  const bf16_t* X; int ldx;
  const bf16_t* W1; int ldw1;
  const float* b1;
  const bf16_t* W2; int ldw2;
  const float* b2;
  const bf16_t* R; int ldr;      // residual (may be NULL)
  bf16_t* Y; int ldy;
  int M, HID;
  const float* ln_w; const float* ln_b; float ln_eps;    // optional LayerNorm over the 256 output channels (NULL: none)
};

// sum over the 4 lanes l, l^16, l^32, l^48 (the lanes that hold the four 64-channel quarters of one token's output row)
__device__ __forceinline__ float ff_rows_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned w = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// W2P: W2's hidden columns arrive PRE-PERMUTED inside every group of 32 (position 8g + e holds hidden 4g + e for e < 4 and
// 16 + 4g + e - 4 for e >= 4 -- the k order of the second MFMA's B operand above; ape_amd.packing.permute_ffn_w2): the W2 fragment
// of a lane is then ONE ds_read_b128 instead of two ds_read_b64 (which reach their LDS rate only from ~4 waves per SIMD).
// RT = 16-row token tiles per wave (2 | 3): a workgroup covers 64 RT rows.  The chunk time is set by the LDS-DMA bytes a CU can keep in
// flight (one 64 KB chunk ahead: ~2.3 us per chunk whatever the MFMA count, measured), so rows per workgroup is the lever: RT = 3
// does 1.5 x the MFMAs per streamed weight byte.  One wave per SIMD owns the whole 512-entry register file: RT = 3 uses ~460 of it.
// H = bf16_t | f16_t: the storage type of x / weights / residual / y and of the hidden activation between the two contractions.
template <bool W2P, int RT, typename H = bf16_t>
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(const FfnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sb1 = reinterpret_cast<float*>(smem + 2 * FF_STAGE);
  float* sb2 = sb1 + p.HID;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fm = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * (64 * RT) + wave * (16 * RT);

  for (int i = tid; i < p.HID; i += 256) sb1[i] = p.b1[i];
  for (int i = tid; i < FF_N; i += 256) sb2[i] = p.b2[i];

  // ---- x fragments: B operand of the first MFMA (token = rt*16 + fm, k = ks*32 + 8g .. +8)
  bf16x8_t xf[RT][8];
  int tok[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    tok[rt] = row0 + rt * 16 + fm;
    const bf16_t* xp = p.X + (size_t)(tok[rt] < p.M ? tok[rt] : p.M - 1) * p.ldx + g * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xf[rt][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(xp + ks * 32));
  }

  // ---- LDS-DMA sources: wave w issues instructions i = 8w .. 8w+7 of each image (64 lanes x 16 B = 1 KB each)
  uint32_t w1off[8], w2off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = wave * 8 + j;
    {
      const int rho = 2 * i + (lane >> 5), qp = lane & 31;
      const int q = qp ^ (rho & 31);
      w1off[j] = (uint32_t)rho * (uint32_t)p.ldw1 + (uint32_t)q * 8u;            // + chunk * 64 * ldw1
    }
    {
      const int rho = 8 * i + (lane >> 3), qp = lane & 7;
      const int q = qp ^ ((rho >> 1) & 7);
      const int ot = rho >> 4, m = rho & 15;
      // output channel of W2 image row rho = 16 ot + 4 g + r: 32 (ot >> 1) + 8 g + 4 (ot & 1) + r -- a lane then owns, for every
      // pair of output tiles q = ot >> 1, the 8 consecutive channels 32 q + 8 g .. + 7, so that one epilogue store (fixed q) has
      // the four lanes g of a token write 64 contiguous bytes (64 consecutive channels per lane left every store instruction
      // four isolated 16-byte pieces per token; same for the residual loads)
      const int ch = (ot >> 1) * 32 + (m >> 2) * 8 + (ot & 1) * 4 + (m & 3);
      w2off[j] = (uint32_t)ch * (uint32_t)p.ldw2 + (uint32_t)q * 8u;             // + chunk * 64
    }
  }
  // The LDS-DMA of chunk c + 1 is issued through inline asm, NOT __builtin_amdgcn_global_load_lds: hipcc cannot tell the DMA's
  // destination stage from the stage the fragment reads address (one array, run-time stage index) and drains the DMA queue
  // (s_waitcnt vmcnt(0)) in front of the first ds_read after every issue -- the weights of the next chunk were waited for before
  // any MFMA of this one (measured: 2.8 us per chunk = DMA latency + compute, nothing overlapped).  An asm statement is outside its
  // bookkeeping (cdna_hip_programming.md 5.7); completion is ours: `s_waitcnt vmcnt(0)` + barrier at the end of the chunk.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(ff_lds_void_t*)smem;
  const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
  auto glds16 = [&](const void* gsrc, uint32_t lds_dst) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  };
  // piece j of a chunk's DMA (this wave's 1 KB of W1 + 1 KB of W2); a chunk = pieces 0 .. 7.  In the loop the pieces are spread over
  // the MFMA batches: sixteen back-to-back LDS-DMA issues (~100 cycles each: the address path) at the top of a chunk kept a wave
  // that is alone on its SIMD from issuing a single MFMA for ~1.6 k cycles
  auto issue_piece = [&](int c, int s, int j) __attribute__((always_inline)) {
    const uint32_t d1 = lds0 + (uint32_t)s * FF_STAGE + wave_u * 8192u;
    const uint32_t d2 = d1 + 32768u;
    const bf16_t* s1 = p.W1 + (size_t)c * FF_HC * p.ldw1;
    const bf16_t* s2 = p.W2 + (size_t)c * FF_HC;
    glds16(s1 + w1off[j], d1 + j * 1024);
    glds16(s2 + w2off[j], d2 + j * 1024);
  };
  auto issue = [&](int c, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_piece(c, s, j);
  };

  // make hipcc wait for the x fragments HERE: left pending, its counted vmcnt waits for them sit inside the chunk loop (it cannot
  // know they have long landed) and -- VMEM returns in order -- drain the untracked LDS-DMA queue with them
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(xf[rt][ks]));

  f32x4_t yacc[16][RT];
#pragma unroll
  for (int ot = 0; ot < 16; ++ot)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) yacc[ot][rt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nc = p.HID / FF_HC;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    const int s = c & 1;
    const int cn = c + 1 < nc ? c + 1 : c;        // pieces of chunk c + 1 go to the other stage, released by the barrier that ended chunk c - 1
                                                  // (last chunk: a harmless reload of itself -- no branch inside the MFMA stream)
    const unsigned char* i1 = smem + s * FF_STAGE;
    const unsigned char* i2 = i1 + 32768;

    // Fragment reads run AHEAD of the MFMAs that consume them (three register sets of 8 fragments, rotating): a wave is
    // alone on its SIMD (128 KB of LDS per workgroup), so nothing else hides the LDS latency -- left to itself hipcc emits
    // read -> wait -> 2 MFMAs -> read ..., i.e. one exposed LDS round trip per MFMA pair (measured: 7.8 k cycles per chunk for
    // 2 k cycles of MFMA).  The sched_barriers pin the order.
    bf16x8_t fa[8], fb[8], fc[8];
    float4 bch[4];
    auto rd1 = [&](int ht, bf16x8_t (&f)[8]) __attribute__((always_inline)) {        // W1 fragments of hidden tile ht, all 8 k steps
      const int rho = ht * 16 + fm;
      const unsigned char* rowp = i1 + rho * 512;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) f[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(rowp + (((ks * 4 + g) ^ (rho & 31)) << 4)));
    };
    auto rd2 = [&](int kk, int og, bf16x8_t (&f)[8]) __attribute__((always_inline)) { // W2 fragments of channel tiles og*8 .. +7, k step kk
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rho = (og * 8 + j) * 16 + fm;
        const int sw = (rho >> 1) & 7;
        if (W2P) {
          f[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const short8_t*>(i2 + rho * 128 + (((kk * 4 + g) ^ sw) << 4)));
        } else {
          const unsigned char* rowp = i2 + rho * 128 + (g & 1) * 8;
          const uint2 lo = *reinterpret_cast<const uint2*>(rowp + (((kk * 4 + (g >> 1)) ^ sw) << 4));
          const uint2 hi = *reinterpret_cast<const uint2*>(rowp + (((kk * 4 + 2 + (g >> 1)) ^ sw) << 4));
          f[j] = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
      }
    };
    f32x4_t hacc[4][RT];
    auto mm1 = [&](int ht, const bf16x8_t (&f)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) hacc[ht][rt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) hacc[ht][rt] = h16<H>::mfma(f[ks], xf[rt][ks], hacc[ht][rt]);
    };
    bf16x8_t hb[2][RT];                                                                 // [kk][rt]: the activations as B operands
    auto act = [&](int ht) __attribute__((always_inline)) {
      // bias + ReLU: lane holds hidden c*64 + ht*16 + 4g .. +3 of token rt*16 + fm
      const float4 b = bch[ht];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        hacc[ht][rt][0] = fmaxf(hacc[ht][rt][0] + b.x, 0.f);
        hacc[ht][rt][1] = fmaxf(hacc[ht][rt][1] + b.y, 0.f);
        hacc[ht][rt][2] = fmaxf(hacc[ht][rt][2] + b.z, 0.f);
        hacc[ht][rt][3] = fmaxf(hacc[ht][rt][3] + b.w, 0.f);
      }
    };
    auto pack = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const uint4 u = make_uint4(h16<H>::pack2(hacc[2 * kk][rt][0], hacc[2 * kk][rt][1]), h16<H>::pack2(hacc[2 * kk][rt][2], hacc[2 * kk][rt][3]),
                                   h16<H>::pack2(hacc[2 * kk + 1][rt][0], hacc[2 * kk + 1][rt][1]),
                                   h16<H>::pack2(hacc[2 * kk + 1][rt][2], hacc[2 * kk + 1][rt][3]));
        hb[kk][rt] = __builtin_bit_cast(bf16x8_t, u);
      }
    };
    auto mm2 = [&](int kk, int og, const bf16x8_t (&f)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          yacc[og * 8 + j][rt] = h16<H>::mfma(f[j], hb[kk][rt], yacc[og * 8 + j][rt]);
    };
#define FF_SB() __builtin_amdgcn_sched_barrier(0)
    // ---- H^T = W1c . x^T (4 hidden tiles x 2 token tiles x 8 k steps), then Y^T += W2c . H^T (k step kk = hidden tiles 2kk, 2kk + 1,
    //      permuted inside the step; 16 channel tiles in two groups of 8)
    // three register sets, reads TWO groups ahead of the MFMAs, and ALL of a group's companions -- the 8 ds_read_b128 of the group
    // after next, the bias / ReLU / pack VALU of the previous hidden tile -- issued INSIDE the MFMA stream (per two MFMAs: one LDS
    // read, a few VALU): a wave alone on its SIMD has no other wave to fill the matrix pipe while it issues reads (measured with the
    // reads between the batches: 5.8 k cycles per chunk for 2.2 k of MFMA issue)
#define FF_MIX(NREAD)                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {      \
    __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);   \
    if (i_ < (NREAD)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);    \
  }
    // the chunk's four bias quads first: read behind the fragments they would turn every counted lgkmcnt wait into a full drain
#pragma unroll
    for (int ht = 0; ht < 4; ++ht) bch[ht] = *reinterpret_cast<const float4*>(sb1 + c * FF_HC + ht * 16 + 4 * g);
    FF_SB();
#define FF_DMA(J) issue_piece(cn, s ^ 1, (J)); FF_SB();
    rd1(0, fa); rd1(1, fb); FF_SB();
    rd1(2, fc); mm1(0, fa); FF_MIX(8); FF_SB(); FF_DMA(0)
    rd1(3, fa); act(0); mm1(1, fb); FF_MIX(8); FF_SB(); FF_DMA(1)
    rd2(0, 0, fb); act(1); pack(0); mm1(2, fc); FF_MIX(8); FF_SB(); FF_DMA(2)
    rd2(0, 1, fc); act(2); mm1(3, fa); FF_MIX(8); FF_SB(); FF_DMA(3)
    rd2(1, 0, fa); act(3); pack(1); mm2(0, 0, fb); FF_MIX(8); FF_SB(); FF_DMA(4)
    rd2(1, 1, fb); mm2(0, 1, fc); FF_MIX(8); FF_SB(); FF_DMA(5)
    mm2(1, 0, fa); FF_SB(); FF_DMA(6)
    FF_DMA(7)
    mm2(1, 1, fb); FF_SB();
#undef FF_DMA
#undef FF_MIX
#undef FF_SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk c + 1 has landed (for this wave's share; the barrier covers the others)
    __syncthreads();
  }

  // ---- epilogue: lane owns channels 32 q + 8 g .. + 7 (q = 0..7) of its tokens; 16-byte pieces (ot pair 2q, 2q+1 = 8 channels).
  // Optional LayerNorm of the finished row (the post-FFN norm of the transformer layer, detrex BaseTransformerLayer "ffn", "norm"):
  // a token's 256 channels sit in the 4 lanes fm + 16 g, so the statistics are two lane swaps; computed on the fp32 sums (two-pass
  // variance), i.e. without the bf16 rounding a separate LayerNorm launch would read -- and without its 89 MB round trip.
  // Every lane runs the arithmetic (clamped row for the residual of rows past M: cross-lane swaps need all lanes); only the store
  // is predicated.
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int tk = tok[rt] < p.M ? tok[rt] : p.M - 1;
    const bf16_t* rp = p.R != nullptr ? p.R + (size_t)tk * p.ldr + g * 8 : nullptr;
    float v[64];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 ba = *reinterpret_cast<const float4*>(sb2 + q * 32 + g * 8);
      const float4 bb = *reinterpret_cast<const float4*>(sb2 + q * 32 + g * 8 + 4);
      v[q * 8 + 0] = yacc[2 * q][rt][0] + ba.x; v[q * 8 + 1] = yacc[2 * q][rt][1] + ba.y;
      v[q * 8 + 2] = yacc[2 * q][rt][2] + ba.z; v[q * 8 + 3] = yacc[2 * q][rt][3] + ba.w;
      v[q * 8 + 4] = yacc[2 * q + 1][rt][0] + bb.x; v[q * 8 + 5] = yacc[2 * q + 1][rt][1] + bb.y;
      v[q * 8 + 6] = yacc[2 * q + 1][rt][2] + bb.z; v[q * 8 + 7] = yacc[2 * q + 1][rt][3] + bb.w;
      if (rp != nullptr) {
        float r[8];
        ld8<H>(reinterpret_cast<const H*>(rp + q * 32), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q * 8 + e] += r[e];
      }
    }
    if (p.ln_w != nullptr) {
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 64; ++e) sum += v[e];
      const float mean = ff_rows_sum(sum) * (1.f / FF_N);
      float d2 = 0.f;
#pragma unroll
      for (int e = 0; e < 64; ++e) { v[e] -= mean; d2 = fmaf(v[e], v[e], d2); }
      const float rstd = rsqrtf(ff_rows_sum(d2) * (1.f / FF_N) + p.ln_eps);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 w = *reinterpret_cast<const float4*>(p.ln_w + (q >> 1) * 32 + g * 8 + (q & 1) * 4);
        const float4 b = *reinterpret_cast<const float4*>(p.ln_b + (q >> 1) * 32 + g * 8 + (q & 1) * 4);
        v[q * 4 + 0] = fmaf(v[q * 4 + 0] * rstd, w.x, b.x); v[q * 4 + 1] = fmaf(v[q * 4 + 1] * rstd, w.y, b.y);
        v[q * 4 + 2] = fmaf(v[q * 4 + 2] * rstd, w.z, b.z); v[q * 4 + 3] = fmaf(v[q * 4 + 3] * rstd, w.w, b.w);
      }
    }
    if (tok[rt] < p.M) {
      bf16_t* yp = p.Y + (size_t)tok[rt] * p.ldy + g * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) st8<H>(reinterpret_cast<H*>(yp + q * 32), v + q * 8);
    }
  }
}

extern "C" int ape_hip_ffn_fused(const void* X, int ldx, const void* W1, int ldw1, const float* b1, const void* W2, int ldw2, const float* b2,
                                 const void* residual, int ldr, void* Y, int ldy, int M, int K, int HID, int N, int w2_permuted, int dt,
                                 const float* ln_weight, const float* ln_bias, float ln_eps, void* stream) {
  APE_CHECK_ARG(X && W1 && b1 && W2 && b2 && Y && M > 0, "ape_hip_ffn_fused: null pointer / empty problem");
  APE_CHECK_ARG(ape_is16(dt), "ape_hip_ffn_fused: dt must be APE_DT_BF16 or APE_DT_F16 (got %d)", dt);
  APE_CHECK_ARG(K == FF_K && N == FF_N && HID % FF_HC == 0 && HID >= FF_HC && HID <= 4096,
                "ape_hip_ffn_fused: the kernel is built for 256 -> HID -> 256 with HID %% 64 == 0, HID <= 4096 (got %d -> %d -> %d)", K, HID, N);
  APE_CHECK_ARG(ldx % 8 == 0 && ldw1 % 8 == 0 && ldw2 % 8 == 0 && ldy % 8 == 0 && (residual == nullptr || ldr % 8 == 0),
                "ape_hip_ffn_fused: leading dimensions must be multiples of 8");
  APE_CHECK_ARG(((uintptr_t)X) % 16 == 0 && ((uintptr_t)W1) % 16 == 0 && ((uintptr_t)W2) % 16 == 0 && ((uintptr_t)Y) % 16 == 0 &&
                    ((uintptr_t)residual) % 16 == 0 && ((uintptr_t)b1) % 16 == 0 && ((uintptr_t)b2) % 16 == 0,
                "ape_hip_ffn_fused: 16-byte aligned pointers");
  APE_CHECK_ARG((size_t)HID * ldw1 * 2 < (1ull << 32) && (size_t)FF_N * ldw2 * 2 < (1ull << 32), "ape_hip_ffn_fused: weights too large for 32-bit offsets");
  FfnParams p;
  p.X = (const bf16_t*)X; p.ldx = ldx; p.W1 = (const bf16_t*)W1; p.ldw1 = ldw1; p.b1 = b1; p.W2 = (const bf16_t*)W2; p.ldw2 = ldw2; p.b2 = b2;
  p.R = (const bf16_t*)residual; p.ldr = ldr; p.Y = (bf16_t*)Y; p.ldy = ldy; p.M = M; p.HID = HID;
  APE_CHECK_ARG((ln_weight == nullptr) == (ln_bias == nullptr) && ((uintptr_t)ln_weight) % 16 == 0 && ((uintptr_t)ln_bias) % 16 == 0,
                "ape_hip_ffn_fused: LayerNorm weight and bias come together, 16-byte aligned");
  p.ln_w = ln_weight; p.ln_b = ln_bias; p.ln_eps = ln_eps;
  const size_t lds = 2 * FF_STAGE + (size_t)(HID + FF_N) * sizeof(float);
  // 192-row workgroups once they still give every CU at least one (the weights stream once per workgroup); APE_FFN_RT=2|3 overrides (A/B)
  const char* rt_env = getenv("APE_FFN_RT");
  const int rt = rt_env != nullptr ? atoi(rt_env) : (ceil_div(M, 192) >= 256 ? 3 : 2);
#define FF_LAUNCH(W2P_, RT_, H_)                                                                                              \
  do {                                                                                                                        \
    static ApeOncePerDevice attr__;                                                                                               \
    if (attr__.first()) {                                                                                                            \
      (void)hipFuncSetAttribute((const void*)ffn_fused_kernel<W2P_, RT_, H_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    }                                                                                                                         \
    APE_LAUNCH((ffn_fused_kernel<W2P_, RT_, H_>), dim3(ceil_div(M, 64 * RT_)), dim3(256), lds, (hipStream_t)stream, p); \
  } while (0)
  if (dt == APE_DT_F16) {
    if (w2_permuted && rt == 3) FF_LAUNCH(true, 3, f16_t); else if (w2_permuted) FF_LAUNCH(true, 2, f16_t); else FF_LAUNCH(false, 2, f16_t);
  } else {
    if (w2_permuted && rt == 3) FF_LAUNCH(true, 3, bf16_t); else if (w2_permuted) FF_LAUNCH(true, 2, bf16_t); else FF_LAUNCH(false, 2, bf16_t);
  }
#undef FF_LAUNCH
  APE_CHECK_LAUNCH("ffn_fused_kernel");
  return 0;
}
