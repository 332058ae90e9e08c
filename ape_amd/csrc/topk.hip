// topk.hip -- the data-dependent selections of the forward pass as fixed-shape device code (gfx950), replacing the
// tensor-library sort / cumsum / scatter chains around the two NMS stages:
//
//  * encoder proposals (ape/modeling/ape_deta/deformable_transformer_vl.py:503-533, 565-627)
//      enc_finalize        per token: ambiguous-head argmax, + anchors, sigmoid, cxcywh -> clamped xyxy
//      proposal_topk       per level: top pre_nms_topk of sigmoid(logit) (ties: lowest index) with the zero-score fillers a
//                          short level borrows, and the "naive top-k" list of the fallback (:600-606)
//      proposal_order      the 5 x 1000 candidates in global descending-logit order and, level-major, as NMS segments
//      proposal_quota      per-level quota (num_queries / L) + fill-up in score order, or the fallback list
//  * final detections (ape/modeling/ape_deta/fast_rcnn.py:97-201, deformable_detr_segm_vl.py:759-810)
//      det_boxes           cxcywh -> xyxy * (w,h,w,h), finite test, clip
//      class_sort          per class: queries by descending sigmoid score (stable), validity flags
//      det_topk            top-k of the NMS survivors over all (class, query) pairs
//
// One 1024-thread workgroup owns one selection problem: an 8-bit MSD radix select over order-preserving 32-bit keys (LDS
// histogram, wave-aggregated atomics so that runs of equal keys cost one atomic per wave), a stable compaction (ties
// by lowest index = the tie rule DESIGN.md defines), and a bitonic sort of <= 1024 (key, index) composites in LDS.
// Latency-bound by construction (these are the serial joints of the pipeline); what they buy is ~100 fewer launches.
#include "common.h"
#include "../../include/ape_hip.h"

typedef unsigned long long u64;
#define TK_THREADS 1024

__device__ __forceinline__ uint32_t ordkey(float x) {
  const uint32_t u = __float_as_uint(x);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);   // larger float <-> larger key; NaN (positive) above +inf
}
__device__ __forceinline__ float ordkey_inv(uint32_t k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

struct TkLevels {
  int start[8];
  int n[8];
  int L;
};
__device__ __forceinline__ int level_of(const TkLevels& lv, int idx) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) l += (i < lv.L && idx >= lv.start[i]) ? 1 : 0;
  return l;
}

// ---- bitonic sort, descending, of N (power of two) u64 in LDS by a 1024-thread workgroup
template <int N>
__device__ __forceinline__ void bitonic_desc(u64* s) {
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < N; i += TK_THREADS) {
        const int p = i ^ j;
        if (p > i) {
          const u64 a = s[i], b = s[p];
          const bool first_larger = (i & k) == 0;
          if (first_larger ? (a < b) : (a > b)) { s[i] = b; s[p] = a; }
        }
      }
      __syncthreads();
    }
  }
}

// ---- radix select: the k-th largest key of key(0..n-1) (n > k >= 1) -> thr, and how many keys == thr belong to the top k
template <typename KeyFn>
__device__ __forceinline__ void radix_select(KeyFn key, int n, int k, uint32_t* hist, uint32_t* sh, uint32_t& thr, uint32_t& rem) {
  uint32_t prefix = 0, r = (uint32_t)k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int b = threadIdx.x; b < 256; b += TK_THREADS) hist[b] = 0;
    __syncthreads();
    const uint32_t himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    for (int i0 = 0; i0 < n; i0 += TK_THREADS) {
      const int i = i0 + threadIdx.x;
      const uint32_t kk = i < n ? key(i) : 0u;
      bool live = i < n && (kk & himask) == prefix;
      const uint32_t bin = (kk >> shift) & 255u;
      // wave-aggregated histogram update: one atomic per distinct bin per wave
      u64 todo = __ballot(live);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
        const u64 same = __ballot(live && bin == lb);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
        todo &= ~same;
        if (live && bin == lb) live = false;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t cum = 0;
      int b = 255;
      for (; b > 0; --b) {
        if (cum + hist[b] >= r) break;
        cum += hist[b];
      }
      sh[0] = prefix | ((uint32_t)b << shift);
      sh[1] = r - cum;
    }
    __syncthreads();
    prefix = sh[0];
    r = sh[1];
    __syncthreads();
  }
  thr = prefix;
  rem = r;
}

// ---- top-k of key(0..n-1) into comp[0..1023] (sorted descending; entries >= min(k, n) are zero).  composite(i) =
//      key << 32 | (0xffffffff - index): equal keys come out in ascending index order.  Streaming form: key(i) is evaluated once
//      per pass, so the source must be cheap to re-read (LDS) -- the register-resident form below is the one for global memory.
template <typename KeyFn, typename CompFn>
__device__ __forceinline__ void block_topk(KeyFn key, CompFn composite, int n, int k, u64* comp, uint32_t* hist, uint32_t* sh, uint32_t* wcnt) {
  comp[threadIdx.x] = 0ull;
  __syncthreads();
  if (n <= k) {
    for (int i = threadIdx.x; i < n; i += TK_THREADS) comp[i] = composite(i);
    __syncthreads();
  } else {
    uint32_t thr, rem;
    radix_select(key, n, k, hist, sh, thr, rem);
    const uint32_t ngt = (uint32_t)k - rem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t base_gt = 0, base_eq = 0;   // uniform across the workgroup (every thread tracks them)
    for (int i0 = 0; i0 < n; i0 += TK_THREADS) {
      const int i = i0 + threadIdx.x;
      const uint32_t kk = i < n ? key(i) : 0u;
      const bool gt = i < n && kk > thr, eq = i < n && kk == thr;
      const u64 bg = __ballot(gt), be = __ballot(eq);
      const u64 lt = (1ull << lane) - 1ull;
      if (lane == 0) { wcnt[wave] = (uint32_t)__popcll(bg); wcnt[16 + wave] = (uint32_t)__popcll(be); }
      __syncthreads();
      uint32_t wg = 0, we = 0, tg = 0, te = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const uint32_t cg = wcnt[w], ce = wcnt[16 + w];
        if (w < wave) { wg += cg; we += ce; }
        tg += cg; te += ce;
      }
      if (gt) comp[base_gt + wg + (uint32_t)__popcll(bg & lt)] = composite(i);
      if (eq) {
        const uint32_t rank = base_eq + we + (uint32_t)__popcll(be & lt);
        if (rank < rem) comp[ngt + rank] = composite(i);
      }
      base_gt += tg;
      base_eq += te;
      __syncthreads();
    }
  }
  bitonic_desc<1024>(comp);
}

// ---- the same for PER * 1024 keys held in REGISTERS: thread t owns elements e = j * 1024 + t (j < PER; coalesced loads, all
//      PER of them in flight at once -- a single workgroup that re-reads global memory once per pass pays a full memory latency
//      per element and pass: 784 us for 87 k tokens, measured), e < n valid.  gbase = index of element 0 in the composites.
template <int PER>
__device__ __forceinline__ void block_topk_regs(const uint32_t (&kk)[PER], int n, int k, uint32_t gbase, u64* comp, uint32_t* hist,
                                                uint32_t* sh, uint32_t* wcnt) {
  comp[threadIdx.x] = 0ull;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (n <= k) {                                    // n <= k <= 1024: only j = 0 holds elements
    if ((int)threadIdx.x < n) comp[threadIdx.x] = ((u64)kk[0] << 32) | (u64)(0xffffffffu - (gbase + threadIdx.x));
    __syncthreads();
  } else {
    uint32_t prefix = 0, r = (uint32_t)k;
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (int b = threadIdx.x; b < 256; b += TK_THREADS) hist[b] = 0;
      __syncthreads();
      const uint32_t himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int e = j * TK_THREADS + threadIdx.x;
        bool live = e < n && (kk[j] & himask) == prefix;
        const uint32_t bin = (kk[j] >> shift) & 255u;
        u64 todo = __ballot(live);
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
          const u64 same = __ballot(live && bin == lb);
          if (lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
          todo &= ~same;
          if (live && bin == lb) live = false;
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t cum = 0;
        int b = 255;
        for (; b > 0; --b) {
          if (cum + hist[b] >= r) break;
          cum += hist[b];
        }
        sh[0] = prefix | ((uint32_t)b << shift);
        sh[1] = r - cum;
      }
      __syncthreads();
      prefix = sh[0];
      r = sh[1];
      __syncthreads();
    }
    const uint32_t thr = prefix, rem = r, ngt = (uint32_t)k - rem;
    uint32_t base_gt = 0, base_eq = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = j * TK_THREADS + threadIdx.x;
      const bool gt = e < n && kk[j] > thr, eq = e < n && kk[j] == thr;
      const u64 bg = __ballot(gt), be = __ballot(eq);
      const u64 lt = (1ull << lane) - 1ull;
      if (lane == 0) { wcnt[wave] = (uint32_t)__popcll(bg); wcnt[16 + wave] = (uint32_t)__popcll(be); }
      __syncthreads();
      uint32_t wg = 0, we = 0, tg = 0, te = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const uint32_t cg = wcnt[w], ce = wcnt[16 + w];
        if (w < wave) { wg += cg; we += ce; }
        tg += cg; te += ce;
      }
      const u64 c = ((u64)kk[j] << 32) | (u64)(0xffffffffu - (gbase + (uint32_t)e));
      if (gt) comp[base_gt + wg + (uint32_t)__popcll(bg & lt)] = c;
      if (eq) {
        const uint32_t rank = base_eq + we + (uint32_t)__popcll(be & lt);
        if (rank < rem) comp[ngt + rank] = c;
      }
      base_gt += tg;
      base_eq += te;
      __syncthreads();
    }
  }
  bitonic_desc<1024>(comp);
}

// Two-stage top-k over a large array: stage 1 = one workgroup per chunk of PER * 1024 elements (register resident) writes its
// sorted top-k composites to ws[chunk * 1024 ..]; stage 2 = one workgroup per problem gathers (chunks x k) composites into LDS
// and selects among them (the global top-k is a subset of the union of the chunks' top-k; equal keys keep ascending index
// order because chunks are index ranges in ascending order and each chunk's list is sorted by (key desc, index asc)).
#define TK_STAGE2_MAX 12288        // composites stage 2 holds in LDS (96 KiB)
#define TK_MAX_JOBS 96
struct TkJobs {
  int njobs, nseg;
  int seg_first[8], seg_jobs[8], seg_k[8], seg_n[8], seg_mode[8];     // mode 0: key = x, 1: key = sigmoid(x)
  int start[TK_MAX_JOBS], n[TK_MAX_JOBS], seg[TK_MAX_JOBS];
};

__device__ __forceinline__ void stage2_gather(const u64* __restrict__ ws, int first_job, int njobs, int k, u64* lds) {
  const int m = njobs * k;
  for (int i = threadIdx.x; i < m; i += TK_THREADS) lds[i] = ws[(size_t)(first_job + i / k) * 1024 + (i % k)];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- encoder proposals
// cls2 [T,2], d [T,8] (main | ambiguous box deltas), anchors [T,4] (logit space, +inf where unusable)
__global__ __launch_bounds__(256) void enc_finalize_kernel(const float* __restrict__ cls2, const float* __restrict__ d,
                                                           const float* __restrict__ anchors, int T, float* __restrict__ enc_class,
                                                           float* __restrict__ enc_coord, float* __restrict__ xyxy) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float2 c = *reinterpret_cast<const float2*>(cls2 + 2 * (size_t)t);
  const bool pick = c.y > c.x;                                  // the larger logit, the first on ties (:527-533)
  enc_class[t] = pick ? c.y : c.x;
  const float4 dd = *reinterpret_cast<const float4*>(d + 8 * (size_t)t + (pick ? 4 : 0));
  const float4 a = *reinterpret_cast<const float4*>(anchors + 4 * (size_t)t);
  const float4 u = make_float4(dd.x + a.x, dd.y + a.y, dd.z + a.z, dd.w + a.w);
  *reinterpret_cast<float4*>(enc_coord + 4 * (size_t)t) = u;
  const float cx = sigmoid_f(u.x), cy = sigmoid_f(u.y), w = sigmoid_f(u.z), h = sigmoid_f(u.w);
  float4 b = make_float4(cx - 0.5f * w, cy - 0.5f * h, cx + 0.5f * w, cy + 0.5f * h);
  b.x = fminf(fmaxf(b.x, 0.f), 1.f); b.y = fminf(fmaxf(b.y, 0.f), 1.f);
  b.z = fminf(fmaxf(b.z, 0.f), 1.f); b.w = fminf(fmaxf(b.w, 0.f), 1.f);
  *reinterpret_cast<float4*>(xyxy + 4 * (size_t)t) = b;
}

// stage 1: one workgroup per chunk of a level (key = sigmoid(logit)) or of the whole token range (fallback list, key = logit)
template <int PER>
__global__ __launch_bounds__(TK_THREADS) void topk_stage1_kernel(const float* __restrict__ x, TkJobs jobs, u64* __restrict__ ws) {
  __shared__ u64 comp[1024];
  __shared__ uint32_t hist[256], sh[2], wcnt[32];
  const int job = blockIdx.x, seg = jobs.seg[job], start = jobs.start[job], n = jobs.n[job];
  const bool sig = jobs.seg_mode[seg] != 0;
  uint32_t kk[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int e = j * TK_THREADS + threadIdx.x;
    const float v = e < n ? x[start + e] : 0.f;
    kk[j] = e < n ? ordkey(sig ? sigmoid_f(v) : v) : 0u;
  }
  block_topk_regs<PER>(kk, n, jobs.seg_k[seg], (uint32_t)start, comp, hist, sh, wcnt);
  ws[(size_t)job * 1024 + threadIdx.x] = comp[threadIdx.x];
}

// stage 2: workgroup l < L: level l's candidates cand[l*k .. +k); workgroup L: the fallback list alt[0 .. k_alt)
__global__ __launch_bounds__(TK_THREADS) void proposal_topk2_kernel(const u64* __restrict__ ws, TkJobs jobs, TkLevels lv, int k, int k_alt,
                                                                    int32_t* __restrict__ cand, int32_t* __restrict__ alt) {
  extern __shared__ u64 lds[];
  __shared__ u64 comp[1024];
  __shared__ uint32_t hist[256], sh[2], wcnt[32];
  const int l = blockIdx.x;
  const int kk = jobs.seg_k[l], m = jobs.seg_jobs[l] * kk;
  stage2_gather(ws, jobs.seg_first[l], jobs.seg_jobs[l], kk, lds);
  auto key = [&](int i) { return (uint32_t)(lds[i] >> 32); };
  auto composite = [&](int i) { return lds[i]; };
  block_topk(key, composite, m, kk, comp, hist, sh, wcnt);
  if (l == lv.L) {
    for (int r = threadIdx.x; r < k_alt; r += TK_THREADS) alt[r] = (int32_t)(0xffffffffu - (uint32_t)(comp[r] & 0xffffffffull));
    return;
  }
  const int start = lv.start[l], n = lv.n[l];
  const int own = n < k ? n : k;
  for (int r = threadIdx.x; r < k; r += TK_THREADS) {
    int idx;
    if (r < own) {
      idx = (int)(0xffffffffu - (uint32_t)(comp[r] & 0xffffffffull));      // composites carry the token index
    } else {                       // torch.topk over sigmoid * level_mask: the zero scores of OTHER levels, lowest index first
      const int e = r - n;
      idx = e >= start ? e + n : e;
    }
    cand[l * k + r] = idx;
  }
}

// 5-field exclusive scan over the workgroup (fields < 65536): out = counts of all lower threads, tot = workgroup totals
__device__ __forceinline__ void scan5(const uint32_t v[5], uint32_t out[5], uint32_t tot[5], u64* sa, uint32_t* sb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 a = (u64)v[0] | ((u64)v[1] << 16) | ((u64)v[2] << 32) | ((u64)v[3] << 48);
  uint32_t b = v[4];
  u64 ia = a;
  uint32_t ib = b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t lo = __shfl_up((uint32_t)(ia & 0xffffffffull), o, 64), hi = __shfl_up((uint32_t)(ia >> 32), o, 64);
    const uint32_t tb = __shfl_up(ib, o, 64);
    if (lane >= o) { ia += ((u64)hi << 32) | lo; ib += tb; }
  }
  __syncthreads();                 // sa / sb may still be read from a previous call
  if (lane == 63) { sa[wave] = ia; sb[wave] = ib; }
  __syncthreads();
  u64 wa = 0, ta = 0;
  uint32_t wb = 0, tb2 = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    if (w < wave) { wa += sa[w]; wb += sb[w]; }
    ta += sa[w]; tb2 += sb[w];
  }
  const u64 ea = wa + ia - a;
  out[0] = (uint32_t)(ea & 0xffff); out[1] = (uint32_t)((ea >> 16) & 0xffff); out[2] = (uint32_t)((ea >> 32) & 0xffff);
  out[3] = (uint32_t)((ea >> 48) & 0xffff); out[4] = wb + ib - b;
  tot[0] = (uint32_t)(ta & 0xffff); tot[1] = (uint32_t)((ta >> 16) & 0xffff); tot[2] = (uint32_t)((ta >> 32) & 0xffff);
  tot[3] = (uint32_t)((ta >> 48) & 0xffff); tot[4] = tb2;
}

#define PO_PAD 8192
#define PO_PER (PO_PAD / TK_THREADS)
// one workgroup: candidates in global descending-logit order (stable: position in `cand`) = "A order"; level-major with
// the A order kept inside a level = "B order" (the NMS segments).  n <= 8192, L <= 5.
__global__ __launch_bounds__(TK_THREADS) void proposal_order_kernel(const int32_t* __restrict__ cand, int n, const float* __restrict__ logit,
                                                                    const float* __restrict__ xyxy, TkLevels lv,
                                                                    float* __restrict__ boxes_b, int32_t* __restrict__ groups_b,
                                                                    int32_t* __restrict__ seg, int32_t* __restrict__ cand_a,
                                                                    int32_t* __restrict__ lv_a, int32_t* __restrict__ pos_b) {
  extern __shared__ u64 po_lds[];   // 2 x (chunks x 1024) composites: chunk-sorted, then merged
  __shared__ u64 sa[16];
  __shared__ uint32_t sb[16];
  // Sort = independent bitonic sorts of the 1024-element chunks (55 passes over n elements instead of 91 passes over 8192: round 3's
  // single 8192-element network took 143 us) + a rank merge: all composites are distinct, so an element's global rank is its position
  // in its own chunk plus, for every other chunk, the number of larger elements there (a 10-step binary search in LDS).
  const int nchunks = (n + 1023) >> 10, npad = nchunks << 10;
  u64* work = po_lds;
  u64* srt = po_lds + npad;
  for (int p = threadIdx.x; p < npad; p += TK_THREADS)     // padding: distinct values below every real composite (low word >= 2^32 - 8192)
    work[p] = p < n ? (((u64)ordkey(logit[cand[p]]) << 32) | (u64)(0xffffffffu - (uint32_t)p)) : (u64)(npad - 1 - p);
  __syncthreads();
  for (int k = 2; k <= 1024; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += TK_THREADS) {
        const int q = i ^ j;                                 // j < 1024: the partner lives in the same chunk
        if (q > i) {
          const u64 a = work[i], b = work[q];
          const bool first_larger = ((i & 1023) & k) == 0;
          if (first_larger ? (a < b) : (a > b)) { work[i] = b; work[q] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < npad; i += TK_THREADS) {
    const u64 v = work[i];
    const int c = i >> 10;
    int rank = i & 1023;
    for (int o = 0; o < nchunks; ++o) {
      if (o == c) continue;
      const u64* ch = work + (o << 10);
      int lo = 0, hi = 1024;                                 // first position whose element is smaller than v
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ch[mid] > v) lo = mid + 1; else hi = mid;
      }
      rank += lo;
    }
    srt[rank] = v;
  }
  __syncthreads();
  // thread t owns A positions t*PO_PER .. +PO_PER
  int idx[PO_PER], lev[PO_PER];
  uint32_t cnt[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < PO_PER; ++j) {
    const int a = threadIdx.x * PO_PER + j;
    idx[j] = -1; lev[j] = 0;
    if (a < n) {
      const int p = (int)(0xffffffffu - (uint32_t)(srt[a] & 0xffffffffull));
      idx[j] = cand[p];
      lev[j] = level_of(lv, idx[j]);
      cnt[lev[j]] += 1;
    }
  }
  uint32_t ex[5], tot[5];
  scan5(cnt, ex, tot, sa, sb);
  uint32_t segs[6];
  segs[0] = 0;
#pragma unroll
  for (int l = 0; l < 5; ++l) segs[l + 1] = segs[l] + tot[l];
  if (threadIdx.x == 0)
    for (int l = 0; l <= lv.L; ++l) seg[l] = (int32_t)segs[l];
#pragma unroll
  for (int j = 0; j < PO_PER; ++j) {
    const int a = threadIdx.x * PO_PER + j;
    if (a < n) {
      const int l = lev[j];
      const int b = (int)(segs[l] + ex[l]);
      ex[l] += 1;
      cand_a[a] = idx[j]; lv_a[a] = l; pos_b[a] = b;
      groups_b[b] = l;
      *reinterpret_cast<float4*>(boxes_b + 4 * (size_t)b) = *reinterpret_cast<const float4*>(xyxy + 4 * (size_t)idx[j]);
    }
  }
}

// one workgroup: (:598-627) per level the first nq / L survivors in score order, then the best of the rest up to nq; with fewer
// than nq survivors in total the fallback list takes the candidates' place.  out [nq] int64, zero where no candidate exists.
__global__ __launch_bounds__(TK_THREADS) void proposal_quota_kernel(const int32_t* __restrict__ cand_a, const int32_t* __restrict__ lv_a,
                                                                    const int32_t* __restrict__ pos_b, const uint8_t* __restrict__ keep_b,
                                                                    int n, const int32_t* __restrict__ alt, int n_alt, TkLevels lv, int nq,
                                                                    long long* __restrict__ out) {
  __shared__ u64 sa[16];
  __shared__ uint32_t sb[16];
  int X[PO_PER], lev[PO_PER];
  bool valid[PO_PER];
  uint32_t c[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < PO_PER; ++j) {
    const int a = threadIdx.x * PO_PER + j;
    X[j] = 0; lev[j] = 0; valid[j] = false;
    if (a < n) {
      X[j] = cand_a[a]; lev[j] = lv_a[a];
      valid[j] = keep_b[pos_b[a]] != 0;
      c[0] += valid[j] ? 1u : 0u;
    }
  }
  uint32_t ex[5], tot[5];
  scan5(c, ex, tot, sa, sb);
  const bool use_alt = (int)tot[0] < nq;
  if (use_alt) {
#pragma unroll
    for (int j = 0; j < PO_PER; ++j) {
      const int a = threadIdx.x * PO_PER + j;
      valid[j] = a < n && a < n_alt;
      X[j] = valid[j] ? alt[a] : 0;
      lev[j] = level_of(lv, X[j]);
    }
  }
  for (int r = threadIdx.x; r < nq; r += TK_THREADS) out[r] = 0;
  // per-level rank among the valid ones
#pragma unroll
  for (int l = 0; l < 5; ++l) c[l] = 0;
#pragma unroll
  for (int j = 0; j < PO_PER; ++j)
    if (valid[j]) c[lev[j]] += 1;
  scan5(c, ex, tot, sa, sb);
  const uint32_t quota = (uint32_t)(nq / lv.L);
  bool sel[PO_PER];
  uint32_t c2[5] = {0, 0, 0, 0, 0};        // [0] selected, [1] valid & not selected
#pragma unroll
  for (int j = 0; j < PO_PER; ++j) {
    sel[j] = false;
    if (valid[j]) {
      sel[j] = ex[lev[j]] < quota;
      ex[lev[j]] += 1;
      c2[sel[j] ? 0 : 1] += 1;
    }
  }
  scan5(c2, ex, tot, sa, sb);
  const uint32_t need = (uint32_t)nq - tot[0];
  uint32_t c3[5] = {0, 0, 0, 0, 0};
  uint32_t rest = ex[1];
#pragma unroll
  for (int j = 0; j < PO_PER; ++j) {
    if (valid[j] && !sel[j]) {
      if (rest < need) sel[j] = true;
      rest += 1;
    }
    c3[0] += sel[j] ? 1u : 0u;
  }
  scan5(c3, ex, tot, sa, sb);
  uint32_t slot = ex[0];
#pragma unroll
  for (int j = 0; j < PO_PER; ++j)
    if (sel[j]) {
      if (slot < (uint32_t)nq) out[slot] = (long long)X[j];
      slot += 1;
    }
}

// ---------------------------------------------------------------------------------------------- final detections
// one wave per query: xyxy = cxcywh -> corners * scale, clip to [0, scale]; rows with a non-finite box or score become zero boxes
__global__ __launch_bounds__(64) void det_boxes_kernel(const float* __restrict__ logits, int ldl, int K, const float* __restrict__ boxes,
                                                       const float* __restrict__ scale, float* __restrict__ xyxy,
                                                       uint8_t* __restrict__ finite) {
  const int q = blockIdx.x, lane = threadIdx.x;
  bool ok = true;
  for (int c = lane; c < K; c += 64) ok = ok && isfinite(sigmoid_f(logits[(size_t)q * ldl + c]));
  ok = __ballot(!ok) == 0ull;
  if (lane == 0) {
    const float4 b = *reinterpret_cast<const float4*>(boxes + 4 * (size_t)q);
    const float4 s = *reinterpret_cast<const float4*>(scale);
    float4 o = make_float4((b.x - 0.5f * b.z) * s.x, (b.y - 0.5f * b.w) * s.y, (b.x + 0.5f * b.z) * s.z, (b.y + 0.5f * b.w) * s.w);
    ok = ok && isfinite(o.x) && isfinite(o.y) && isfinite(o.z) && isfinite(o.w);
    o.x = fminf(fmaxf(o.x, 0.f), s.x); o.y = fminf(fmaxf(o.y, 0.f), s.y);
    o.z = fminf(fmaxf(o.z, 0.f), s.z); o.w = fminf(fmaxf(o.w, 0.f), s.w);
    if (!ok) o = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(xyxy + 4 * (size_t)q) = o;
    finite[q] = ok ? 1 : 0;
  }
}

// one workgroup per class: the Q <= 1024 queries by descending sigmoid score, ties by lowest query index
__global__ __launch_bounds__(TK_THREADS) void class_sort_kernel(const float* __restrict__ logits, int ldl, int Q, const uint8_t* __restrict__ finite,
                                                                float thresh, float* __restrict__ sorted, int32_t* __restrict__ order,
                                                                uint8_t* __restrict__ valid) {
  __shared__ u64 comp[1024];
  const int c = blockIdx.x, q = threadIdx.x;
  comp[q] = q < Q ? (((u64)ordkey(sigmoid_f(logits[(size_t)q * ldl + c])) << 32) | (u64)(0xffffffffu - (uint32_t)q)) : 0ull;
  __syncthreads();
  bitonic_desc<1024>(comp);
  if (q < Q) {
    const float s = ordkey_inv((uint32_t)(comp[q] >> 32));
    const int src = (int)(0xffffffffu - (uint32_t)(comp[q] & 0xffffffffull));
    sorted[(size_t)c * Q + q] = s;
    order[(size_t)c * Q + q] = src;
    valid[(size_t)c * Q + q] = (s > thresh && finite[src]) ? 1 : 0;
  }
}

// top-k of the NMS survivors over all (class, rank) pairs; suppressed pairs score -1 (fast_rcnn.py:192-201).  Stage 1: one
// workgroup per PER * 1024 pairs; stage 2: one workgroup over the chunks' lists
template <int PER>
__global__ __launch_bounds__(TK_THREADS) void det_topk1_kernel(const float* __restrict__ sorted, const uint8_t* __restrict__ keep, int total,
                                                               int k, u64* __restrict__ ws) {
  __shared__ u64 comp[1024];
  __shared__ uint32_t hist[256], sh[2], wcnt[32];
  const int start = blockIdx.x * (PER * TK_THREADS);
  const int n = min(PER * TK_THREADS, total - start);
  uint32_t kk[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int e = j * TK_THREADS + threadIdx.x;
    kk[j] = e < n ? ordkey(keep[start + e] ? sorted[start + e] : -1.f) : 0u;
  }
  block_topk_regs<PER>(kk, n, k, (uint32_t)start, comp, hist, sh, wcnt);
  ws[(size_t)blockIdx.x * 1024 + threadIdx.x] = comp[threadIdx.x];
}

// intermediate stage for selections whose (lists x k) exceeds stage 2's LDS (e.g. LVIS-1203 x 900 pairs with test_topk_per_image > 722):
// workgroup b merges lists b * group .. of `in` (k sorted composites each, stride 1024) into list b of `out`.  The union-of-top-k
// argument of the two-stage scheme holds at every level, and composites carry (key, ~index), so ties still resolve to the lowest index.
__global__ __launch_bounds__(TK_THREADS) void topk_merge_kernel(const u64* __restrict__ in, int nlists, int group, int k, u64* __restrict__ out) {
  extern __shared__ u64 lds[];
  __shared__ u64 comp[1024];
  __shared__ uint32_t hist[256], sh[2], wcnt[32];
  const int first = blockIdx.x * group;
  const int cnt = min(group, nlists - first);
  stage2_gather(in, first, cnt, k, lds);
  auto key = [&](int i) { return (uint32_t)(lds[i] >> 32); };
  auto composite = [&](int i) { return lds[i]; };
  block_topk(key, composite, cnt * k, k, comp, hist, sh, wcnt);
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = comp[threadIdx.x];
}

__global__ __launch_bounds__(TK_THREADS) void det_topk2_kernel(const u64* __restrict__ ws, int nchunks, const int32_t* __restrict__ order,
                                                               const float* __restrict__ xyxy, int Q, int k, float* __restrict__ det_boxes,
                                                               float* __restrict__ det_scores, long long* __restrict__ det_classes,
                                                               long long* __restrict__ det_query) {
  extern __shared__ u64 lds[];
  __shared__ u64 comp[1024];
  __shared__ uint32_t hist[256], sh[2], wcnt[32];
  stage2_gather(ws, 0, nchunks, k, lds);
  auto key = [&](int i) { return (uint32_t)(lds[i] >> 32); };
  auto composite = [&](int i) { return lds[i]; };
  block_topk(key, composite, nchunks * k, k, comp, hist, sh, wcnt);
  for (int r = threadIdx.x; r < k; r += TK_THREADS) {
    const int flat = (int)(0xffffffffu - (uint32_t)(comp[r] & 0xffffffffull));
    const int qi = order[flat];
    det_scores[r] = ordkey_inv((uint32_t)(comp[r] >> 32));
    det_classes[r] = (long long)(flat / Q);
    det_query[r] = (long long)qi;
    *reinterpret_cast<float4*>(det_boxes + 4 * (size_t)r) = *reinterpret_cast<const float4*>(xyxy + 4 * (size_t)qi);
  }
}

// ---------------------------------------------------------------------------------------------- C-ABI
static int fill_levels(TkLevels& lv, const int* level_start, const int* level_n, int L) {
  APE_CHECK_ARG(level_start && level_n && L >= 1 && L <= 5, "proposal selection: 1 <= levels <= 5");
  for (int i = 0; i < 8; ++i) { lv.start[i] = i < L ? level_start[i] : 0x7fffffff; lv.n[i] = i < L ? level_n[i] : 0; }
  lv.L = L;
  return 0;
}

extern "C" int ape_hip_enc_finalize(const float* cls2, const float* d, const float* anchors, int T, float* enc_class, float* enc_coord,
                                    float* xyxy, void* stream) {
  APE_CHECK_ARG(cls2 && d && anchors && enc_class && enc_coord && xyxy && T > 0, "ape_hip_enc_finalize: bad arguments");
  APE_LAUNCH(enc_finalize_kernel, dim3(ceil_div(T, 256)), dim3(256), 0, (hipStream_t)stream, cls2, d, anchors, T, enc_class,
                     enc_coord, xyxy);
  APE_CHECK_LAUNCH("enc_finalize_kernel");
  return 0;
}

// elements per stage-1 workgroup: the smallest of 8 / 32 / 64 thousand that keeps (chunks x k) of every problem inside stage 2's LDS
static int topk_per(const int* seg_n, const int* seg_k, int nseg) {
  const int pers[3] = {8, 32, 64};
  for (int pi = 0; pi < 3; ++pi) {
    bool ok = true;
    int jobs = 0;
    for (int s = 0; s < nseg; ++s) {
      const int c = ceil_div(seg_n[s], pers[pi] * TK_THREADS);
      ok = ok && (long long)c * seg_k[s] <= TK_STAGE2_MAX;
      jobs += c;
    }
    if (ok && jobs <= TK_MAX_JOBS) return pers[pi];
  }
  return 0;
}

static void set_stage2_attr() {
  static ApeOncePerDevice done;
  if (done.first()) {
    (void)hipFuncSetAttribute((const void*)proposal_topk2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    (void)hipFuncSetAttribute((const void*)det_topk2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    (void)hipFuncSetAttribute((const void*)topk_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
  }
}

// uint64 words of workspace the two-stage selections need: 1024 per stage-1 workgroup
extern "C" int ape_hip_topk_workspace_words(int n_total) {
  // proposals: <= TK_MAX_JOBS workgroups; detections: one per 8 k pairs at most
  // + half as many again for the lists of an intermediate merge level (det_topk with lists x k beyond stage 2's LDS)
  const int det = ceil_div(n_total > 0 ? n_total : 1, 8 * TK_THREADS);
  const int lists = det > TK_MAX_JOBS ? det : TK_MAX_JOBS;
  return 1024 * (lists + lists / 2 + 1);
}

extern "C" int ape_hip_proposal_topk(const float* logit, int T, const int* level_start, const int* level_n, int L, int k, int k_alt,
                                     uint64_t* workspace, int32_t* cand, int32_t* alt, void* stream) {
  TkLevels lv;
  if (fill_levels(lv, level_start, level_n, L)) return -1;
  APE_CHECK_ARG(logit && workspace && cand && alt && T > 0 && k >= 1 && k <= 1024 && k_alt >= 1 && k_alt <= 1024 && k <= T && k_alt <= T,
                "ape_hip_proposal_topk: 1 <= k, k_alt <= min(1024, T)");
  int seg_n[8], seg_k[8];
  for (int l = 0; l < L; ++l) { seg_n[l] = level_n[l]; seg_k[l] = k; }
  seg_n[L] = T; seg_k[L] = k_alt;
  const int per = topk_per(seg_n, seg_k, L + 1);
  APE_CHECK_ARG(per > 0, "ape_hip_proposal_topk: %d tokens do not fit the two-stage selection", T);
  TkJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  jobs.nseg = L + 1;
  const int chunk = per * TK_THREADS;
  for (int s = 0; s <= L; ++s) {
    const int base = s < L ? level_start[s] : 0;
    jobs.seg_first[s] = jobs.njobs; jobs.seg_k[s] = seg_k[s]; jobs.seg_n[s] = seg_n[s]; jobs.seg_mode[s] = s < L ? 1 : 0;
    for (int c0 = 0; c0 < seg_n[s]; c0 += chunk) {
      jobs.start[jobs.njobs] = base + c0; jobs.n[jobs.njobs] = seg_n[s] - c0 < chunk ? seg_n[s] - c0 : chunk; jobs.seg[jobs.njobs] = s;
      ++jobs.njobs;
    }
    jobs.seg_jobs[s] = jobs.njobs - jobs.seg_first[s];
  }
  hipStream_t st = (hipStream_t)stream;
  u64* ws = reinterpret_cast<u64*>(workspace);
  if (per == 8) APE_LAUNCH(topk_stage1_kernel<8>, dim3(jobs.njobs), dim3(TK_THREADS), 0, st, logit, jobs, ws);
  else if (per == 32) APE_LAUNCH(topk_stage1_kernel<32>, dim3(jobs.njobs), dim3(TK_THREADS), 0, st, logit, jobs, ws);
  else APE_LAUNCH(topk_stage1_kernel<64>, dim3(jobs.njobs), dim3(TK_THREADS), 0, st, logit, jobs, ws);
  set_stage2_attr();
  int mmax = 0;
  for (int s = 0; s <= L; ++s) mmax = mmax > jobs.seg_jobs[s] * jobs.seg_k[s] ? mmax : jobs.seg_jobs[s] * jobs.seg_k[s];
  APE_LAUNCH(proposal_topk2_kernel, dim3(L + 1), dim3(TK_THREADS), (size_t)mmax * sizeof(u64), st, ws, jobs, lv, k, k_alt, cand, alt);
  APE_CHECK_LAUNCH("proposal_topk");
  return 0;
}

extern "C" int ape_hip_proposal_order(const int32_t* cand, int n, const float* logit, const float* xyxy, const int* level_start,
                                      const int* level_n, int L, float* boxes_b, int32_t* groups_b, int32_t* seg, int32_t* cand_a,
                                      int32_t* lv_a, int32_t* pos_b, void* stream) {
  TkLevels lv;
  if (fill_levels(lv, level_start, level_n, L)) return -1;
  APE_CHECK_ARG(cand && logit && xyxy && boxes_b && groups_b && seg && cand_a && lv_a && pos_b && n > 0 && n <= PO_PAD,
                "ape_hip_proposal_order: 1 <= n <= 8192 candidates");
  static ApeOncePerDevice attr_done;
  if (attr_done.first()) {   // 64 KiB of dynamic LDS next to the kernel's static arrays needs the opt-in attribute
    (void)hipFuncSetAttribute((const void*)proposal_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PO_PAD * sizeof(u64));
  }
  APE_LAUNCH(proposal_order_kernel, dim3(1), dim3(TK_THREADS), (size_t)2 * ((n + 1023) / 1024) * 1024 * sizeof(u64), (hipStream_t)stream, cand, n, logit, xyxy, lv,
                     boxes_b, groups_b, seg, cand_a, lv_a, pos_b);
  APE_CHECK_LAUNCH("proposal_order_kernel");
  return 0;
}

extern "C" int ape_hip_proposal_quota(const int32_t* cand_a, const int32_t* lv_a, const int32_t* pos_b, const uint8_t* keep_b, int n,
                                      const int32_t* alt, int n_alt, const int* level_start, const int* level_n, int L, int nq,
                                      int64_t* out, void* stream) {
  TkLevels lv;
  if (fill_levels(lv, level_start, level_n, L)) return -1;
  APE_CHECK_ARG(cand_a && lv_a && pos_b && keep_b && alt && out && n > 0 && n <= PO_PAD && n_alt >= 0 && nq > 0,
                "ape_hip_proposal_quota: bad arguments (n <= 8192)");
  APE_LAUNCH(proposal_quota_kernel, dim3(1), dim3(TK_THREADS), 0, (hipStream_t)stream, cand_a, lv_a, pos_b, keep_b, n, alt, n_alt,
                     lv, nq, (long long*)out);
  APE_CHECK_LAUNCH("proposal_quota_kernel");
  return 0;
}

extern "C" int ape_hip_det_sort(const float* logits, int ldl, int Q, int K, const float* boxes, const float* scale, float thresh,
                                float* xyxy, uint8_t* finite, float* sorted, int32_t* order, uint8_t* valid, void* stream) {
  APE_CHECK_ARG(logits && boxes && scale && xyxy && finite && sorted && order && valid && K > 0, "ape_hip_det_sort: bad arguments");
  APE_CHECK_ARG(Q > 0 && Q <= 1024, "ape_hip_det_sort: 1 <= queries <= 1024 (got %d)", Q);
  hipStream_t st = (hipStream_t)stream;
  APE_LAUNCH(det_boxes_kernel, dim3(Q), dim3(64), 0, st, logits, ldl, K, boxes, scale, xyxy, finite);
  APE_LAUNCH(class_sort_kernel, dim3(K), dim3(TK_THREADS), 0, st, logits, ldl, Q, finite, thresh, sorted, order, valid);
  APE_CHECK_LAUNCH("ape_hip_det_sort");
  return 0;
}

extern "C" int ape_hip_det_topk(const float* sorted, const uint8_t* keep, const int32_t* order, const float* xyxy, int K, int Q, int k,
                                uint64_t* workspace, float* det_boxes, float* det_scores, int64_t* det_classes, int64_t* det_query,
                                void* stream) {
  APE_CHECK_ARG(sorted && keep && order && xyxy && workspace && det_boxes && det_scores && det_classes && det_query, "ape_hip_det_topk: null pointer");
  APE_CHECK_ARG(K > 0 && Q > 0 && (long long)K * Q < 0x7fffffffLL && k >= 1 && k <= 1024 && k <= K * Q, "ape_hip_det_topk: 1 <= k <= min(1024, K*Q)");
  const int total = K * Q;
  int per = topk_per(&total, &k, 1);
  const bool merge = per == 0;           // (lists x k) beyond stage 2's LDS at every chunk size: 64 k-element chunks + merge levels
  if (merge) per = 64;
  int nchunks = ceil_div(total, per * TK_THREADS);
  hipStream_t st = (hipStream_t)stream;
  u64* ws = reinterpret_cast<u64*>(workspace);
  if (per == 8) APE_LAUNCH(det_topk1_kernel<8>, dim3(nchunks), dim3(TK_THREADS), 0, st, sorted, keep, total, k, ws);
  else if (per == 32) APE_LAUNCH(det_topk1_kernel<32>, dim3(nchunks), dim3(TK_THREADS), 0, st, sorted, keep, total, k, ws);
  else APE_LAUNCH(det_topk1_kernel<64>, dim3(nchunks), dim3(TK_THREADS), 0, st, sorted, keep, total, k, ws);
  set_stage2_attr();
  if (merge) {
    // ping-pong between the chunk lists and the spare half of the workspace until one stage-2 workgroup can hold what is left
    const int cap_lists = ape_hip_topk_workspace_words(total) / 1024;
    u64* a = ws;
    u64* b = ws + (size_t)1024 * (cap_lists - (cap_lists / 3));      // the spare third starts behind the stage-1 lists
    const int group = TK_STAGE2_MAX / k;                              // >= 12 lists per merge workgroup (k <= 1024)
    while ((long long)nchunks * k > TK_STAGE2_MAX) {
      const int nout = ceil_div(nchunks, group);
      APE_LAUNCH(topk_merge_kernel, dim3(nout), dim3(TK_THREADS), (size_t)group * k * sizeof(u64), st, a, nchunks, group, k, b);
      u64* t = a; a = b; b = t;
      nchunks = nout;
    }
    ws = a;
  }
  APE_LAUNCH(det_topk2_kernel, dim3(1), dim3(TK_THREADS), (size_t)nchunks * k * sizeof(u64), st, ws, nchunks, order, xyxy, Q, k,
                     det_boxes, det_scores, (long long*)det_classes, (long long*)det_query);
  APE_CHECK_LAUNCH("det_topk");
  return 0;
}
