// Iterative box refinement of the deformable decoder (deformable_transformer_vl.py:232-246, with_box_refine):
//     new_ref = sigmoid(delta + inverse_sigmoid(ref))          detrex.utils.inverse_sigmoid, eps = 1e-3
//     ref_in  = new_ref[:, None, :] * cat(valid_ratios, valid_ratios)[None]   (:203-210, the next layer's MSDA reference)
// One tiny kernel instead of nine elementwise launches per decoder layer (the decoder is launch-latency bound).
#include "common.h"
#include "../../include/ape_hip.h"

__global__ __launch_bounds__(256) void box_refine_kernel(const float* __restrict__ delta, int ldd, const float* __restrict__ ref,
                                                         const float* __restrict__ vr4, int L, int Q, float eps,
                                                         float* __restrict__ new_ref, float* __restrict__ ref_in) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= Q * 4) return;
  const int q = gid >> 2, c = gid & 3;
  float x = ref[gid];
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
  float z = logf(x1 / x2);
  if (delta != nullptr) z += delta[(size_t)q * ldd + c];
  const float r = delta != nullptr ? 1.f / (1.f + expf(-z)) : ref[gid];
  if (new_ref != nullptr) new_ref[gid] = r;
  if (ref_in != nullptr)
    for (int l = 0; l < L; ++l) ref_in[((size_t)q * L + l) * 4 + c] = r * vr4[l * 4 + c];
}

// delta [Q,4] fp32 (row stride ldd) or NULL (then new_ref = ref: only the per-level reference is produced),
// ref [Q,4], vr4 [L,4] -> new_ref [Q,4] (may be NULL), ref_in [Q,L,4] (may be NULL)
extern "C" int ape_hip_box_refine(const float* delta, int ldd, const float* ref, const float* vr4, int L, int Q, float eps,
                                  float* new_ref, float* ref_in, void* stream) {
  APE_CHECK_ARG(ref && Q > 0 && (ref_in == nullptr || (vr4 != nullptr && L > 0)) && (new_ref || ref_in),
                "ape_hip_box_refine: bad args");
  APE_LAUNCH(box_refine_kernel, dim3(ceil_div(Q * 4, 256)), dim3(256), 0, (hipStream_t)stream, delta, ldd, ref, vr4, L, Q, eps,
                     new_ref, ref_in);
  APE_CHECK_LAUNCH("ape_hip_box_refine");
  return 0;
}

// ------------------------------------------------------------------------------------------
// Query initialisation of the two-stage decoder (deformable_transformer_vl.py:412-420, 629-645).
//   query_init   : the selected proposals' unactivated boxes -> reference = sigmoid(coords) [Q,4] fp32 and their sine embedding
//                  pe [Q, 4 * P] (element (c, i): v = sigmoid(coord_c) * 2 pi / dim_t[i]; sin(v) for even i, cos(v) for odd i),
//                  stored in the pos_trans GEMM's operand type; also the int32 copy of the proposal indices the row gathers take.
//   query_finish : pos_trans_norm / pix_trans_norm (LayerNorm over 2E and E columns, two-pass variance) of the two GEMM outputs,
//                  split, add: query_pos = LN(pos)[:, :E], query = LN(pos)[:, E:] + LN(pix), and the first decoder layer's
//                  attention input query + query_pos (computed from the ROUNDED outputs, like the tensor-level expression).
// Two launches instead of ~20 elementwise launches of the tensor library.
// ------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float round_to<f16_t>(float v) { return (float)(f16_t)sat_h(v); }

template <typename TO>
__global__ __launch_bounds__(256) void query_init_kernel(const float* __restrict__ coords, const int64_t* __restrict__ topk, int T,
                                                         const float* __restrict__ dim_t, int P, float scale, int Q,
                                                         float* __restrict__ reference, TO* __restrict__ pe, int ldpe,
                                                         int32_t* __restrict__ topk32) {
  const int q = blockIdx.x;
  int64_t t = topk[q];
  t = t < 0 ? 0 : (t >= T ? T - 1 : t);
  if (threadIdx.x == 0 && topk32 != nullptr) topk32[q] = (int32_t)t;
  for (int j = threadIdx.x; j < 4 * P; j += 256) {
    const int c = j / P, i = j % P;
    const float x = coords[(size_t)t * 4 + c];
    const float s = 1.f / (1.f + expf(-x));
    if (i == 0) reference[q * 4 + c] = s;
    const float v = (s * scale) / dim_t[i];
    stf<TO>(pe + (size_t)q * ldpe + j, (i & 1) ? cosf(v) : sinf(v));
  }
}

extern "C" int ape_hip_query_init(const float* coords, const int64_t* topk, int T, const float* dim_t, int P, float scale, int Q,
                                  float* reference, void* pe, int ldpe, int pe_dt, int32_t* topk32, void* stream) {
  APE_CHECK_ARG(coords && topk && dim_t && reference && pe && T > 0 && P > 0 && Q > 0 && ldpe >= 4 * P, "ape_hip_query_init: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (pe_dt == APE_DT_F16) APE_LAUNCH(query_init_kernel<f16_t>, dim3(Q), dim3(256), 0, s, coords, topk, T, dim_t, P, scale, Q, reference, (f16_t*)pe, ldpe, topk32);
  else if (pe_dt == APE_DT_BF16) APE_LAUNCH(query_init_kernel<bf16_t>, dim3(Q), dim3(256), 0, s, coords, topk, T, dim_t, P, scale, Q, reference, (bf16_t*)pe, ldpe, topk32);
  else if (pe_dt == APE_DT_F32)
    APE_LAUNCH(query_init_kernel<float>, dim3(Q), dim3(256), 0, s, coords, topk, T, dim_t, P, scale, Q, reference, (float*)pe, ldpe, topk32);
  else
    APE_CHECK_ARG(false, "ape_hip_query_init: pe must be f32, bf16 or f16");
  APE_CHECK_LAUNCH("ape_hip_query_init");
  return 0;
}

// one wave per query row; E <= 512 (8 values per lane of the 2E-wide row, 4 of the E-wide row ... held in registers)
template <typename TO>
__global__ __launch_bounds__(256) void query_finish_kernel(const float* __restrict__ pos, int ldpos, const float* __restrict__ pix,
                                                           int ldpix, int Q, int E, const float* __restrict__ wpos,
                                                           const float* __restrict__ bpos, float eps_pos,
                                                           const float* __restrict__ wpix, const float* __restrict__ bpix,
                                                           float eps_pix, TO* __restrict__ query_pos, TO* __restrict__ query,
                                                           TO* __restrict__ query_sum, int ldo) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  constexpr int MAXP = 16, MAXX = 8;                    // 2E <= 1024, E <= 512
  float vp[MAXP], vx[MAXX];
  const float* pr = pos + (size_t)q * ldpos;
  const float* xr = pix + (size_t)q * ldpix;
  float sp = 0.f, sx = 0.f;
#pragma unroll
  for (int k = 0; k < MAXP; ++k) { const int c = lane + 64 * k; vp[k] = c < 2 * E ? pr[c] : 0.f; sp += vp[k]; }
#pragma unroll
  for (int k = 0; k < MAXX; ++k) { const int c = lane + 64 * k; vx[k] = c < E ? xr[c] : 0.f; sx += vx[k]; }
  const float mp = wave_sum(sp) / (float)(2 * E), mx = wave_sum(sx) / (float)E;
  float qp = 0.f, qx = 0.f;
#pragma unroll
  for (int k = 0; k < MAXP; ++k) { const int c = lane + 64 * k; const float d = c < 2 * E ? vp[k] - mp : 0.f; qp += d * d; }
#pragma unroll
  for (int k = 0; k < MAXX; ++k) { const int c = lane + 64 * k; const float d = c < E ? vx[k] - mx : 0.f; qx += d * d; }
  const float rp = rsqrtf(wave_sum(qp) / (float)(2 * E) + eps_pos), rx = rsqrtf(wave_sum(qx) / (float)E + eps_pix);
  // columns c < E of the normalised pos row are query_pos; column E + c pairs with pix column c: with 64-lane strides the pair
  // (E + c, c) lives in the same lane iff E % 64 == 0 (E = 256 in every configuration; checked by the launcher)
  const int kE = E / 64;
#pragma unroll
  for (int k = 0; k < MAXX; ++k) {
    const int c = lane + 64 * k;
    if (c < E) {
      const float a = (vp[k] - mp) * rp * wpos[c] + bpos[c];
      float hi = 0.f;
#pragma unroll
      for (int kk = 0; kk < MAXP; ++kk) if (kk == k + kE) hi = (vp[kk] - mp) * rp * wpos[E + c] + bpos[E + c];
      const float b = hi + ((vx[k] - mx) * rx * wpix[c] + bpix[c]);
      const float ar = round_to<TO>(a), br = round_to<TO>(b);
      stf<TO>(query_pos + (size_t)q * ldo + c, ar);
      stf<TO>(query + (size_t)q * ldo + c, br);
      stf<TO>(query_sum + (size_t)q * ldo + c, ar + br);
    }
  }
}

extern "C" int ape_hip_query_finish(const float* pos, int ldpos, const float* pix, int ldpix, int Q, int E, const float* wpos,
                                    const float* bpos, float eps_pos, const float* wpix, const float* bpix, float eps_pix,
                                    void* query_pos, void* query, void* query_sum, int ldo, int out_dt, void* stream) {
  APE_CHECK_ARG(pos && pix && wpos && bpos && wpix && bpix && query_pos && query && query_sum && Q > 0, "ape_hip_query_finish: bad args");
  APE_CHECK_ARG(E > 0 && E <= 512 && E % 64 == 0 && ldpos >= 2 * E && ldpix >= E && ldo >= E, "ape_hip_query_finish: E must be a multiple of 64, <= 512");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(ceil_div(Q, 4)), block(256);
  if (out_dt == APE_DT_F16) APE_LAUNCH(query_finish_kernel<f16_t>, grid, block, 0, s, pos, ldpos, pix, ldpix, Q, E, wpos, bpos, eps_pos, wpix, bpix, eps_pix,
                       (f16_t*)query_pos, (f16_t*)query, (f16_t*)query_sum, ldo);
  else if (out_dt == APE_DT_BF16) APE_LAUNCH(query_finish_kernel<bf16_t>, grid, block, 0, s, pos, ldpos, pix, ldpix, Q, E, wpos, bpos, eps_pos, wpix, bpix, eps_pix,
                       (bf16_t*)query_pos, (bf16_t*)query, (bf16_t*)query_sum, ldo);
  else if (out_dt == APE_DT_F32)
    APE_LAUNCH(query_finish_kernel<float>, grid, block, 0, s, pos, ldpos, pix, ldpix, Q, E, wpos, bpos, eps_pos, wpix, bpix, eps_pix,
                       (float*)query_pos, (float*)query, (float*)query_sum, ldo);
  else
    APE_CHECK_ARG(false, "ape_hip_query_finish: outputs must be f32, bf16 or f16");
  APE_CHECK_LAUNCH("ape_hip_query_finish");
  return 0;
}

// ------------------------------------------------------------------------------------------
// Detection records of one image (detector_postprocess, deformable_detr_segm_vl.py:857-872 + detectron2's
// Boxes.scale / clip / nonempty): boxes rescaled to the output frame and clipped, keep = score >= 0 and non-empty box,
// record row = (x1, y1, x2, y2, score or -1 when dropped, class, query, keep); KEPT ROWS FIRST, both groups in their
// original order (a stable partition -- the host then reads prefixes of the pinned buffers).  One workgroup; replaces
// ~25 elementwise / sort / gather launches of the tensor library per image.
// frame [8] = (sx, sy, sx, sy, width, height, width, height).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void det_records_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                            const int64_t* __restrict__ classes, const int64_t* __restrict__ query,
                                                            const float* __restrict__ frame, int k, float* __restrict__ rec,
                                                            float* __restrict__ boxes_out, int32_t* __restrict__ order) {
  __shared__ int wtot[16];
  __shared__ int total_keep;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = frame[i];
  // pass 1: how many rows are kept in total (the dropped group starts there)
  int cnt = 0;
  for (int i = tid; i < k; i += 1024) {
    float b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = fminf(fmaxf(boxes[i * 4 + c] * f[c], 0.f), f[4 + c]);
    cnt += (scores[i] >= 0.f && (b[2] - b[0]) > 0.f && (b[3] - b[1]) > 0.f) ? 1 : 0;
  }
  cnt = (int)wave_sum((float)cnt);
  if (lane == 0) wtot[wave] = cnt;
  __syncthreads();
  if (tid == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wtot[w]; total_keep = s; }
  __syncthreads();
  int kbase = 0, dbase = total_keep;
  for (int i0 = 0; i0 < k; i0 += 1024) {
    const int i = i0 + tid;
    const bool in = i < k;
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    bool keep = false;
    float sc = -1.f;
    if (in) {
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = fminf(fmaxf(boxes[i * 4 + c] * f[c], 0.f), f[4 + c]);
      sc = scores[i];
      keep = sc >= 0.f && (b[2] - b[0]) > 0.f && (b[3] - b[1]) > 0.f;
    }
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();                               // wtot of the previous chunk has been consumed
    if (lane == 0) wtot[wave] = __popcll(m);
    __syncthreads();
    int wbase = 0, chunk_keep = 0;
    for (int w = 0; w < 16; ++w) { const int t = wtot[w]; if (w < wave) wbase += t; chunk_keep += t; }
    if (in) {
      const int kept_before = wbase + before;                      // kept rows of this chunk in front of row i
      const int pos = keep ? kbase + kept_before : dbase + (tid - kept_before);
      float* r = rec + (size_t)pos * 8;
      r[0] = b[0]; r[1] = b[1]; r[2] = b[2]; r[3] = b[3];
      r[4] = keep ? sc : -1.f;
      r[5] = (float)classes[i];
      r[6] = (float)query[i];
      r[7] = keep ? 1.f : 0.f;
      float* bo = boxes_out + (size_t)pos * 4;
      bo[0] = b[0]; bo[1] = b[1]; bo[2] = b[2]; bo[3] = b[3];
      order[pos] = i;
    }
    const int nin = min(1024, k - i0);
    kbase += chunk_keep;
    dbase += nin - chunk_keep;
  }
}

extern "C" int ape_hip_det_records(const float* boxes, const float* scores, const int64_t* classes, const int64_t* query,
                                   const float* frame, int k, float* rec, float* boxes_out, int32_t* order, void* stream) {
  APE_CHECK_ARG(boxes && scores && classes && query && frame && rec && boxes_out && order && k > 0, "ape_hip_det_records: bad args");
  APE_LAUNCH(det_records_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, boxes, scores, classes, query, frame, k, rec,
                     boxes_out, order);
  APE_CHECK_LAUNCH("ape_hip_det_records");
  return 0;
}
