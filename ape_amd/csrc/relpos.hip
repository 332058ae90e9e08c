// Decomposed relative positions of the EVA-01 MIM ViT (ape/modeling/backbone/vit_eva.py:121-146, utils_eva.py:132-161) as EXTRA
// CHANNELS of the attention operands, so that the flash-attention kernel (attention.hip) needs no bias input:
//     attn[q, (kh, kw)] = scale q.k + q.Rh[qh - kh] + q.Rw[qw - kw]
//                       = q_ext . k_ext     with  q_ext = [scale q | q.Rh[qh - 0 .. qh - (Hk-1)] | q.Rw[qw - 0 ..] | 0],
//                                                 k_ext = [k       | one-hot(kh)                | one-hot(kw)     | 0]
// The dot products q.R* of a query with ALL 2 Hk - 1 (+ 2 Wk - 1) table rows come from one MFMA GEMM (rows = (token, head), the tables
// as the weight matrix: ape_amd/modeling/backbone/vit_eva.py); this kernel only GATHERS them into place, scales q and writes the
// one-hot key channels.  One wave per (token, head).
#include "common.h"
#include "../../include/ape_hip.h"

struct RelposParams {
  const void* q; const void* k; int ldqk;      // [rows, >= nh * hs] views: head h at columns h * hs .. + hd
  const void* t; int ldt; int tper;            // [rows * tper, >= 2 Hk - 1 + 2 Wk - 1]: q . [Rh ; Rw]^T, row token * tper + head
  const int* ty; const int* tx; int period;    // position of token (row % period) inside its attention group
  void* qe; void* ke; int lde;                 // [rows, nh * hdq]
  int rows, nh, hs, hd, Hk, Wk, hdq;
  float scale;
};

template <typename T>
__global__ __launch_bounds__(256) void relpos_extend_kernel(const RelposParams p) {
  const int lane = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (row, head)
  if (item >= (long long)p.rows * p.nh) return;
  const int row = (int)(item / p.nh), h = (int)(item % p.nh);
  const int y = p.ty[row % p.period], x = p.tx[row % p.period];
  const T* q = reinterpret_cast<const T*>(p.q) + (size_t)row * p.ldqk + h * p.hs;
  const T* k = reinterpret_cast<const T*>(p.k) + (size_t)row * p.ldqk + h * p.hs;
  const T* t = reinterpret_cast<const T*>(p.t) + ((size_t)row * p.tper + h) * p.ldt;
  T* qe = reinterpret_cast<T*>(p.qe) + (size_t)row * p.lde + h * p.hdq;
  T* ke = reinterpret_cast<T*>(p.ke) + (size_t)row * p.lde + h * p.hdq;
  for (int c = lane; c < p.hdq; c += 64) {
    float qv = 0.f, kv = 0.f;
    if (c < p.hd) {
      qv = ldf<T>(q + c) * p.scale;
      kv = ldf<T>(k + c);
    } else if (c < p.hd + p.Hk) {
      const int kh = c - p.hd;
      qv = ldf<T>(t + (y - kh + p.Hk - 1));
      kv = kh == y ? 1.f : 0.f;
    } else if (c < p.hd + p.Hk + p.Wk) {
      const int kw = c - p.hd - p.Hk;
      qv = ldf<T>(t + (2 * p.Hk - 1) + (x - kw + p.Wk - 1));
      kv = kw == x ? 1.f : 0.f;
    }
    stf<T>(qe + c, qv);
    stf<T>(ke + c, kv);
  }
}

extern "C" int ape_hip_relpos_extend(const void* q, const void* k, int ldqk, const void* t, int ldt, int tper, const int* ty, const int* tx, int period,
                                     void* q_ext, void* k_ext, int lde, int rows, int nh, int hs, int hd, int Hk, int Wk, int hdq, float scale,
                                     int dt, void* stream) {
  APE_CHECK_ARG(q && k && t && ty && tx && q_ext && k_ext && rows > 0 && nh > 0 && period > 0 && tper >= nh, "ape_hip_relpos_extend: null pointer / empty problem");
  APE_CHECK_ARG(hd > 0 && hd <= hs && Hk > 0 && Wk > 0 && hd + Hk + Wk <= hdq && ldt >= 2 * Hk - 1 + 2 * Wk - 1 && lde >= nh * hdq && ldqk >= nh * hs,
                "ape_hip_relpos_extend: hd %d + Hk %d + Wk %d must fit hdq %d; leading dimensions too small", hd, Hk, Wk, hdq);
  RelposParams p;
  p.q = q; p.k = k; p.ldqk = ldqk; p.t = t; p.ldt = ldt; p.tper = tper; p.ty = ty; p.tx = tx; p.period = period; p.qe = q_ext; p.ke = k_ext; p.lde = lde;
  p.rows = rows; p.nh = nh; p.hs = hs; p.hd = hd; p.Hk = Hk; p.Wk = Wk; p.hdq = hdq; p.scale = scale;
  const long long items = (long long)rows * nh;
  const dim3 grid((unsigned)((items + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dt == APE_DT_F16) APE_LAUNCH(relpos_extend_kernel<f16_t>, grid, block, 0, s, p);
  else if (dt == APE_DT_BF16) APE_LAUNCH(relpos_extend_kernel<bf16_t>, grid, block, 0, s, p);
  else if (dt == APE_DT_F32) APE_LAUNCH(relpos_extend_kernel<float>, grid, block, 0, s, p);
  else { ape_set_error("ape_hip_relpos_extend: dtype code %d", dt); return -1; }
  APE_CHECK_LAUNCH("ape_hip_relpos_extend");
  return 0;
}
