// Shared device/host helpers for the ape_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint16_t bf16_t;  // raw bfloat16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define APE_DT_F32 0
#define APE_DT_BF16 1
#define APE_DT_F16 2
typedef _Float16 f16_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even via gfx950's v_cvt_pk_bf16_f32 (same rule as torch's float->bfloat16 cast; NaN stays NaN)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

template <> __device__ __forceinline__ float ldf<f16_t>(const f16_t* p) { return (float)*p; }
// every float -> IEEE half store SATURATES at +-65504 (v_med3_f32; a NaN stays a NaN): the half flavour of the pipeline
// (GEMM operands, activations) must not turn an outlier into an infinity that the next contraction spreads over a row
#define APE_HALF_MAX 65504.f
__device__ __forceinline__ float sat_h(float v) { return __builtin_amdgcn_fmed3f(v, -APE_HALF_MAX, APE_HALF_MAX); }
template <> __device__ __forceinline__ void stf<f16_t>(f16_t* p, float v) { *p = (f16_t)sat_h(v); }

// 4 consecutive elements (p must be aligned to 4 elements)
template <typename T> __device__ __forceinline__ void ld4(const T* p, float v[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float v[4]) {
  float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float v[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float v[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float v[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float v[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}
// 8 consecutive elements (p aligned to 8 elements)
template <typename T> __device__ __forceinline__ void ld8(const T* p, float v[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float v[8]) {
  ld4<float>(p, v); ld4<float>(p + 4, v + 4);
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float v[8]) {
  uint4 t = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
  v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
template <> __device__ __forceinline__ void ld8<f16_t>(const f16_t* p, float v[8]) {
  const f16x8_t t = *reinterpret_cast<const f16x8_t*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float v[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float v[8]) {
  st4<float>(p, v); st4<float>(p + 4, v + 4);
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float v[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]),
                                            pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
// round-to-nearest-even pair conversion (same rule as torch's float -> half cast)
__device__ __forceinline__ uint32_t pack2h(float a, float b) {
  const f32x2_t v = {sat_h(a), sat_h(b)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
template <> __device__ __forceinline__ void ld4<f16_t>(const f16_t* p, float v[4]) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  const f16x2_t a = __builtin_bit_cast(f16x2_t, t.x), b = __builtin_bit_cast(f16x2_t, t.y);
  v[0] = (float)a[0]; v[1] = (float)a[1]; v[2] = (float)b[0]; v[3] = (float)b[1];
}
template <> __device__ __forceinline__ void st4<f16_t>(f16_t* p, const float v[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3]));
}
template <> __device__ __forceinline__ void st8<f16_t>(f16_t* p, const float v[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack2h(v[0], v[1]), pack2h(v[2], v[3]), pack2h(v[4], v[5]), pack2h(v[6], v[7]));
}

// ---- the two 16-bit flavours of the pipeline: H = bf16_t (8 significant bits, fp32 range) | f16_t (11 bits, the reference's
// own evaluation dtype: tools/train_net.py:642 model.to(torch.float16)).  Same data movement, same MFMA rate
// (v_mfma_f32_16x16x32_{bf16,f16}); kernels are templated on H and every 16-bit operand of ONE call has the same H.
template <typename H> struct h16;
template <> struct h16<bf16_t> {
  static constexpr int dt = APE_DT_BF16;
  static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack2bf(a, b); }
  static __device__ __forceinline__ f32x4_t mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct h16<f16_t> {
  static constexpr int dt = APE_DT_F16;
  static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack2h(a, b); }
  static __device__ __forceinline__ f32x4_t mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {   // fragments travel as raw 16 bytes
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
// fp32 dot product of two packed 16-bit pairs on top of c (v_dot2c_f32_bf16 / v_dot2c_f32_f16): a.lo * b.lo + a.hi * b.hi + c
template <typename H> __device__ __forceinline__ float dot2acc(uint32_t a, uint32_t b, float c);
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <> __device__ __forceinline__ float dot2acc<bf16_t>(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
template <> __device__ __forceinline__ float dot2acc<f16_t>(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
}
// the same instruction as a volatile asm statement (acc += a . b): stays where it is written -- the builtin is pure arithmetic that
// hipcc sinks to its first use, e.g. out of a pipelined main loop's phase and behind the barriers that delimit it
template <typename H> __device__ __forceinline__ void dot2acc_pinned(float& acc, uint32_t a, uint32_t b);
template <> __device__ __forceinline__ void dot2acc_pinned<bf16_t>(float& acc, uint32_t a, uint32_t b) {
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void dot2acc_pinned<f16_t>(float& acc, uint32_t a, uint32_t b) {
  asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}
template <typename H> struct ones2;                       // (1, 1) as a packed pair
template <> struct ones2<bf16_t> { static constexpr uint32_t v = 0x3f803f80u; };
template <> struct ones2<f16_t> { static constexpr uint32_t v = 0x3c003c00u; };
// two packed 16-bit values -> two floats
template <typename H> __device__ __forceinline__ void unpack2(uint32_t u, float& a, float& b);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t u, float& a, float& b) {
  a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t u, float& a, float& b) {
  const f16x2_t t = __builtin_bit_cast(f16x2_t, u);
  a = (float)t[0]; b = (float)t[1];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host side error plumbing (C-ABI: 0 = ok, negative = error, message via ape_hip_last_error) ----
extern "C" const char* ape_hip_last_error(void);
void ape_set_error(const char* fmt, ...);

// ---- launches.  Every kernel of the library goes out through APE_LAUNCH: the plain triple-chevron launch, or -- while the calling
// thread has metering on (meter.cpp, ape_hip_meter_begin) -- hipExtLaunchKernelGGL with the launch's own (start, stop) events, whose
// elapsed time is the dispatch's begin-to-end duration (what rocprofv3's kernel trace reports).  KERNEL is stringified as the launch's
// name in the meter's records: the template expression as written at the launch site.
#include <hip/hip_ext.h>
hipEvent_t* ape_meter_pair(const char* kernel, hipStream_t stream);
#define APE_LAUNCH(KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                                                  \
  do {                                                                                                                     \
    hipEvent_t* ev__ = ape_meter_pair(#KERNEL, STREAM);                                                                           \
    if (ev__ != nullptr) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, ev__[0], ev__[1], 0, __VA_ARGS__);        \
    else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                                                \
  } while (0)

#define APE_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ape_set_error(__VA_ARGS__);           \
      return -1;                            \
    }                                       \
  } while (0)
#define APE_CHECK_LAUNCH(name)                                                    \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      ape_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
      return -2;                                                                  \
    }                                                                             \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// ---- per-DEVICE launcher state: function attributes (hipFuncAttributeMaxDynamicSharedMemorySize) and the CU count belong to
// the device that is current at the launch, not to the first one a process happened to use
constexpr int APE_MAX_DEVICES = 64;
static inline int ape_current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= APE_MAX_DEVICES) d = 0;
  return d;
}
struct ApeOncePerDevice {          // `static ApeOncePerDevice f; if (f.first()) { ...set attributes... }`
  bool done[APE_MAX_DEVICES] = {};
  bool first() { const int d = ape_current_device(); const bool f = !done[d]; done[d] = true; return f; }
};
static inline int ape_cu_count() {   // compute units of the current device, rounded down to a multiple of 8 (XCD-aware tile orders); 256 if unknown
  static int n[APE_MAX_DEVICES] = {};
  const int d = ape_current_device();
  if (n[d] == 0) {
    int c = 0;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c < 8) c = 256;
    n[d] = c & ~7;
  }
  return n[d];
}
static inline bool ape_is16(int dt) { return dt == APE_DT_BF16 || dt == APE_DT_F16; }
// the 16-bit storage kind of one call: 0 = none of the dtypes is 16-bit, APE_DT_BF16 / APE_DT_F16, -1 = mixed or unknown
static inline int ape_h16_kind(const int* dts, int n) {
  int k = 0;
  for (int i = 0; i < n; ++i) {
    const int d = dts[i];
    if (d == APE_DT_F32) continue;
    if (!ape_is16(d)) return -1;
    if (k != 0 && k != d) return -1;
    k = d;
  }
  return k;
}
#define APE_H16_KIND(...) ([&]() { const int d__[] = {__VA_ARGS__}; return ape_h16_kind(d__, (int)(sizeof(d__) / sizeof(int))); }())
