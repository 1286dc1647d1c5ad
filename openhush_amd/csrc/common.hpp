// common.hpp — shared host/device helpers for libohw (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <set>
#include <utility>
#include <stdexcept>
#include <string>

#include "../../include/ohw.h"

namespace ohw {

// Process-wide gate around stream capture.  While ANY stream of the process captures, HIP (ROCm 7.2) rejects what other
// threads do in the meantime - a kernel launch on a CU-masked (blocking) stream, a synchronous hipMemset - with
// "operation would make the legacy stream depend on a capturing blocking stream", and that error also invalidates the
// capture.  Two engines driven by two threads hit it (tests/test_gpu_configs.py).  Every C-ABI entry that touches the
// device holds the gate shared (outermost call of the thread only: entries nest); the one place that captures a graph -
// once per (state, batch, sampler parameters) - takes it exclusively for the few milliseconds the capture lasts.
struct ApiScope {
  ApiScope();
  ~ApiScope();
  ApiScope(const ApiScope&) = delete;
  ApiScope& operator=(const ApiScope&) = delete;
};
struct ApiRelease {    // inside an ApiScope: gives the gate up while this thread only WAITS for other threads of the library
  ApiRelease();        // (a lane that captures a graph needs the gate exclusively; a waiting holder would block it for ever)
  ~ApiRelease();
  ApiRelease(const ApiRelease&) = delete;
  ApiRelease& operator=(const ApiRelease&) = delete;
};
struct CaptureGate {   // inside an ApiScope: shared -> exclusive for the lifetime of the object
  CaptureGate();
  ~CaptureGate();
  CaptureGate(const CaptureGate&) = delete;
  CaptureGate& operator=(const CaptureGate&) = delete;
};

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
  if (e != hipSuccess) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    int code = (e == hipErrorOutOfMemory) ? OHW_E_OOM : (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? OHW_E_NO_GPU : OHW_E_TRANSCRIBE;
    throw Error(code, buf);
  }
}
#define HIP_CHECK(x) ::ohw::hip_check((x), #x, __FILE__, __LINE__)

// Raise a kernel's dynamic-LDS limit once per (kernel, device): function attributes are per device, and one
// process may own engines on several GPUs.
inline void ensure_dynamic_lds(const void* func, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  hip_check(hipGetDevice(&dev), "hipGetDevice", __FILE__, __LINE__);
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({func, dev})) return;
  hip_check(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "hipFuncSetAttribute", __FILE__, __LINE__);
  done.insert({func, dev});
}

// 16-bit storage types of the compute path
using bf16_t = __bf16;
using f16_t = _Float16;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename T> struct TypeOps;
template <> struct TypeOps<bf16_t> {
  using vec8 = bf16x8;
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct TypeOps<f16_t> {
  using vec8 = f16x8;
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename T> using vec8_t = typename TypeOps<T>::vec8;

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// pack two floats into one 32-bit word of two T (round to nearest even)
template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) T v2;
  v2 p; p.x = (T)lo; p.y = (T)hi;
  return __builtin_bit_cast(unsigned, p);
}
template <typename T> __device__ __forceinline__ void unpack2(unsigned w, float& lo, float& hi) {
  typedef __attribute__((ext_vector_type(2))) T v2;
  v2 p = __builtin_bit_cast(v2, w);
  lo = (float)p.x; hi = (float)p.y;
}

// exact-erf GELU of the published model.  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32
// noise), branch-free: the library erff is piecewise with divergent branches and cost 25 % of the mlp.0
// GEMM when fused into its epilogue.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
  const float erf_abs = 1.0f - poly * e;
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// Activation operand of the decoder's skinny GEMMs (decode.hip): 16-row tiles in MFMA B-fragment order,
// T [ceil(M/16)][K/32][64 lanes][8], lane = (m % 16) + 16 * ((k % 32) / 8).  A wave then loads one k-block of a
// tile as 1 KiB contiguous (8 whole cache lines, each read once) instead of 16 half-used lines 2*K bytes apart -
// with row-major activations those loads, not the weight stream, set the time of every such GEMM.
__device__ __forceinline__ int64_t act_tiled_offset(int m, int k, int K) {
  return ((((int64_t)(m >> 4) * (K >> 5) + (k >> 5)) * 64 + (m & 15) + 16 * ((k & 31) >> 3)) << 3) + (k & 7);
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

// XCD-aware bijective remap of a 1-D block id: blocks with consecutive logical ids share an XCD
// (cdna_hip_programming.md T1).  Speed only; never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  unsigned q = nwg / nx, r = nwg % nx, xcd = bid % nx, idx = bid / nx;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

constexpr int CHUNK_SAMPLES = 480000;
constexpr int CHUNK_FRAMES = 3000;
constexpr int N_FFT = 400;
constexpr int HOP = 160;
constexpr int N_FREQ = 201;
constexpr int MEL_CPAD = 128;   // channel padding of the time-major mel image (conv1 K = 3 * 128)
constexpr int MEL_ROWS = CHUNK_FRAMES + 2;  // one zero row before and after (conv padding)

}  // namespace ohw
