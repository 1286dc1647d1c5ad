// misc.hip — LayerNorm (fp32 residual stream in, 16-bit GEMM operand out) and small converters.
#include "kernels.hpp"

namespace ohw {

// one wave per row; two-pass mean / variance in registers (same arithmetic order class as the oracle)
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, int64_t rows, int d, int tiled) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * d;
  // unconditional (clamped) loads, masked arithmetic: see the note in decode.hip's fused LayerNorm
  f32x4 v[MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = *(const f32x4*)(xr + (c < d ? c : 0));
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float ok = c < d ? 1.f : 0.f;
    sum += ok * ((v[i].x + v[i].y) + (v[i].z + v[i].w));
  }
  const float mean = wave_sum(sum) / (float)d;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float ok = c < d ? 1.f : 0.f;
    const float a = v[i].x - mean, b2 = v[i].y - mean, c2 = v[i].z - mean, d2 = v[i].w - mean;
    var += ok * ((a * a + b2 * b2) + (c2 * c2 + d2 * d2));
  }
  const float rstd = rsqrtf(wave_sum(var) / (float)d + 1e-5f);
  T* yr = y + row * d;
  f32x4 g[MAXV], bb[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    g[i] = *(const f32x4*)(gamma + (c < d ? c : 0));
    bb[i] = *(const f32x4*)(beta + (c < d ? c : 0));
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    u32x2 w;
    w.x = pack2<T>((v[i].x - mean) * rstd * g[i].x + bb[i].x, (v[i].y - mean) * rstd * g[i].y + bb[i].y);
    w.y = pack2<T>((v[i].z - mean) * rstd * g[i].z + bb[i].z, (v[i].w - mean) * rstd * g[i].w + bb[i].w);
    if (c < d) *(u32x2*)(tiled ? y + act_tiled_offset((int)row, c, d) : yr + c) = w;
  }
}

template <typename T>
void launch_layernorm(const float* x, const float* gamma, const float* beta, void* y, int64_t rows, int d, hipStream_t s, bool tiled) {
  if (rows <= 0) return;
  if (tiled && d % 32 != 0) throw Error(OHW_E_INVALID_ARG, "layernorm: the tiled output needs d % 32 == 0");
  const int tl = tiled ? 1 : 0;
  if (d % 4 != 0 || d > 2048) throw Error(OHW_E_INVALID_ARG, "layernorm: d must be a multiple of 4 and <= 2048");
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  if (d <= 512) hipLaunchKernelGGL((layernorm_kernel<T, 2>), dim3(blocks), dim3(256), 0, s, x, gamma, beta, (T*)y, rows, d, tl);
  else if (d <= 1024) hipLaunchKernelGGL((layernorm_kernel<T, 4>), dim3(blocks), dim3(256), 0, s, x, gamma, beta, (T*)y, rows, d, tl);
  else if (d <= 1280) hipLaunchKernelGGL((layernorm_kernel<T, 5>), dim3(blocks), dim3(256), 0, s, x, gamma, beta, (T*)y, rows, d, tl);
  else hipLaunchKernelGGL((layernorm_kernel<T, 8>), dim3(blocks), dim3(256), 0, s, x, gamma, beta, (T*)y, rows, d, tl);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ void to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
template <typename T>
void launch_to_f32(const void* src, float* dst, int64_t n, hipStream_t s) {
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL((to_f32_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)src, dst, n);
  HIP_CHECK(hipGetLastError());
}

#define INST(T) \
  template void launch_layernorm<T>(const float*, const float*, const float*, void*, int64_t, int, hipStream_t, bool); \
  template void launch_to_f32<T>(const void*, float*, int64_t, hipStream_t);
INST(bf16_t)
INST(f16_t)
#undef INST

}  // namespace ohw
