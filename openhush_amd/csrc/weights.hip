// weights.hip — procedural weight fill and repacking into the engine's HBM layouts (gfx950).
// The generator restates openhush_amd/synth.py (and oracle/whisper_ref.c) bit for bit.
#include "kernels.hpp"

namespace ohw {

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

__global__ void synth_fill_kernel(float* __restrict__ dst, int64_t n, uint32_t key, float scale, float offset, int round_f16) {
  // one rounding per operation, like numpy: no mul+add contraction.  The operators are written here
  // (not through HIP's __f*_rn helpers, which are inlined WITH their own contract flags).
#pragma clang fp contract(off)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = fmix32(((uint32_t)i * 0x9E3779B1u) ^ key);
    const float u = (float)(h >> 8) * 0x1p-23f - 1.0f;   // exact
    float v = u * scale;
    if (offset != 0.0f) v = v + offset;
    if (round_f16) v = (float)(_Float16)v;
    dst[i] = v;
  }
}
void launch_synth_fill(float* dst, int64_t n, uint32_t key, float scale, float offset, int round_f16, hipStream_t s) {
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(synth_fill_kernel, dim3(blocks), dim3(256), 0, s, dst, n, key, scale, offset, round_f16);
  HIP_CHECK(hipGetLastError());
}

__global__ void f16_to_f32_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
void launch_f16_to_f32(const void* src, float* dst, int64_t n, hipStream_t s) {
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(blocks), dim3(256), 0, s, (const _Float16*)src, dst, n);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ void convert_rows_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t rows, int64_t cols, int64_t ld_dst) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    dst[r * ld_dst + c] = (T)src[i];
  }
}
template <typename T>
void launch_convert_rows(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_dst, hipStream_t s) {
  const int64_t n = rows * cols;
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL((convert_rows_kernel<T>), dim3(blocks), dim3(256), 0, s, src, (T*)dst, rows, cols, ld_dst);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ void repack_conv_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t d_out, int64_t c_in, int64_t c_pad) {
  const int64_t n = d_out * 3 * c_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i % c_pad, k = (i / c_pad) % 3, o = i / (3 * c_pad);
    dst[i] = c < c_in ? (T)src[(o * c_in + c) * 3 + k] : (T)0.0f;
  }
}
template <typename T>
void launch_repack_conv(const float* src, void* dst, int64_t d_out, int64_t c_in, int64_t c_pad, hipStream_t s) {
  const int64_t n = d_out * 3 * c_pad;
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL((repack_conv_kernel<T>), dim3(blocks), dim3(256), 0, s, src, (T*)dst, d_out, c_in, c_pad);
  HIP_CHECK(hipGetLastError());
}

// element (n, k) -> ((n/16 * (K/32) + k/32) * 64 + (n%16) + 16*((k%32)/8)) * 8 + k%8
template <typename T>
__global__ void repack_tiled_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t N, int64_t n_pad, int64_t K) {
  const int64_t n_el = n_pad * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (int64_t)gridDim.x * blockDim.x) {
    // i indexes the destination
    const int64_t j = i & 7, lane = (i >> 3) & 63, blk = i >> 9;
    const int64_t kb = blk % (K / 32), nt = blk / (K / 32);
    const int64_t n = nt * 16 + (lane & 15), k = kb * 32 + (lane >> 4) * 8 + j;
    dst[i] = n < N ? (T)src[n * K + k] : (T)0.0f;
  }
}
template <typename T>
void launch_repack_tiled(const float* src, void* dst, int64_t N, int64_t n_pad, int64_t K, hipStream_t s) {
  const int64_t n = n_pad * K;
  int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL((repack_tiled_kernel<T>), dim3(blocks), dim3(256), 0, s, src, (T*)dst, N, n_pad, K);
  HIP_CHECK(hipGetLastError());
}

// wsum[n] = sum_k W[n][k] over the ROUNDED 16-bit values of a fragment-tiled matrix (what the MFMA multiplies): the
// decoder's post-norm GEMMs compute rstd * (x W^T - mean * wsum) instead of normalising x first (decode.hip)
template <typename T>
__global__ __launch_bounds__(256) void tiled_rowsum_kernel(const T* __restrict__ w, float* __restrict__ wsum, int64_t N, int64_t K) {
  __shared__ float red[4];
  const int64_t n = blockIdx.x;
  const int64_t kblocks = K / 32;
  float acc = 0.f;
  for (int64_t k = threadIdx.x; k < K; k += 256)
    acc += (float)w[((((n >> 4) * kblocks + (k >> 5)) * 64 + (n & 15) + 16 * ((k & 31) >> 3)) << 3) + (k & 7)];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) wsum[n] = (red[0] + red[1]) + (red[2] + red[3]);
}
template <typename T>
void launch_tiled_rowsum(const void* w, float* wsum, int64_t N, int64_t K, hipStream_t s) {
  hipLaunchKernelGGL((tiled_rowsum_kernel<T>), dim3((unsigned)N), dim3(256), 0, s, (const T*)w, wsum, N, K);
  HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void fold_ln_kernel(float* __restrict__ w, float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int64_t K) {
  __shared__ float red[4];
  float* row = w + (int64_t)blockIdx.x * K;
  float acc = 0.f;
  for (int64_t k = threadIdx.x; k < K; k += 256) acc += beta[k] * row[k];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) bias[blockIdx.x] += (red[0] + red[1]) + (red[2] + red[3]);
  for (int64_t k = threadIdx.x; k < K; k += 256) row[k] *= gamma[k];
}
void launch_fold_ln(float* w, float* bias, const float* gamma, const float* beta, int64_t N, int64_t K, hipStream_t s) {
  hipLaunchKernelGGL(fold_ln_kernel, dim3((unsigned)N), dim3(256), 0, s, w, bias, gamma, beta, K);
  HIP_CHECK(hipGetLastError());
}

#define INST(T) \
  template void launch_convert_rows<T>(const float*, void*, int64_t, int64_t, int64_t, hipStream_t); \
  template void launch_repack_conv<T>(const float*, void*, int64_t, int64_t, int64_t, hipStream_t);  \
  template void launch_repack_tiled<T>(const float*, void*, int64_t, int64_t, int64_t, hipStream_t); \
  template void launch_tiled_rowsum<T>(const void*, float*, int64_t, int64_t, hipStream_t);
INST(bf16_t)
INST(f16_t)
#undef INST

}  // namespace ohw
