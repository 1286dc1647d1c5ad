// mel.hip — log-mel front end on the GPU: periodic-Hann STFT (n_fft 400, hop 160)
// as a direct 400-point real DFT in fp32, power spectrum, slaney filterbank from the model file,
// log10, per-window max, clamp to max-8, (x+4)/4.  Output: fp32 [B][n_mels][3000] (API / tests) and
// the 16-bit time-major image [B][3002][128] that conv1's GEMM reads as overlapping rows.
// HBM-bound on paper (1.92 MB in, 1.54 MB out per window) but tiny: < 1 % of a window's time.
#include "kernels.hpp"

namespace ohw {

constexpr int MEL_FR = 8;       // frames per workgroup
constexpr int MEL_THREADS = 256;

__device__ __forceinline__ int ordered_bits(float v) {
  int k = __float_as_int(v);
  return k >= 0 ? k : k ^ 0x7fffffff;
}
__device__ __forceinline__ float from_ordered_bits(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ void mel_init_kernel(int32_t* max_bits, int batch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) max_bits[i] = (int32_t)0x80000000;
}

__global__ __launch_bounds__(MEL_THREADS) void mel_power_kernel(MelParams p) {
  __shared__ float fr[N_FFT][MEL_FR];      // windowed frames, frame index fastest (broadcast reads)
  __shared__ float tw[2][N_FFT];
  __shared__ float pw[MEL_FR][N_FREQ + 7];
  __shared__ float red[MEL_THREADS / 64];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * MEL_FR;
  const int tid = threadIdx.x;
  const bool rec = p.offsets != nullptr;
  const int64_t off = rec ? p.offsets[b] : 0;
  const int64_t n = rec ? p.n_total : (p.n_samples[b] < CHUNK_SAMPLES ? p.n_samples[b] : CHUNK_SAMPLES);
  const float* pcm = rec ? p.pcm : p.pcm + (int64_t)b * p.pcm_stride;
  for (int i = tid; i < N_FFT; i += MEL_THREADS) { tw[0][i] = p.twiddle[i]; tw[1][i] = p.twiddle[N_FFT + i]; }
  for (int i = tid; i < N_FFT * MEL_FR; i += MEL_THREADS) {
    const int f = i / N_FFT, j = i % N_FFT;
    const int t = t0 + f;
    int64_t pos = (int64_t)t * HOP - N_FFT / 2 + j;   // index into the zero-padded 30 s signal
    float v = 0.0f;
    if (t < CHUNK_FRAMES) {
      if (rec) {
        pos += off;                                                       // index into the recording
        if (pos < 0) pos = -pos;                                          // reflected only at the recording's start
      } else {
        if (pos < 0) pos = -pos;                                          // reflect at the start
        else if (pos >= CHUNK_SAMPLES) pos = p.mode == OHW_MEL_REFLECT ? 2 * (int64_t)(CHUNK_SAMPLES - 1) - pos : -1;
      }
      if (pos >= 0 && pos < n) v = pcm[pos];
    }
    fr[j][f] = v * p.window[j];
  }
  __syncthreads();
  if (tid < N_FREQ) {
    float re[MEL_FR], im[MEL_FR];
#pragma unroll
    for (int f = 0; f < MEL_FR; ++f) { re[f] = 0.f; im[f] = 0.f; }
    int idx = 0;
    for (int j = 0; j < N_FFT; ++j) {
      const float c = tw[0][idx], s = tw[1][idx];
      const f32x4 x0 = *(const f32x4*)&fr[j][0];
      const f32x4 x1 = *(const f32x4*)&fr[j][4];
      re[0] += x0.x * c; im[0] -= x0.x * s; re[1] += x0.y * c; im[1] -= x0.y * s;
      re[2] += x0.z * c; im[2] -= x0.z * s; re[3] += x0.w * c; im[3] -= x0.w * s;
      re[4] += x1.x * c; im[4] -= x1.x * s; re[5] += x1.y * c; im[5] -= x1.y * s;
      re[6] += x1.z * c; im[6] -= x1.z * s; re[7] += x1.w * c; im[7] -= x1.w * s;
      idx += tid; if (idx >= N_FFT) idx -= N_FFT;
    }
#pragma unroll
    for (int f = 0; f < MEL_FR; ++f) pw[f][tid] = re[f] * re[f] + im[f] * im[f];
  }
  __syncthreads();
  float lmax = -INFINITY;
  for (int o = tid; o < p.n_mels * MEL_FR; o += MEL_THREADS) {
    const int f = o % MEL_FR, j = o / MEL_FR;
    const int t = t0 + f;
    if (t >= CHUNK_FRAMES) continue;
    const float* fj = p.filters + (int64_t)j * N_FREQ;
    float acc = 0.f;
    for (int k = 0; k < N_FREQ; ++k) acc += fj[k] * pw[f][k];
    const float lv = log10f(fmaxf(acc, 1e-10f));
    if (!p.max_only) p.logmel[((int64_t)b * p.n_mels + j) * CHUNK_FRAMES + t] = lv;
    lmax = fmaxf(lmax, lv);
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0) red[tid >> 6] = lmax;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int i = 1; i < MEL_THREADS / 64; ++i) m = fmaxf(m, red[i]);
    if (m > -INFINITY) atomicMax(&p.max_bits[p.shared_max ? 0 : b], ordered_bits(m));
  }
}

// clamp / scale in place and write the time-major 16-bit image (rows 1..3000; pad rows stay zero)
template <typename T>
__global__ __launch_bounds__(256) void mel_normalize_kernel(MelParams p) {
  __shared__ float tile[128][65];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  const float mx = from_ordered_bits(p.max_bits[p.shared_max ? 0 : b]);
  const float floor_v = mx - 8.0f;
  for (int i = tid; i < p.n_mels * 64; i += 256) {
    const int c = i / 64, tt = i % 64;
    const int t = t0 + tt;
    float v = 0.f;
    if (t < CHUNK_FRAMES) {
      float* ptr = p.logmel + ((int64_t)b * p.n_mels + c) * CHUNK_FRAMES + t;
      v = fmaxf(*ptr, floor_v);
      v = (v + 4.0f) * 0.25f;
      *ptr = v;
    }
    tile[c][tt] = v;
  }
  __syncthreads();
  T* img = (T*)p.mel_t + (int64_t)b * MEL_ROWS * MEL_CPAD;
  for (int i = tid; i < 64 * (MEL_CPAD / 2); i += 256) {
    const int tt = i / (MEL_CPAD / 2), c2 = (i % (MEL_CPAD / 2)) * 2;
    const int t = t0 + tt;
    if (t >= CHUNK_FRAMES) continue;
    const float v0 = c2 < p.n_mels ? tile[c2][tt] : 0.f;
    const float v1 = c2 + 1 < p.n_mels ? tile[c2 + 1][tt] : 0.f;
    *(unsigned*)(img + (int64_t)(1 + t) * MEL_CPAD + c2) = pack2<T>(v0, v1);
  }
}

template <typename T>
void launch_mel(const MelParams& p, hipStream_t s) {
  if (!p.shared_max) hipLaunchKernelGGL(mel_init_kernel, dim3((p.batch + 63) / 64), dim3(64), 0, s, p.max_bits, p.batch);
  hipLaunchKernelGGL(mel_power_kernel, dim3((CHUNK_FRAMES + MEL_FR - 1) / MEL_FR, p.batch), dim3(MEL_THREADS), 0, s, p);
  if (!p.max_only) hipLaunchKernelGGL((mel_normalize_kernel<T>), dim3((CHUNK_FRAMES + 63) / 64, p.batch), dim3(256), 0, s, p);
  HIP_CHECK(hipGetLastError());
}
template void launch_mel<bf16_t>(const MelParams&, hipStream_t);
template void launch_mel<f16_t>(const MelParams&, hipStream_t);

}  // namespace ohw
