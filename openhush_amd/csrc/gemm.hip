// gemm.hip — 128x128x64 LDS-tiled MFMA GEMM for the encoder / cross-KV projections (gfx950).
//
// C[M][N] = A[M][K] * W[N][K]^T, both operands K-contiguous 16-bit (bf16 or f16), fp32 accumulate.
// MFMA v_mfma_f32_16x16x32 with SWAPPED operands: the MFMA "A" operand is the weight tile (rows = n),
// the "B" operand the activation tile (cols = m), so D[n][m] leaves each lane holding 4 consecutive n
// of ONE output row m; the four n-subtiles of a wave are interleaved (LDS row permutation at staging
// time) so a lane ends up with 16 contiguous n = 32 B (16-bit out) / 64 B (fp32 out) per row: wide,
// line-filling epilogue stores.
//
// Work split: 256 threads = 4 waves (2 along m x 2 along n), each wave a 64x64 output tile
// (4x4 MFMA tiles, 64 accumulator VGPRs).  LDS: 2 stages x (128x64 A + 128x64 W) x 2 B = 64 KiB,
// 16-byte chunks XOR-swizzled by (row & 7) so every ds_read_b128 fragment read is conflict-free.
// Global -> LDS goes through registers (loads of tile k+1 are issued before the MFMAs of tile k and
// written to the other LDS stage afterwards: one barrier per K-step).
// Roofline: MFMA-bound (arithmetic intensity = 64 flop/B at this tile).
#include "gemm.hpp"
#include "gemm_epilogue.hpp"

namespace ohw {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;

template <typename T, int EPI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(GemmParams p) {
  using Ops = TypeOps<T>;
  using vec8 = typename Ops::vec8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // stage s: A tile at s*32768, W tile at s*32768 + 16384
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const unsigned n_tiles_n = (unsigned)(p.N / BN);
  const unsigned n_tiles_m = (unsigned)((p.M + BM - 1) / BM);
  const unsigned nwg = n_tiles_n * n_tiles_m;
  const unsigned lid = xcd_remap(blockIdx.x, nwg);
  const int64_t m0 = (int64_t)(lid / n_tiles_n) * BM;
  const int64_t n0 = (int64_t)(lid % n_tiles_n) * BN;

  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;

  // ---- staging assignment: thread -> 4 rows x one 16-byte chunk, for each operand ----
  const int srow = tid >> 3;  // 0..31 (+32*i)
  const int chunk = tid & 7;
  const T* a_ptr[4];
  const T* w_ptr[4];
  int a_lds[4], w_lds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = srow + 32 * i;
    int64_t m = m0 + r;
    if (m > p.M - 1) m = p.M - 1;
    int64_t b = 0, rr = m;
    if (p.rows_per_batch < p.M) { const unsigned bb = (unsigned)m / (unsigned)p.rows_per_batch; b = bb; rr = m - (int64_t)bb * p.rows_per_batch; }
    a_ptr[i] = A + b * p.a_batch_stride + rr * p.lda + chunk * 8;
    a_lds[i] = r * 128 + ((chunk ^ (r & 7)) << 4);
    w_ptr[i] = W + (n0 + r) * p.K + chunk * 8;
    // LDS row permutation inside each 64-row half: n_local = q*16 + ni*4 + j  ->  rho = ni*16 + q*4 + j
    const int nl = r & 63;
    const int rho = (r & 64) + (((nl >> 2) & 3) << 4) + ((nl >> 4) << 2) + (nl & 3);
    w_lds[i] = 16384 + rho * 128 + ((chunk ^ (rho & 7)) << 4);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rw[4];
  const int KT = (int)(p.K / BK);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *(const u32x4*)(a_ptr[i]);
    rw[i] = *(const u32x4*)(w_ptr[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *(u32x4*)(smem + a_lds[i]) = ra[i];
    *(u32x4*)(smem + w_lds[i]) = rw[i];
  }
  __syncthreads();

  // fragment read addresses (within a stage)
  const int fr = lane & 15, fq = lane >> 4;
  int a_rd[4], w_rd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ar = wm * 64 + i * 16 + fr;
    a_rd[i] = ar * 128;  // chunk applied per k-substep
    const int wr = wn * 64 + i * 16 + fr;
    w_rd[i] = 16384 + wr * 128;
  }
  const int sw = fr & 7;  // (row & 7) == (fr & 7) for every tile row used above

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = (kt & 1) * 32768;
    const bool more = kt + 1 < KT;
    if (more) {
      const int koff = (kt + 1) * BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *(const u32x4*)(a_ptr[i] + koff);
        rw[i] = *(const u32x4*)(w_ptr[i] + koff);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int coff = ((s * 4 + fq) ^ sw) << 4;
      vec8 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *(const vec8*)(smem + cur + a_rd[i] + coff);
        fw[i] = *(const vec8*)(smem + cur + w_rd[i] + coff);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Ops::mfma16(fw[ni], fa[mi], acc[mi][ni]);
    }
    if (more) {
      const int nxt = ((kt + 1) & 1) * 32768;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(u32x4*)(smem + nxt + a_lds[i]) = ra[i];
        *(u32x4*)(smem + nxt + w_lds[i]) = rw[i];
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane (fq, fr) holds, for each mi, n = n0 + wn*64 + fq*16 + [0,16) of row m ----
  const int64_t nb = n0 + wn * 64 + fq * 16;
  float bias[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bias[j] = p.bias ? p.bias[nb + j] : 0.0f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int64_t m = m0 + wm * 64 + mi * 16 + fr;
    if (m >= p.M) continue;
    float v[16];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[ni * 4 + j] = acc[mi][ni][j] + bias[ni * 4 + j];
    gemm_store_row<T, EPI>(p, m, nb, v);
  }
}

template <typename T, int EPI>
static void launch_one(const GemmParams& p, hipStream_t stream) {
  const unsigned nwg = (unsigned)((p.N / BN) * ((p.M + BM - 1) / BM));
  ensure_dynamic_lds((const void*)gemm_kernel<T, EPI>, 65536);
  hipLaunchKernelGGL((gemm_kernel<T, EPI>), dim3(nwg), dim3(GEMM_THREADS), 65536, stream, p);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
void launch_gemm(const GemmParams& p, int epilogue, hipStream_t stream) {
  if (p.M <= 0) return;
  // big problems go to the 256x256 direct-to-LDS kernel (gemm256.hip); small N or small M stay here
  // ... and so do problems with fewer 256x256 tiles than half the CUs (one or two 30 s windows): four times as many
  // 128x128 tiles fill the chip better (encoder of one window: 10.2 -> 7.5 ms)
  static const int min_tiles = getenv("OHW_GEMM256_MIN_TILES") ? atoi(getenv("OHW_GEMM256_MIN_TILES")) : 128;
  const int64_t tiles256 = ((p.M + 255) / 256) * (p.N / 256);
  if (p.N % 256 == 0 && p.M >= 1024 && p.K % 64 == 0 && tiles256 >= min_tiles) { launch_gemm256<T>(p, epilogue, stream); return; }
  if (p.N % BN != 0 || p.K % BK != 0 || p.lda % 8 != 0 || p.a_batch_stride % 8 != 0 || p.rows_per_batch <= 0 || p.M >= ((int64_t)1 << 31) || p.N >= ((int64_t)1 << 31))
    throw Error(OHW_E_INVALID_ARG, "gemm: N must be a multiple of 128, K of 64, row strides of 8 elements");
  switch (epilogue) {
    case EPI_BIAS_T: launch_one<T, EPI_BIAS_T>(p, stream); break;
    case EPI_BIAS_GELU_T: launch_one<T, EPI_BIAS_GELU_T>(p, stream); break;
    case EPI_BIAS_RESID_F32: launch_one<T, EPI_BIAS_RESID_F32>(p, stream); break;
    case EPI_GELU_POS_F32: launch_one<T, EPI_GELU_POS_F32>(p, stream); break;
    case EPI_F32: launch_one<T, EPI_F32>(p, stream); break;
    case EPI_CROSSKV_T: launch_one<T, EPI_CROSSKV_T>(p, stream); break;
    default: throw Error(OHW_E_INVALID_ARG, "gemm: unknown epilogue");
  }
}

template void launch_gemm<bf16_t>(const GemmParams&, int, hipStream_t);
template void launch_gemm<f16_t>(const GemmParams&, int, hipStream_t);

}  // namespace ohw
