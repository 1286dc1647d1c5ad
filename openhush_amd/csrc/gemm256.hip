// gemm256.hip — 256x256x64 MFMA GEMM with direct-to-LDS loads kept in flight across barriers (gfx950).
//
// Same contract and epilogues as gemm.hip (C = A * W^T, swapped MFMA operands, 16 contiguous output
// columns per lane), for the large encoder / cross-K/V GEMMs (N % 256 == 0).
//
// Structure (after cdna_hip_programming.md section 5, "Pipelining across barriers" / the 8-phase idea):
//   * 512 threads = 8 waves (2 along m x 4 along n), each wave a 128x64 output tile = 8x4 MFMA tiles of
//     16x16x32 (128 accumulator VGPRs); one workgroup per CU.
//   * LDS: ONE 128-KiB array = 2 stages x 2 k-halves x (256x32 A + 256x32 W) x 2 B.  A k-half buffer has
//     64-byte rows; its four 16-byte chunks are XOR-swizzled with key[(row >> 2) & 3], key = {0,3,2,1},
//     which makes every ds_read_b128 fragment read conflict-free (bank analysis in DESIGN.md section 5).
//     Buffers are filled by global_load_lds_dwordx4 (1 KiB = 16 rows per wave-instruction): the LDS
//     destination is lane-linear, so the swizzle goes on the per-lane SOURCE address and on the read.
//   * a K-tile is computed in two halves (k 0..31, then k 32..63), 32 MFMAs per wave each.  The refill of
//     a k-half buffer is issued INSIDE the compute phase that follows the barrier which retired its last
//     readers, interleaved with the MFMAs, and stays in flight for ~1.5 K-tiles:
//         start(t): vmcnt(8); barrier; [MFMA k-half 0 of tile t  ||  issue tile t+1 / k-half 1]
//         mid(t)  : vmcnt(8); barrier; [MFMA k-half 1 of tile t  ||  issue tile t+2 / k-half 0]
//     Counted s_waitcnt vmcnt + raw s_barrier only (a __syncthreads() would drain vmcnt to 0).
// Roofline: MFMA.
#include "gemm.hpp"
#include "gemm_epilogue.hpp"

namespace ohw {

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64;
constexpr int G2_THREADS = 512;
constexpr int G2_HALF = 32768;    // bytes per (stage, k-half): A 16 KiB | W 16 KiB
constexpr int G2_STAGE = 2 * G2_HALF;

__device__ __forceinline__ int g2_key(int row) { return (0x6C >> (((row >> 2) & 3) * 2)) & 3; }  // {0,3,2,1}

template <typename T, int EPI>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm256_kernel(GemmParams p) {
  using Ops = TypeOps<T>;
  using vec8 = typename Ops::vec8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const unsigned n_tiles_n = (unsigned)(p.N / G2_BN);
  const unsigned n_tiles_m = (unsigned)((p.M + G2_BM - 1) / G2_BM);
  const unsigned nwg = n_tiles_n * n_tiles_m;
  const unsigned lid = xcd_remap(blockIdx.x, nwg);
  const int64_t m0 = (int64_t)(lid / n_tiles_n) * G2_BM;
  const int64_t n0 = (int64_t)(lid % n_tiles_n) * G2_BN;

  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;

  // ---- staging: per k-half buffer each operand is 16 blocks of 1 KiB (16 rows x 64 B); wave w fills
  // blocks 2w and 2w+1.  lane -> row r = lane>>2 of the block, LDS chunk c = lane&3, holding data chunk
  // c ^ key(row).
  const int sr = lane >> 2, sc = lane & 3;
  const T* a_src[2];
  const T* w_src[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = (wave * 2 + j) * 16 + sr;         // LDS row 0..255
    const int dchunk = sc ^ g2_key(R);
    int64_t m = m0 + R;
    if (m > p.M - 1) m = p.M - 1;
    const int64_t b = m / p.rows_per_batch, rr = m % p.rows_per_batch;
    a_src[j] = A + b * p.a_batch_stride + rr * p.lda + dchunk * 8;
    // LDS row rho (inside each 64-row block: rho = ni*16 + q*4 + jj) holds W row q*16 + ni*4 + jj
    const int rl = R & 63;
    const int nl = (((rl >> 2) & 3) << 4) + ((rl >> 4) << 2) + (rl & 3);
    w_src[j] = W + (n0 + (R & ~63) + nl) * p.K + dchunk * 8;
  }
  const int KT = (int)(p.K / G2_BK);

  // one glds of tile kt / k-half h: q = 0,1 -> A blocks, q = 2,3 -> W blocks
  auto issue1 = [&](int kt, int h, int q) {
    const int koff = kt * G2_BK + h * 32;
    const int buf = (kt & 1) * G2_STAGE + h * G2_HALF;
    if (q < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + koff),
                                       (__attribute__((address_space(3))) void*)(smem + buf + (wave * 2 + q) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[q - 2] + koff),
                                       (__attribute__((address_space(3))) void*)(smem + buf + 16384 + (wave * 2 + (q - 2)) * 1024), 16, 0, 0);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside a k-half buffer: row = tile row + (lane & 15), chunk (lane >> 4) ^ key(row);
  // tile rows start at multiples of 16, so key(row) depends on (lane & 15) only
  const int fr = lane & 15, fq = lane >> 4;
  const int fch = (fq ^ g2_key(fr)) << 4;
  const int a_rd = (wm * 128 + fr) * 64 + fch;
  const int w_rd = 16384 + (wn * 64 + fr) * 64 + fch;

  // prologue: tile 0 both halves, tile 1 k-half 0
#pragma unroll
  for (int q = 0; q < 4; ++q) issue1(0, 0, q);
#pragma unroll
  for (int q = 0; q < 4; ++q) issue1(0, 1, q);
  if (KT > 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue1(1, 0, q);
  }

  for (int kt = 0; kt < KT; ++kt) {
    const bool more1 = kt + 1 < KT, more2 = kt + 2 < KT;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // groups of 4 glds issued after the buffer about to be read:
      //   h = 0: (kt).h1 and, if it exists, (kt+1).h0        h = 1: (kt+1).h0 and (kt+1).h1, if they exist
      if (h == 0) {
        if (more1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        if (more1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // also retires this wave's LDS reads of the previous phase before anyone refills that buffer
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int cur = (kt & 1) * G2_STAGE + h * G2_HALF;
      // what this phase refills: h = 0 -> tile kt+1 / k-half 1 ; h = 1 -> tile kt+2 / k-half 0
      const bool fill = h == 0 ? more1 : more2;
      const int fkt = h == 0 ? kt + 1 : kt + 2, fh = h == 0 ? 1 : 0;
      vec8 fw[4], fa0[4], fa1[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fw[ni] = *(const vec8*)(smem + cur + w_rd + ni * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa0[i] = *(const vec8*)(smem + cur + a_rd + i * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa1[i] = *(const vec8*)(smem + cur + a_rd + (4 + i) * 1024);
      // the refill is issued in two halves in front of the two MFMA clusters; while one wave of a SIMD
      // issues its DMA the other keeps the MFMA pipe busy.  (Measured alternatives, same shapes: all four
      // glds between the clusters, or the second A-fragment group pinned behind the first MFMAs with
      // sched_barrier: both 4 % slower.)
      if (fill) { issue1(fkt, fh, 0); issue1(fkt, fh, 1); }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[i][ni] = Ops::mfma16(fw[ni], fa0[i], acc[i][ni]);
      __builtin_amdgcn_s_setprio(0);
      if (fill) { issue1(fkt, fh, 2); issue1(fkt, fh, 3); }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[4 + i][ni] = Ops::mfma16(fw[ni], fa1[i], acc[4 + i][ni]);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // ---- epilogue: lane (fq, fr) holds, for each mi, columns n0 + wn*64 + fq*16 + [0,16) of row m ----
  const int64_t nb = n0 + wn * 64 + fq * 16;
  float bias[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bias[j] = p.bias ? p.bias[nb + j] : 0.0f;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int64_t m = m0 + wm * 128 + mi * 16 + fr;
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
static void launch256_one(const GemmParams& p, hipStream_t stream) {
  const unsigned nwg = (unsigned)((p.N / G2_BN) * ((p.M + G2_BM - 1) / G2_BM));
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute((const void*)gemm256_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G2_STAGE));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<T, EPI>), dim3(nwg), dim3(G2_THREADS), 2 * G2_STAGE, stream, p);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
void launch_gemm256(const GemmParams& p, int epilogue, hipStream_t stream) {
  if (p.M <= 0) return;
  if (p.N % G2_BN != 0 || p.K % G2_BK != 0 || p.lda % 8 != 0 || p.a_batch_stride % 8 != 0 || p.rows_per_batch <= 0)
    throw Error(OHW_E_INVALID_ARG, "gemm256: N must be a multiple of 256, K of 64, row strides of 8 elements");
  switch (epilogue) {
    case EPI_BIAS_T: launch256_one<T, EPI_BIAS_T>(p, stream); break;
    case EPI_BIAS_GELU_T: launch256_one<T, EPI_BIAS_GELU_T>(p, stream); break;
    case EPI_BIAS_RESID_F32: launch256_one<T, EPI_BIAS_RESID_F32>(p, stream); break;
    case EPI_GELU_POS_F32: launch256_one<T, EPI_GELU_POS_F32>(p, stream); break;
    case EPI_F32: launch256_one<T, EPI_F32>(p, stream); break;
    case EPI_CROSSKV_T: launch256_one<T, EPI_CROSSKV_T>(p, stream); break;
    default: throw Error(OHW_E_INVALID_ARG, "gemm256: unknown epilogue");
  }
}

template void launch_gemm256<bf16_t>(const GemmParams&, int, hipStream_t);
template void launch_gemm256<f16_t>(const GemmParams&, int, hipStream_t);

}  // namespace ohw
