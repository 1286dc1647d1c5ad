// gemm256.hip — 256x256x64 MFMA GEMM with direct-to-LDS loads kept in flight across barriers (gfx950).
//
// Same contract and epilogues as gemm.hip (C = A * W^T, swapped MFMA operands, 16 contiguous output
// columns per lane), for the large encoder / cross-K/V GEMMs (N % 256 == 0).
//
// Structure (cdna_hip_programming.md section 5, "Pipelining across barriers"):
//   * 512 threads = 8 waves (2 along m x 4 along n), each wave a 128x64 output tile = 8x4 MFMA tiles of
//     16x16x32 (128 accumulator VGPRs); one workgroup per CU.
//   * LDS: 2 stages x (256x64 A + 256x64 W) x 2 B = 128 KiB, ONE shared array.  Rows are 128 B; the
//     16-byte chunks of a row are XOR-swizzled by (row & 7).  Tiles are filled by global_load_lds_dwordx4
//     (1 KiB = 8 rows per wave-instruction): the LDS destination is lane-linear, so the swizzle is applied
//     to the per-lane SOURCE address and to the fragment read (rule 21 of the guide).
//   * the loads of tile t+1 stay in flight while tile t is computed: counted s_waitcnt vmcnt(8) + raw
//     s_barrier (a __syncthreads() would drain vmcnt to 0), two barriers per K-tile:
//         wait(tile t landed) ; barrier ; 64 MFMA per wave ; barrier ; issue loads of tile t+2.
// Roofline: MFMA (128 flop per LDS-read byte at this wave tile; HBM traffic per flop 2x lower than the
// 128^2 kernel).
#include "gemm.hpp"
#include "gemm_epilogue.hpp"

namespace ohw {

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64;
constexpr int G2_THREADS = 512;
constexpr int G2_STAGE = 65536;   // bytes per stage: A 32 KiB | W 32 KiB

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

  // ---- staging: wave w fills 1-KiB blocks 4w .. 4w+3 of each operand (8 rows x 128 B per block).
  // lane -> row r = lane>>3 of the block, LDS chunk c = lane&7, which must hold data chunk c ^ r.
  const int sr = lane >> 3, sc = lane & 7;
  const int src_chunk = sc ^ sr;
  const T* a_src[4];
  const T* w_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int R = (wave * 4 + j) * 8 + sr;          // LDS row 0..255
    int64_t m = m0 + R;
    if (m > p.M - 1) m = p.M - 1;
    const int64_t b = m / p.rows_per_batch, rr = m % p.rows_per_batch;
    a_src[j] = A + b * p.a_batch_stride + rr * p.lda + src_chunk * 8;
    // LDS row rho (inside each 64-row block: rho = ni*16 + q*4 + jj) holds W row q*16 + ni*4 + jj
    const int rl = R & 63;
    const int nl = (((rl >> 2) & 3) << 4) + ((rl >> 4) << 2) + (rl & 3);
    w_src[j] = W + (n0 + (R & ~63) + nl) * p.K + src_chunk * 8;
  }
  const int KT = (int)(p.K / G2_BK);

  auto issue = [&](int kt, int stage) {
    const int koff = kt * G2_BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[j] + koff),
                                       (__attribute__((address_space(3))) void*)(smem + stage * G2_STAGE + (wave * 4 + j) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[j] + koff),
                                       (__attribute__((address_space(3))) void*)(smem + stage * G2_STAGE + 32768 + (wave * 4 + j) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = tile row + (lane & 15); (row & 7) == (lane & 7)
  const int fr = lane & 15, fq = lane >> 4, sw = lane & 7;
  const int a_rd = (wm * 128 + fr) * 128;
  const int w_rd = 32768 + (wn * 64 + fr) * 128;

  issue(0, 0);
  if (KT > 1) issue(1, 1);

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = (kt & 1) * G2_STAGE;
    if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // 4 clusters of 16 MFMAs (k-substep s = c >> 1, m-half = c & 1).  The LDS fragment reads of cluster
    // c + 1 are issued before the MFMAs of cluster c and interleaved with them, so their latency is hidden.
    const int coff0 = ((0 * 4 + fq) ^ sw) << 4, coff1 = ((1 * 4 + fq) ^ sw) << 4;
    vec8 fw0[4], fw1[4], fa0[4], fa1[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fw0[ni] = *(const vec8*)(smem + cur + w_rd + ni * 2048 + coff0);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa0[i] = *(const vec8*)(smem + cur + a_rd + i * 2048 + coff0);
    // cluster 0 (s=0, half 0) | prefetch cluster 1 (s=0, half 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) fa1[i] = *(const vec8*)(smem + cur + a_rd + (4 + i) * 2048 + coff0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[i][ni] = Ops::mfma16(fw0[ni], fa0[i], acc[i][ni]);
    __builtin_amdgcn_s_setprio(0);
    // cluster 1 (s=0, half 1) | prefetch cluster 2 (s=1, half 0) + W fragments of s=1
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fw1[ni] = *(const vec8*)(smem + cur + w_rd + ni * 2048 + coff1);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa0[i] = *(const vec8*)(smem + cur + a_rd + i * 2048 + coff1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[4 + i][ni] = Ops::mfma16(fw0[ni], fa1[i], acc[4 + i][ni]);
    __builtin_amdgcn_s_setprio(0);
    // cluster 2 (s=1, half 0) | prefetch cluster 3 (s=1, half 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) fa1[i] = *(const vec8*)(smem + cur + a_rd + (4 + i) * 2048 + coff1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[i][ni] = Ops::mfma16(fw1[ni], fa0[i], acc[i][ni]);
    __builtin_amdgcn_s_setprio(0);
    // cluster 3 (s=1, half 1)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[4 + i][ni] = Ops::mfma16(fw1[ni], fa1[i], acc[4 + i][ni]);
    __builtin_amdgcn_s_setprio(0);
    if (kt + 2 < KT) {
      // every wave has finished reading this stage before anyone refills it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(kt + 2, kt & 1);
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
