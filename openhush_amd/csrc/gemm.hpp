// gemm.hpp — dense MFMA GEMM  C[M][N] = A[M][K] * W[N][K]^T  (+ fused epilogue), gfx950.
#pragma once
#include "common.hpp"

namespace ohw {

enum GemmEpilogue {
  EPI_BIAS_T = 0,        // out T [M][ldc] = acc + bias
  EPI_BIAS_GELU_T = 1,   // out T = gelu(acc + bias)           (conv1, mlp.0); rows remapped per batch
  EPI_BIAS_RESID_F32 = 2,// out f32 [M][ldc] += acc + bias     (attn.out, mlp.2)
  EPI_GELU_POS_F32 = 3,  // out f32 = gelu(acc + bias) + pos[m % rows_per_batch][n]   (conv2)
  EPI_F32 = 4,           // out f32 = acc + bias               (debug / logits)
  EPI_CROSSKV_T = 5      // out T head-major: n -> (slab = n / d, h, dh); slab-major [slab][B][H][T][64]
};

struct GemmParams {
  const void* A;      // T, row m at A + (m / rows_per_batch) * a_batch_stride + (m % rows_per_batch) * lda
  const void* W;      // T [N][K] row-major
  const float* bias;  // [N] or nullptr
  void* out;
  const float* pos;   // EPI_GELU_POS_F32: [rows_per_batch][N]
  int64_t M, N, K;
  int64_t lda, a_batch_stride, rows_per_batch;
  int64_t ldc, c_batch_stride;  // output row m at out + (m / rows_per_batch) * c_batch_stride + (m % rows_per_batch) * ldc
  // EPI_CROSSKV_T
  int32_t d_model, n_head, t_len, batch;   // batch = windows of the decode batch the K/V belong to
  int32_t batch_offset;                    // first window of this GEMM's rows inside that decode batch (ohw_encode_slice)
  int32_t group_m;    // gemm256: m-tiles per L2-locality group (set by the launcher)
};

// N % 128 == 0, K % 64 == 0, lda/ldc/strides multiples of 8 elements (16-byte rows)
template <typename T> void launch_gemm(const GemmParams& p, int epilogue, hipStream_t stream);
// the 256x256x64 variant (N % 256 == 0); launch_gemm dispatches to it for large problems
template <typename T> void launch_gemm256(const GemmParams& p, int epilogue, hipStream_t stream);

}  // namespace ohw
