// gemm_epilogue.hpp — fused epilogues shared by the MFMA GEMM kernels: a lane stores 16 contiguous
// output columns nb .. nb+15 of ONE row m (bias already added into v[]).
#pragma once
#include "gemm.hpp"

namespace ohw {

template <typename T, int EPI>
__device__ __forceinline__ void gemm_store_row(const GemmParams& p, int64_t m, int64_t nb, float (&v)[16]) {
  // 64-bit division is ~100 instructions on the GPU and this runs once per output row per lane
  int64_t b = 0, rr = m;
  if (p.rows_per_batch < p.M) { const unsigned bb = (unsigned)m / (unsigned)p.rows_per_batch; b = bb; rr = m - (int64_t)bb * p.rows_per_batch; }
  if constexpr (EPI == EPI_BIAS_T || EPI == EPI_BIAS_GELU_T) {
    if constexpr (EPI == EPI_BIAS_GELU_T) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = gelu_erf(v[j]);
    }
    T* o = (T*)p.out + b * p.c_batch_stride + rr * p.ldc + nb;
    u32x4 lo, hi;
    lo.x = pack2<T>(v[0], v[1]); lo.y = pack2<T>(v[2], v[3]); lo.z = pack2<T>(v[4], v[5]); lo.w = pack2<T>(v[6], v[7]);
    hi.x = pack2<T>(v[8], v[9]); hi.y = pack2<T>(v[10], v[11]); hi.z = pack2<T>(v[12], v[13]); hi.w = pack2<T>(v[14], v[15]);
    *(u32x4*)o = lo;
    *(u32x4*)(o + 8) = hi;
  } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
    float* o = (float*)p.out + b * p.c_batch_stride + rr * p.ldc + nb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 old = *(const f32x4*)(o + 4 * j);
      old.x += v[4 * j]; old.y += v[4 * j + 1]; old.z += v[4 * j + 2]; old.w += v[4 * j + 3];
      *(f32x4*)(o + 4 * j) = old;
    }
  } else if constexpr (EPI == EPI_GELU_POS_F32) {
    float* o = (float*)p.out + b * p.c_batch_stride + rr * p.ldc + nb;
    const float* ps = p.pos + rr * p.N + nb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 pp = *(const f32x4*)(ps + 4 * j);
      f32x4 r;
      r.x = gelu_erf(v[4 * j]) + pp.x; r.y = gelu_erf(v[4 * j + 1]) + pp.y;
      r.z = gelu_erf(v[4 * j + 2]) + pp.z; r.w = gelu_erf(v[4 * j + 3]) + pp.w;
      *(f32x4*)(o + 4 * j) = r;
    }
  } else if constexpr (EPI == EPI_F32) {
    float* o = (float*)p.out + b * p.c_batch_stride + rr * p.ldc + nb;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(o + 4 * j) = (f32x4){v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
  } else if constexpr (EPI == EPI_CROSSKV_T) {
    // n -> slab (layer*2 + k/v), head, dh ; out[slab][b][h][t][64]
    const int64_t slab = nb / p.d_model, rem = nb % p.d_model;
    const int64_t h = rem >> 6, dh = rem & 63;
    T* o = (T*)p.out + ((((slab * p.batch + b + p.batch_offset) * p.n_head + h) * p.t_len + rr) << 6) + dh;
    u32x4 lo, hi;
    lo.x = pack2<T>(v[0], v[1]); lo.y = pack2<T>(v[2], v[3]); lo.z = pack2<T>(v[4], v[5]); lo.w = pack2<T>(v[6], v[7]);
    hi.x = pack2<T>(v[8], v[9]); hi.y = pack2<T>(v[10], v[11]); hi.z = pack2<T>(v[12], v[13]); hi.w = pack2<T>(v[14], v[15]);
    *(u32x4*)o = lo;
    *(u32x4*)(o + 8) = hi;
  }
}

}  // namespace ohw
