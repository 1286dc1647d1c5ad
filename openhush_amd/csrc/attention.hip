// attention.hip — fused encoder self-attention: softmax(Q K^T / 8) V without materialising the
// 1500 x 1500 scores (5.76 GB if materialised at B = 32).
//
// One workgroup = 4 waves = 128 query rows of one (window, head); each wave owns 32 query rows.
// Per 64-key block (K and V tiles staged through registers into double-buffered, swizzled LDS):
//   S^T = K Q^T   with v_mfma_f32_32x32x16 (operands swapped so a lane holds 16 keys of ONE query:
//                 the row max / row sum need no cross-lane traffic except one lane^32 exchange),
//   online softmax in registers (exp2 domain),
//   O^T += V^T P^T  where the S^T accumulator, converted to 16-bit, is already the B operand
//                 (cdna_hip_programming.md section 3 "An accumulator tile as the next MFMA's operand")
//                 and V^T fragments come from row-major V via ds_read_b64_tr_b16.
// d_head = 64 makes this softmax(VALU)-heavy: 256 flop per exp; the bound is MFMA/VALU co-issue, not HBM.
#include <type_traits>

#include "attention.hpp"

namespace ohw {

constexpr int ATT_THREADS = 256;
constexpr int ATT_QROWS = 128;  // per workgroup
constexpr int ATT_KB = 64;      // keys per block

__device__ __forceinline__ int k_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int v_swz(int row) { return ((row >> 1) & 1) << 2; }

template <typename T>
__global__ __launch_bounds__(ATT_THREADS, 2) void encoder_attention_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                          int batch, int t_len, int n_head) {
  using Ops = TypeOps<T>;
  using vec8 = typename Ops::vec8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];  // stage: K 8 KiB | V 8 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = n_head * 64;
  const int64_t ld = 3 * (int64_t)d;
  const int nqb = (t_len + ATT_QROWS - 1) / ATT_QROWS;
  const unsigned nwg = (unsigned)(nqb * n_head * batch);
  const unsigned lid = xcd_remap(blockIdx.x, nwg);
  const int qb = lid % nqb;
  const int bh = lid / nqb;
  const int h = bh % n_head, b = bh / n_head;

  const T* base = qkv + (int64_t)b * t_len * ld;
  const int ql = lane & 31, hh = lane >> 5;

  // Q fragments: B operand of S^T = K Q^T: lane holds Q[q][16*ks + 8*hh + 0..7]
  int q_row = qb * ATT_QROWS + wave * 32 + ql;
  const bool q_valid = q_row < t_len;
  if (!q_valid) q_row = t_len - 1;
  vec8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const vec8*)(base + (int64_t)q_row * ld + h * 64 + ks * 16 + hh * 8);

  // staging: thread -> keys (tid>>3) and (tid>>3)+32, chunk tid&7, for K and V
  const int skey = tid >> 3, schunk = tid & 7;
  const T* kg = base + d + h * 64 + schunk * 8;
  const T* vg = base + 2 * d + h * 64 + schunk * 8;
  int k_lds[2], v_lds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = skey + 32 * i;
    k_lds[i] = r * 128 + ((schunk ^ k_swz(r)) << 4);
    v_lds[i] = 8192 + r * 128 + ((schunk ^ v_swz(r)) << 4);
  }
  const int nkb = (t_len + ATT_KB - 1) / ATT_KB;
  // K/V of key block j travel through register set j & 1: loaded TWO blocks ahead (at the start of block j - 2), written to
  // the LDS stage j & 1 at the end of block j - 1 - a whole block time after the loads were issued, so the wait in front of
  // the LDS writes finds them landed (one block ahead, the wave stood on that wait for the rest of the memory latency)
  u32x4 rk[2][2], rv[2][2];
  auto gload = [&](auto set, int kb) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = kb * ATT_KB + skey + 32 * i;
      if (key > t_len - 1) key = t_len - 1;
      rk[S][i] = *(const u32x4*)(kg + (int64_t)key * ld);
      rv[S][i] = *(const u32x4*)(vg + (int64_t)key * ld);
    }
  };
  auto lstore = [&](auto set, int stage) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *(u32x4*)(smem + stage * 16384 + k_lds[i]) = rk[S][i];
      *(u32x4*)(smem + stage * 16384 + v_lds[i]) = rv[S][i];
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  gload(Set0{}, 0);
  lstore(Set0{}, 0);
  __syncthreads();

  // K fragment read offsets: lane reads K[key = kt*32 + ql][chunk = 2*ks + hh]
  int k_rd[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) k_rd[kt] = (kt * 32 + ql) * 128;
  const int ksw = k_swz(ql);  // k_swz(kt*32 + ql) == k_swz(ql)
  // V tr-read: lane i=lane&15 -> q_ = i>>2, p = i&3; row = base + 8*jj + 4*hh + q_, col = dt*32 + 16*((lane>>4)&1) + 4p
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = 0.125f * 1.44269504088896340736f;  // d_head^-0.5 * log2(e)

  if (nkb > 1) gload(Set1{}, 1);
  auto block = [&](auto set, int kb) {
    constexpr int S = decltype(set)::value;       // == kb & 1
    const int cur = S * 16384;
    const bool more = kb + 1 < nkb;
    if (kb + 2 < nkb) gload(std::integral_constant<int, S>{}, kb + 2);      // set S held block kb: it is in LDS since the end of block kb - 1

    f32x16 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        vec8 kf = *(const vec8*)(smem + cur + k_rd[kt] + (((2 * ks + hh) ^ ksw) << 4));
        sacc[kt] = Ops::mfma32(kf, qf[ks], sacc[kt]);
      }
    }
    // block max on the raw scores (the scale is positive); only the last key block needs the tail mask
    const int key0 = kb * ATT_KB;
    if (key0 + ATT_KB > t_len) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          sacc[kt][r] = key < t_len ? sacc[kt][r] : -INFINITY;
        }
    }
    float bmax = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) bmax = fmaxf(bmax, sacc[kt][r]);
    bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
    const float m_new = fmaxf(m_run, bmax);          // raw-score units
    // rescale only when some row's running max actually grew (rare after the first blocks); the branch is
    // wave-uniform, and alpha == 1 exactly on the skipped path
    if (!__all(m_new == m_run)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      m_run = m_new;
    }
    const float m_sc = m_run * sc;
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][r], sc, -m_sc));   // exp2(-inf) = 0 for masked keys
        sacc[kt][r] = pv;
        psum += pv;
      }
    l_run += psum;

    // O^T += V^T P^T : 4 k-steps of 16 keys, 2 dh tiles
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      const int kt = sp >> 1, s = sp & 1;
      union { vec8 v; unsigned u[4]; } pf;
#pragma unroll
      for (int j = 0; j < 4; ++j) pf.u[j] = pack2<T>(sacc[kt][8 * s + 2 * j], sacc[kt][8 * s + 2 * j + 1]);
      const int kbase = kt * 32 + 16 * s + 4 * hh + tq;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        union { vec8 v; s16x4 h4[2]; } vf;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int row = kbase + 8 * jj;
          const int colb = (dt * 32 + 16 * tg + 4 * tp) * 2;  // byte column
          const int off = cur + 8192 + row * 128 + ((((colb >> 4) ^ v_swz(row)) << 4) | (colb & 15));
          vf.h4[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off));
        }
        oacc[dt] = Ops::mfma32(vf.v, pf.v, oacc[dt]);
      }
    }
    if (more) lstore(std::integral_constant<int, 1 - S>{}, 1 - S);
    __syncthreads();
  };
  for (int kb = 0; kb < nkb; kb += 2) {
    block(Set0{}, kb);
    if (kb + 1 < nkb) block(Set1{}, kb + 1);
  }

  // finalise: l over both lane halves, write O[q][h*64 + dt*32 + 8g + 4hh + 0..3]
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
  if (q_valid) {
    T* o = out + ((int64_t)b * t_len + q_row) * d + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w.x = pack2<T>(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv);
        w.y = pack2<T>(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
        *(u32x2*)(o + dt * 32 + 8 * g + 4 * hh) = w;
      }
  }
}

template <typename T>
void launch_encoder_attention(const void* qkv, void* out, int batch, int t_len, int n_head, hipStream_t stream) {
  const int nqb = (t_len + ATT_QROWS - 1) / ATT_QROWS;
  const unsigned nwg = (unsigned)(nqb * n_head * batch);
  hipLaunchKernelGGL((encoder_attention_kernel<T>), dim3(nwg), dim3(ATT_THREADS), 0, stream, (const T*)qkv, (T*)out, batch, t_len, n_head);
  HIP_CHECK(hipGetLastError());
}
template void launch_encoder_attention<bf16_t>(const void*, void*, int, int, int, hipStream_t);
template void launch_encoder_attention<f16_t>(const void*, void*, int, int, int, hipStream_t);

}  // namespace ohw
