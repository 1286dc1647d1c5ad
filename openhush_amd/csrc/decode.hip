// decode.hip — kernels of one autoregressive decoder step, gfx950.
//
// The step is HBM-bound: every step streams all decoder weights once (1.81 GB bf16 for large-v3,
// independent of the batch) plus, per window, the cross-attention K/V of every layer (245.8 MB).
// Layouts are chosen so that every wave-instruction reads 1 KiB of contiguous HBM:
//   * decoder linears are stored as MFMA fragments [N/16][K/32][64 lanes][8]: a wave streams its
//     16 output rows as consecutive 1-KiB blocks straight into VGPRs (no LDS round trip: the weights
//     are used once; cdna_hip_programming.md "GEMV / M <= 16 decode weights"),
//   * cross K/V are head-major [B][H][1500][64]: one (window, head) = 192 KiB contiguous per tensor.
#include <type_traits>

#include "kernels.hpp"

namespace ohw {

// ------------------------------------------------------------------------------------------------
// token + position embedding
// ------------------------------------------------------------------------------------------------
#ifdef OHW_TRACE
// in-kernel timeline for tools/dec_trace.py (instrumented build only): first and last workgroup of each launch
// append {100 MHz wall clock, kernel id, stage, which workgroup}
__device__ unsigned long long ohw_trace_buf[1 << 18];
__device__ unsigned ohw_trace_n;
__device__ __forceinline__ void trace_mark(unsigned id, unsigned stage) {
  const unsigned nb = gridDim.x * gridDim.y * gridDim.z, bi = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  __builtin_amdgcn_sched_barrier(0);
  if (threadIdx.x == 0 && (bi == 0 || bi == nb - 1 || bi == nb / 2)) {
    const unsigned k = atomicAdd(&ohw_trace_n, 1u);
    if (k < (1u << 18)) ohw_trace_buf[k] = ((unsigned long long)wall_clock64() << 16) | (id << 8) | (stage << 2) | (bi == 0 ? 0 : bi == nb - 1 ? 1 : 2);
  }
}
#define TRACE(id, stage) trace_mark(id, stage)
#else
#define TRACE(id, stage)
#endif

template <typename T>
__global__ void embed_kernel(const T* __restrict__ emb, const float* __restrict__ pos, const int32_t* __restrict__ tok,
                             const int32_t* __restrict__ n_past, float* __restrict__ x, T* __restrict__ x16, float* __restrict__ stat,
                             int M, int n_new, int d) {
  // one thread per group of 16 columns: besides the fp32 residual row it writes the 16-bit tiled copy and the group's
  // (mean, sum of squared deviations) - what the post-norm GEMMs read instead of whole fp32 rows
  const int m = blockIdx.x;
  const int b = m / n_new, i = m % n_new;
  const int t = tok[m];
  const int pp = n_past[b] + i;
  const int64_t kblocks = d / 32;
  const int n_grp = d / 16;
  for (int g = threadIdx.x; g < n_grp; g += blockDim.x) {
    float v[16];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int k = g * 16 + c;
      const int64_t off = ((((int64_t)(t >> 4) * kblocks + (k >> 5)) * 64 + (t & 15) + 16 * ((k & 31) >> 3)) << 3) + (k & 7);
      v[c] = (float)emb[off] + pos[(int64_t)pp * d + k];
      sum += v[c];
    }
    const float mean = sum * (1.0f / 16.0f);
    float m2 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int k = g * 16 + c;
      x[(int64_t)m * d + k] = v[c];
      if (x16) x16[act_tiled_offset(m, k, d)] = (T)v[c];
      m2 += (v[c] - mean) * (v[c] - mean);
    }
    if (stat) { stat[((int64_t)m * n_grp + g) * 2] = mean; stat[((int64_t)m * n_grp + g) * 2 + 1] = m2; }
  }
}
template <typename T>
void launch_embed(const void* emb, const float* pos, const int32_t* tok, const int32_t* n_past, float* x, void* x16, float* stat, int M,
                  int n_new, int d, hipStream_t s) {
  if (d % 32 != 0) throw Error(OHW_E_INVALID_ARG, "embed: d_model must be a multiple of 32");
  hipLaunchKernelGGL((embed_kernel<T>), dim3(M), dim3(128), 0, s, (const T*)emb, pos, tok, n_past, x, (T*)x16, stat, M, n_new, d);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM  y[M][N] = f(x)[M][K] * W[N][K]^T  for 32 rows per pass, W in fragment tiles.
// grid = (Npad/16, ceil(M/32)); 8 waves split K (interleaved 1-KiB blocks, every block of a wave's
// share in flight at once: the kernel is latency-bound, not issue-bound), fixed-order LDS reduction
// (bitwise reproducible, no atomics).
// LN = true fuses the pre-LayerNorm of the fp32 residual stream into the prologue: the workgroup
// normalises its 32 rows ((x - mean) * rstd; gamma and beta are folded into W and the bias at load, which
// takes two dependent L2 round trips out of every such launch) into LDS (16-bit, rows padded by 16 B
// against bank conflicts) and the MFMA B operand is read from there.  Every workgroup redoes the 32-row LN (160 KB of L2 reads) - cheaper
// than one more dependent launch in a chain of 5 us kernels.
// ------------------------------------------------------------------------------------------------
template <typename T, bool SLOTS, bool COH>
__device__ __forceinline__ void self_attn_row(const T* __restrict__ q, const T* __restrict__ kc, const T* __restrict__ vc, int n_keys, int b, int m, int h,
                                              int n_head, int n_ctx, const int32_t* __restrict__ slots, T* __restrict__ out, float* qs, float* ps,
                                              int lane, unsigned kv_bytes);

constexpr int DG_THREADS = 512;
constexpr int DG_WAVES = DG_THREADS / 64;
constexpr int DG_LN_MAXK = 1280;  // fused LayerNorm: 16 threads per row, 20 float4 each

template <typename T, int EPI, bool LN, int NT, int MT>
__global__ __launch_bounds__(DG_THREADS, 2) void dec_gemm_kernel(DecGemmParams p) {
  // MT = m-tiles (16 rows) per workgroup.  MT = 1 (RESID GEMMs): the two row halves of a weight tile go to two
  // workgroups on the same XCD (the second reads the weights from L2), each pulling half the activation bytes - per-CU
  // ingest, not HBM, bounds these launches.
  static_assert(MT == 1 || MT == 2, "MT");
  using Ops = TypeOps<T>;
  using vec8 = typename Ops::vec8;
  extern __shared__ __attribute__((aligned(16))) unsigned char dg_smem[];
  f32x4* part = (f32x4*)dg_smem;                                      // [8 waves][NT][2][64]
  unsigned char* ylds = dg_smem + DG_WAVES * NT * 2 * 64 * 16;        // LN: [32][K*2 + 16] bytes; post-norm: f32 [32][2] mean, rstd
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt0 = blockIdx.x * NT;
  const int m0 = blockIdx.y * 16 * MT;
  const int kblocks = p.K / 32;
  const int ystride = p.K * 2 + 16;
  const int n_tiles = (p.N + 15) / 16;
  TRACE(16 + EPI * 2 + (LN ? 1 : 0), 0);

  // epilogue coordinates are known up front: one output per thread and n-tile
  const int e_mt = tid >> 8, e_ll = (tid >> 2) & 63, e_reg = tid & 3;
  const int e_m = m0 + e_mt * 16 + (e_ll & 15);
  const bool e_active = MT == 2 || e_mt == 0;
  // split-K (RESID only, gridDim.z slices): this workgroup owns k-blocks [k_lo, k_hi)
  const int ksplit = gridDim.z;
  const int k_lo = (int)((int64_t)kblocks * blockIdx.z / ksplit), k_hi = (int)((int64_t)kblocks * (blockIdx.z + 1) / ksplit);
  float resid_old[NT];
  // producers of the post-norm path (stat_out != null, NT == 1): thread = (m-tile, row, column) so that the 16 columns of
  // a row sit in 16 consecutive lanes (row statistics by shuffles, 64-byte row segments per store)
  const bool stat_epi = EPI == DEPI_BIAS_RESID && NT == 1 && p.stat_out != nullptr && ksplit == 1;
  const int s_mt = tid >> 7, s_r = (tid >> 3) & 15, s_c = (tid & 7) * 2;      // two adjacent columns per thread
  const int s_m = m0 + s_mt * 16 + s_r, s_n = nt0 * 16 + s_c;
  const bool s_active = tid < 128 * MT;
  float resid_old2 = 0.f;
  if (EPI == DEPI_BIAS_RESID && ksplit == 1) {
    // issue the read of the residual now: its latency hides under the weight stream
    if (stat_epi) {
      if (s_active && s_m < p.M && s_n < p.N) {
        const float2 r2 = *(const float2*)((const float*)p.out + (int64_t)s_m * p.ld_out + s_n);
        resid_old[0] = r2.x; resid_old2 = r2.y;
      } else {
        resid_old[0] = 0.f;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = (nt0 + t) * 16 + 4 * (e_ll >> 4) + e_reg;
        resid_old[t] = (e_active && e_m < p.M && n < p.N) ? ((const float*)p.out)[(int64_t)e_m * p.ld_out + n] : 0.f;
      }
    }
  }
  // post-norm consumers: the row statistics of this workgroup's 16 * MT rows from the published per-16-column tiles.
  // 32 lanes per row; lane l merges tiles l, l + 32, l + 64 ... in order (Chan's update), then the xor tree 16 .. 1; only
  // lane 0's result is used, so it is one fixed function of the tiles whatever MT is.  The loads go out before the weights'.
  constexpr int PN_TILES = 4;                  // statistics tiles per lane: rows of up to 4 * 32 * 16 = 2048 columns
  float pn_mb[MT][PN_TILES], pn_qb[MT][PN_TILES];
  const bool pn = !LN && p.pn != 0;
  if (pn) {
    // only the loads here (clamped index, unconditional): they are needed in the epilogue, not before - merging them now
    // would put their round trip in front of the weight stream
    const int l32 = tid & 31, prow = tid >> 5;
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      int m = m0 + q * 16 + prow;
      if (m > p.M - 1) m = p.M - 1;
      const float2* st = (const float2*)p.stat_in + (int64_t)m * p.n_stat;
#pragma unroll
      for (int u = 0; u < PN_TILES; ++u) {
        const int j = l32 + 32 * u;
        const float2 t2 = st[j < p.n_stat ? j : p.n_stat - 1];
        pn_mb[q][u] = t2.x; pn_qb[q][u] = t2.y;
      }
    }
  }

  // fragment-tile pointers of this workgroup's n-tiles
  const vec8* wt[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int nt = nt0 + t;
    if (nt > n_tiles - 1) nt = n_tiles - 1;   // ragged last workgroup: recompute the last tile, never stored
    wt[t] = (const vec8*)p.w + (int64_t)nt * kblocks * 64 + lane;
  }
  // LN variants (K <= 1280: at most 5 k-blocks per wave): the whole weight share of the wave is
  // requested BEFORE the LayerNorm prologue, so the HBM round trip overlaps the prologue's own loads
  constexpr int UL = DG_LN_MAXK / 32 / DG_WAVES;   // 5
  vec8 wpre[NT][UL];
  if constexpr (LN) {
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      int kk = wave + DG_WAVES * u;
      if (kk > kblocks - 1) kk = kblocks - 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) wpre[t][u] = __builtin_nontemporal_load(&wt[t][(int64_t)kk * 64]);
    }
  }

  if constexpr (LN) {
    // all 512 threads: 16 * MT rows, TPR = 32 / MT threads per row, every load of the tile in flight at once
    const float* __restrict__ xf = (const float*)p.x;
    constexpr int TPR = 32 / MT;
    const int row = tid / TPR, sub = tid % TPR;
    int m = m0 + row;
    if (m > p.M - 1) m = p.M - 1;
    const float* xr = xf + (int64_t)m * p.K;
    constexpr int NV = DG_LN_MAXK / (4 * TPR);
    // every load is unconditional (clamped column, masked value): a per-element "load or zero" branch
    // makes hipcc wait vmcnt(0) per element and serialises 20 L2 round trips
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * TPR + sub) * 4;
      const int cc = c < p.K ? c : 0;
      v[i] = *(const f32x4*)(xr + cc);
    }
    // The row statistics must not depend on MT (which follows the CU budget of the stream: a decode on a CU-masked
    // stream has to give the tokens of the same decode on the whole chip, bit for bit).  The sum is DEFINED as that of
    // 32 "virtual" threads per row - virtual thread s adds the float4 groups s, s + 32, s + 64 ... in order - followed
    // by the xor tree 16, 8, 4, 2, 1.  With 16 threads per row (MT = 2) a thread plays the virtual threads `sub` (its
    // even groups) and `sub + 16` (its odd groups) and adds the two: the tree's first level.
    constexpr int VS = 32 / TPR;                 // virtual threads per physical thread: 1 or 2
    float sv[VS];
#pragma unroll
    for (int q = 0; q < VS; ++q) sv[q] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * TPR + sub) * 4;
      const float ok = c < p.K ? 1.f : 0.f;
      sv[i % VS] += ok * ((v[i].x + v[i].y) + (v[i].z + v[i].w));
    }
    float sum = sv[0];
    if constexpr (VS == 2) sum += sv[1];
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)p.K;
    float vv[VS];
#pragma unroll
    for (int q = 0; q < VS; ++q) vv[q] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * TPR + sub) * 4;
      const float ok = c < p.K ? 1.f : 0.f;
      const float a = v[i].x - mean, b2 = v[i].y - mean, c2 = v[i].z - mean, d2 = v[i].w - mean;
      vv[i % VS] += ok * ((a * a + b2 * b2) + (c2 * c2 + d2 * d2));
    }
    float var = vv[0];
    if constexpr (VS == 2) var += vv[1];
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = rsqrtf(var / (float)p.K + 1e-5f);
    // gamma / beta live in the weights and the bias (launch_fold_ln): only the normalisation is left here
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * TPR + sub) * 4;
      u32x2 w2;
      w2.x = pack2<T>((v[i].x - mean) * rstd, (v[i].y - mean) * rstd);
      w2.y = pack2<T>((v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
      if (c < p.K) *(u32x2*)(ylds + row * ystride + c * 2) = w2;
    }
    __syncthreads();
    TRACE(16 + EPI * 2 + 1, 1);
  }

  const T* __restrict__ x = (const T*)p.x;
  // activation tiles: k-block kk of the 16-row tile mt is the 1 KiB at ((mt * kblocks + kk) * 64 + lane) * 8 elements
  const T* x0 = LN ? nullptr : x + ((int64_t)(m0 >> 4) * kblocks * 64 + lane) * 8;
  const T* x1 = LN ? nullptr : x0 + (int64_t)kblocks * 512;
  const bool ok0 = m0 + (lane & 15) < p.M, ok1 = m0 + 16 + (lane & 15) < p.M;
  const unsigned char* y0 = ylds + (lane & 15) * ystride + (lane >> 4) * 16;
  const unsigned char* y1 = y0 + 16 * ystride;
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  auto run = [&](auto ucount, int kb) {
    constexpr int U = decltype(ucount)::value;
    vec8 w[NT][U], a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) w[t][u] = __builtin_nontemporal_load(&wt[t][(int64_t)(kb + DG_WAVES * u) * 64]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = kb + DG_WAVES * u;
      if constexpr (LN) {
        a[u] = *(const vec8*)(y0 + kk * 64);
        if constexpr (MT == 2) b[u] = *(const vec8*)(y1 + kk * 64);
      }
    }
    if constexpr (!LN) {
      // activation fragments: lanes whose row lies past M stay zero and load nothing (one exec-masked batch of loads,
      // not a branch per load) - at batch 1 a 16-row tile holds one real row
      if (ok0) {
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = *(const vec8*)(x0 + (int64_t)(kb + DG_WAVES * u) * 512);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) a[u][e] = 0;
      }
      if constexpr (MT == 2) {
        if (ok1) {
#pragma unroll
          for (int u = 0; u < U; ++u) b[u] = *(const vec8*)(x1 + (int64_t)(kb + DG_WAVES * u) * 512);
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) b[u][e] = 0;
        }
      }
    }
    // keep every load above issued before the first MFMA (the scheduler otherwise sinks loads next to
    // their use to save registers and the wave pays one HBM round trip per k-block)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t][0] = Ops::mfma16(w[t][u], a[u], acc[t][0]);
        if constexpr (MT == 2) acc[t][1] = Ops::mfma16(w[t][u], b[u], acc[t][1]);
      }
  };
  if constexpr (LN) {
    vec8 a[UL], b[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int kk = wave + DG_WAVES * u;
      const int kc = kk < kblocks ? kk : 0;
      a[u] = *(const vec8*)(y0 + kc * 64);
      if constexpr (MT == 2) b[u] = *(const vec8*)(y1 + kc * 64);
      if (kk >= kblocks) {   // register select, not a load branch: blocks past K contribute zero
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[u][e] = 0; if constexpr (MT == 2) b[u][e] = 0; }
      }
    }
#pragma unroll
    for (int u = 0; u < UL; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t][0] = Ops::mfma16(wpre[t][u], a[u], acc[t][0]);
        if constexpr (MT == 2) acc[t][1] = Ops::mfma16(wpre[t][u], b[u], acc[t][1]);
      }
  } else {
    int kb = k_lo + wave;
    // long K (mlp.2; one CU pulls its 160 KiB at ~17 GB/s, which is why this is the slowest of the small GEMMs: a
    // 4-way split-K variant with an ordered combine kernel was measured at the same 15 us and dropped): every weight block of the wave's share is requested at once (20 KiB per wave, one HBM
    // round trip); the activation fragments are L2 hits and are fetched five at a time right before use
    if constexpr (NT == 1)
    for (; kb + DG_WAVES * 19 < k_hi; kb += DG_WAVES * 20) {
      constexpr int UW = 20;
      vec8 w[NT][UW];
#pragma unroll
      for (int u = 0; u < UW; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t][u] = __builtin_nontemporal_load(&wt[t][(int64_t)(kb + DG_WAVES * u) * 64]);
#pragma unroll
      for (int g = 0; g < UW; g += 5) {
        vec8 a[5], b[5];
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) { a[u][e] = 0; if constexpr (MT == 2) b[u][e] = 0; }
        if (ok0) {
#pragma unroll
          for (int u = 0; u < 5; ++u) a[u] = *(const vec8*)(x0 + (int64_t)(kb + DG_WAVES * (g + u)) * 512);
        }
        if constexpr (MT == 2) {
          if (ok1) {
#pragma unroll
            for (int u = 0; u < 5; ++u) b[u] = *(const vec8*)(x1 + (int64_t)(kb + DG_WAVES * (g + u)) * 512);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t][0] = Ops::mfma16(w[t][g + u], a[u], acc[t][0]);
            if constexpr (MT == 2) acc[t][1] = Ops::mfma16(w[t][g + u], b[u], acc[t][1]);
          }
      }
    }
    if constexpr (NT == 1)
    for (; kb + DG_WAVES * 9 < k_hi; kb += DG_WAVES * 10) run(std::integral_constant<int, 10>{}, kb);
    for (; kb + DG_WAVES * 4 < k_hi; kb += DG_WAVES * 5) run(std::integral_constant<int, 5>{}, kb);
    for (; kb < k_hi; kb += DG_WAVES) run(std::integral_constant<int, 1>{}, kb);
  }

  if (pn) {
    // row statistics from the tiles loaded at the start: lane l merges tiles l, l + 32, l + 64 ... in order (Chan's
    // update), then the xor tree 16 .. 1; only lane 0's result is used - one fixed function of the tiles whatever MT is
    float* pst = (float*)ylds;
    const int l32 = tid & 31, prow = tid >> 5;
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int u = 0; u < PN_TILES; ++u) {
        if (l32 + 32 * u < p.n_stat) {
          const float nn = cnt + 16.f, delta = pn_mb[q][u] - mean;
          mean += delta * (16.f / nn);
          m2 += pn_qb[q][u] + delta * delta * (cnt * 16.f / nn);
          cnt = nn;
        }
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float cb = __shfl_xor(cnt, o, 64), mb = __shfl_xor(mean, o, 64), qb = __shfl_xor(m2, o, 64);
        const float nn = cnt + cb;
        if (nn > 0.f) {
          const float delta = mb - mean;
          mean += delta * (cb / nn);
          m2 += qb + delta * delta * (cnt * cb / nn);
        }
        cnt = nn;
      }
      if (l32 == 0) {
        pst[(q * 16 + prow) * 2] = mean;
        pst[(q * 16 + prow) * 2 + 1] = rsqrtf(m2 / (float)p.K + 1e-5f);
      }
    }
    // visible to the epilogue behind the __syncthreads() that follows the partial-sum stores
  }
  TRACE(16 + EPI * 2 + (LN ? 1 : 0), 2);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    part[((wave * NT + t) * 2 + 0) * 64 + lane] = acc[t][0];
    if constexpr (MT == 2) part[((wave * NT + t) * 2 + 1) * 64 + lane] = acc[t][1];
  }
  __syncthreads();
  if constexpr (EPI == DEPI_BIAS_RESID && NT == 1 && MT == 2) {
    if (ksplit > 1) {
      // Cross-workgroup split-K without spinning and without float atomics: every slice writes its 32 x 16 partial
      // tile through to memory (sc1 stores, drained), takes a ticket, and the slice that draws the last ticket sums
      // all slices in slice order (bitwise reproducible whatever the arrival order) and applies the epilogue.
      // Hand-off form: cdna_hip_programming.md Guideline 16 (sc1 payload + drain + barrier + one agent-scope add;
      // the last adder reads with sc1 loads behind a workgroup barrier).
      __shared__ unsigned s_ticket;
      const unsigned tile = blockIdx.y * gridDim.x + blockIdx.x;
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.slab, 0, p.slab_bytes, 0x00020000);
      const int l2 = tid & 63, mt2 = (tid >> 6) & 1;
      if (tid < 128) {
        f32x4 sum = part[(0 * 2 + mt2) * 64 + l2];
#pragma unroll
        for (int w = 1; w < DG_WAVES; ++w) sum += part[(w * 2 + mt2) * 64 + l2];
        union { f32x4 f; u32x4 u; } cv; cv.f = sum;
        __builtin_amdgcn_raw_buffer_store_b128(cv.u, rs, (int)(((tile * ksplit + blockIdx.z) * 128 + tid) * 16), 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) s_ticket = __hip_atomic_fetch_add(p.ticket + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (s_ticket != (unsigned)ksplit - 1) return;
      if (tid == 0) __hip_atomic_store(p.ticket + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
      if (tid < 128) {
        f32x4 tot = {0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < ksplit; ++sl) {
          union { f32x4 f; u32x4 u; } cv;
          cv.u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((tile * ksplit + sl) * 128 + tid) * 16), 0, 16);
          tot += cv.f;
        }
        const int m = m0 + mt2 * 16 + (l2 & 15), n0 = nt0 * 16 + 4 * (l2 >> 4);
        if (m < p.M) {
          float* xo = (float*)p.out + (int64_t)m * p.ld_out + n0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < p.N) xo[r] = xo[r] + (tot[r] + (p.bias ? p.bias[n0 + r] : 0.f));
        }
      }
      TRACE(16 + EPI * 2, 3);
      return;
    }
  }
  // per n-tile: 2 m-tiles x 64 lanes x 4 regs = 512 outputs, one per thread
  // D layout: n = 4*(lane'>>4) + reg, m = mt*16 + (lane' & 15)
  const float* pp = (const float*)part;
  if constexpr (EPI == DEPI_BIAS_RESID && NT == 1) {
    if (stat_epi) {
      if (!s_active) return;
      const int ll = (s_c >> 2) * 16 + s_r, reg = s_c & 3;          // s_c even: both columns in one accumulator register pair
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int w = 0; w < DG_WAVES; ++w) {
        const float2 pv = *(const float2*)&pp[((w * 2 + s_mt) * 64 + ll) * 4 + reg];
        v0 += pv.x; v1 += pv.y;
      }
      const bool ok = s_m < p.M && s_n < p.N;
      if (p.bias && ok) { v0 += p.bias[s_n]; v1 += p.bias[s_n + 1]; }
      const float x0 = resid_old[0] + v0, x1 = resid_old2 + v1;
      if (ok) {
        *(float2*)((float*)p.out + (int64_t)s_m * p.ld_out + s_n) = float2{x0, x1};
        *(unsigned*)((T*)p.x16_out + act_tiled_offset(s_m, s_n, p.N)) = pack2<T>(x0, x1);
      }
      // this tile's 16 columns of the row (8 lanes x 2): mean and sum of squared deviations, two passes over registers
      float sum = x0 + x1;
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      const float mean = sum * (1.0f / 16.0f);
      float q2 = (x0 - mean) * (x0 - mean) + (x1 - mean) * (x1 - mean);
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, 64);
      if (s_c == 0 && s_m < p.M) *(float2*)(p.stat_out + ((int64_t)s_m * (p.N >> 4) + nt0) * 2) = float2{mean, q2};
      TRACE(16 + EPI * 2, 3);
      return;
    }
  }
  const float* pnst = (const float*)ylds;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < DG_WAVES; ++w) v += pp[(((w * NT + t) * 2 + e_mt) * 64 + e_ll) * 4 + e_reg];
    const int n = (nt0 + t) * 16 + 4 * (e_ll >> 4) + e_reg;
    const int m = e_m;
    if (!e_active || m >= p.M || n >= p.N || nt0 + t >= n_tiles) continue;
    if (pn) {
      const int r = e_mt * 16 + (e_ll & 15);
      v = pnst[2 * r + 1] * (v - pnst[2 * r] * p.wsum[n]);
    }
    if (p.bias) v += p.bias[n];
    if constexpr (EPI == DEPI_QKV) {
      const int d = p.d_model;
      const T v16 = (T)v;
      if (n < d) {
        if (p.attn_ticket) {      // the fused self-attention reads it in this launch: written through (sc1)
          __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffffffe, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v16), ro, (int)(((int64_t)m * d + n) * 2), 0, 16);
        } else {
          ((T*)p.out)[(int64_t)m * d + n] = v16;
        }
      } else {
        const int b = m / p.n_new, i = m % p.n_new;
        const int pos = p.n_past[b] + i;
        const int nn = n < 2 * d ? n - d : n - 2 * d;
        const int h = nn >> 6, dh = nn & 63;
        T* cache = (T*)(n < 2 * d ? p.k_cache : p.v_cache);
        if (pos < p.n_ctx) {
          const int64_t at = ((((int64_t)b * p.n_head + h) * p.n_ctx + pos) << 6) + dh;
          if (p.attn_ticket) {
            __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cache, 0, (unsigned)p.kv_bytes, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v16), rc, (int)(at * 2), 0, 16);
          } else {
            cache[at] = v16;
          }
        }
      }
    } else if constexpr (EPI == DEPI_BIAS_T) {
      ((T*)p.out)[(int64_t)m * p.ld_out + n] = (T)v;
    } else if constexpr (EPI == DEPI_BIAS_GELU_T) {
      ((T*)p.out)[act_tiled_offset(m, n, p.N)] = (T)gelu_erf(v);
    } else if constexpr (EPI == DEPI_BIAS_RESID) {
      ((float*)p.out)[(int64_t)m * p.ld_out + n] = resid_old[t] + v;
    } else if constexpr (EPI == DEPI_LOGITS) {
      if (m % p.n_new == p.n_new - 1) ((float*)p.out)[(int64_t)(m / p.n_new) * p.ld_out + n] = v;
    }
  }
  if constexpr (EPI == DEPI_QKV && (NT == 1 || NT == 2 || NT == 4)) {
    if (p.attn_ticket) {
      // Single-token steps of at most 16 rows: the masked self-attention of a head runs HERE, in the workgroup that publishes the
      // last of the head's q / k / v columns (12 / NT workgroups per head) - one launch less per layer (an option, OFF: measured
      // 1.6 us per layer slower than the boundary it removes, engine.hip).  Hand-off form as in the split-K path above: the columns were stored
      // write-through (sc1), every wave drains its stores, one agent-scope add per workgroup; the last adder reads q and the
      // cache with sc1 loads (self_attn_row<COH>): same arithmetic, same order, same bits as self_attn_kernel.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                  // the partial tiles are consumed: their LDS serves from here on
      volatile unsigned* s_attn_ticket = (volatile unsigned*)((float*)part + DG_WAVES * 128);   // (no static LDS: the LN image already takes the CU's 160 KB)
      const int sec_tiles = p.d_model / 16;
      const int head = ((nt0 % sec_tiles) * 16) >> 6;
      if (tid == 0) *s_attn_ticket = __hip_atomic_fetch_add(p.attn_ticket + head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (*s_attn_ticket != 3u * (4 / NT) - 1u) return;
      if (tid == 0) __hip_atomic_store(p.attn_ticket + head, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
      float* qs = (float*)part + wave * 128;          // 512 B of LDS per wave
      float* ps = qs + 64;
      for (int m = wave; m < p.M; m += DG_WAVES) {
        int n_keys = p.n_past[m] + 1;
        if (n_keys > p.n_ctx) n_keys = p.n_ctx;
        if (p.attn_slots)
          self_attn_row<T, true, true>((const T*)p.out, (const T*)p.k_cache, (const T*)p.v_cache, n_keys, m, m, head, p.n_head, p.n_ctx,
                                       p.attn_slots + (int64_t)m * p.n_ctx, (T*)p.attn_out, qs, ps, lane, (unsigned)p.kv_bytes);
        else
          self_attn_row<T, false, true>((const T*)p.out, (const T*)p.k_cache, (const T*)p.v_cache, n_keys, m, m, head, p.n_head, p.n_ctx,
                                        nullptr, (T*)p.attn_out, qs, ps, lane, (unsigned)p.kv_bytes);
      }
    }
  }
  TRACE(16 + EPI * 2 + (LN ? 1 : 0), 3);
}

template <typename T, int EPI, bool LN, int NT, int MT>
static void dec_gemm_launch(const DecGemmParams& p, hipStream_t s) {
  const int n_tiles = (p.N + 15) / 16;
  dim3 grid((n_tiles + NT - 1) / NT, (p.M + 16 * MT - 1) / (16 * MT), p.ksplit > 1 ? p.ksplit : 1);
  const size_t smem = (size_t)DG_WAVES * NT * 2 * 64 * 16 + (LN ? (size_t)16 * MT * (p.K * 2 + 16) : 256);
  if (LN) ensure_dynamic_lds((const void*)dec_gemm_kernel<T, EPI, LN, NT, MT>, 160 * 1024);
  hipLaunchKernelGGL((dec_gemm_kernel<T, EPI, LN, NT, MT>), grid, dim3(DG_THREADS), smem, s, p);
}

template <typename T, int EPI, bool LN>
static void dec_gemm_pick(const DecGemmParams& p, hipStream_t s) {
  // with the LayerNorm image in LDS one workgroup fills a CU: keep the grid within one wave of the CUs this
  // launch may use (256, or the decoder's share when two batches are in flight)
  const int n_tiles = (p.N + 15) / 16;
  static const int logits_nt = getenv("OHW_LOGITS_NT") ? atoi(getenv("OHW_LOGITS_NT")) : 2;
  static const int msplit = getenv("OHW_DEC_MSPLIT") ? atoi(getenv("OHW_DEC_MSPLIT")) : 1;
  const int cus = p.cu_budget > 0 ? p.cu_budget : 256;
  const int mt = (p.M + 15) / 16;          // 16-row m-tiles
  if constexpr (EPI == DEPI_BIAS_RESID && !LN) {
    // one m-tile per workgroup when that still fits one wave of CUs (and the hand-off path is not in use); a single
    // m-tile (batch <= 16) never takes the two-tile kernel, whose second tile would be loaded for nothing
    if (msplit && p.ksplit <= 1 && (mt == 1 || n_tiles * mt <= 2 * cus)) { dec_gemm_launch<T, EPI, LN, 1, 1>(p, s); return; }
  }
  if constexpr (!LN && (EPI == DEPI_QKV || EPI == DEPI_BIAS_T || EPI == DEPI_BIAS_GELU_T)) {
    // post-norm consumers: the same grid rule as the LayerNorm variants (one workgroup per CU: at 512 threads and up to
    // 256 registers a second workgroup does not fit beside it; a two-round grid doubled the QKV launch, 4.9 -> 8.9 us)
    if (p.pn && msplit && mt <= 2) {
      if (n_tiles * mt <= cus) { dec_gemm_launch<T, EPI, LN, 1, 1>(p, s); return; }
      if ((n_tiles + 1) / 2 * mt <= cus || mt == 1) { dec_gemm_launch<T, EPI, LN, 2, 1>(p, s); return; }
    }
    if (p.pn && n_tiles > cus) { dec_gemm_launch<T, EPI, LN, 2, 2>(p, s); return; }
  }
  if constexpr (LN) {
    // one m-tile per workgroup halves the fp32 rows a workgroup normalises; pick the n-tiles per workgroup that keep
    // the grid within one wave of CUs
    if (msplit && mt <= 2) {
      if (n_tiles * mt <= cus) { dec_gemm_launch<T, EPI, LN, 1, 1>(p, s); return; }
      if ((n_tiles + 1) / 2 * mt <= cus || mt == 1) { dec_gemm_launch<T, EPI, LN, 2, 1>(p, s); return; }
    }
  }
  if constexpr (EPI == DEPI_LOGITS) {
    if (msplit && mt == 1 && logits_nt == 2) { dec_gemm_launch<T, EPI, LN, 2, 1>(p, s); return; }
  }
  // many rows on few CUs (a lane of the LANES schedule: 96 rows on 64 CUs): when the largest grid above would still take
  // more than two rounds of the CUs, a workgroup takes four n-tiles (QKV at 96 rows: 360 workgroups -> 180; the
  // out-projections: 240 -> 60, one round).  A tile's arithmetic does not depend on which workgroup computes it.
  static const int nt4 = getenv("OHW_DEC_NT4") ? atoi(getenv("OHW_DEC_NT4")) : 1;
  if constexpr ((LN && (EPI == DEPI_QKV || EPI == DEPI_BIAS_T || EPI == DEPI_BIAS_GELU_T)) || (!LN && (EPI == DEPI_BIAS_RESID || EPI == DEPI_LOGITS))) {
    const int64_t wgs = LN ? (int64_t)((n_tiles + 1) / 2) * ((mt + 1) / 2) : (int64_t)n_tiles * ((mt + 1) / 2);
    if (nt4 && mt > 2 && p.ksplit <= 1 && !p.pn && !p.stat_out && p.K <= DG_LN_MAXK && wgs > 2 * cus) { dec_gemm_launch<T, EPI, LN, 4, 2>(p, s); return; }
    // (mlp.2, K = 5120, keeps its single-tile workgroups: two tiles per workgroup measured the same within the run-to-run noise)
  }
  if ((LN && n_tiles > cus) || (EPI == DEPI_LOGITS && logits_nt == 2)) dec_gemm_launch<T, EPI, LN, 2, 2>(p, s);
  else dec_gemm_launch<T, EPI, LN, 1, 2>(p, s);
}

template <typename T>
void launch_dec_gemm(const DecGemmParams& p, int epilogue, hipStream_t s) {
  if (p.K % 32 != 0) throw Error(OHW_E_INVALID_ARG, "dec_gemm: K must be a multiple of 32");
  const bool ln = p.ln != 0;
  if (p.ksplit > 1) {
    const int64_t tiles = (int64_t)((p.N + 15) / 16) * ((p.M + 31) / 32);
    if (epilogue != DEPI_BIAS_RESID || !p.slab || !p.ticket || tiles * p.ksplit * 2048 > p.slab_bytes || p.ksplit > p.K / 32)
      throw Error(OHW_E_INVALID_ARG, "dec_gemm: split-K needs the RESID epilogue and a slab of tiles * ksplit * 2 KiB");
  }
  if (ln && (p.K % 64 != 0 || p.K > DG_LN_MAXK)) throw Error(OHW_E_INVALID_ARG, "dec_gemm: fused LayerNorm needs K <= 1280, K % 64 == 0");
  if (p.pn && (ln || !p.stat_in || !p.wsum || p.n_stat * 16 != p.K || p.n_stat > 128)) throw Error(OHW_E_INVALID_ARG, "dec_gemm: post-norm needs statistics of K / 16 tiles per row and the weights' row sums");
  if (p.stat_out && (epilogue != DEPI_BIAS_RESID || !p.x16_out || p.N % 32 != 0 || p.ksplit > 1))
    throw Error(OHW_E_INVALID_ARG, "dec_gemm: statistics come from the unsplit RESID epilogue with N % 32 == 0");
  switch (epilogue) {
    case DEPI_QKV: if (ln) dec_gemm_pick<T, DEPI_QKV, true>(p, s); else dec_gemm_pick<T, DEPI_QKV, false>(p, s); break;
    case DEPI_BIAS_T: if (ln) dec_gemm_pick<T, DEPI_BIAS_T, true>(p, s); else dec_gemm_pick<T, DEPI_BIAS_T, false>(p, s); break;
    case DEPI_BIAS_GELU_T: if (ln) dec_gemm_pick<T, DEPI_BIAS_GELU_T, true>(p, s); else dec_gemm_pick<T, DEPI_BIAS_GELU_T, false>(p, s); break;
    case DEPI_BIAS_RESID: if (ln) throw Error(OHW_E_INVALID_ARG, "dec_gemm: no LN variant"); dec_gemm_pick<T, DEPI_BIAS_RESID, false>(p, s); break;
    case DEPI_LOGITS: if (ln) throw Error(OHW_E_INVALID_ARG, "dec_gemm: no LN variant"); dec_gemm_pick<T, DEPI_LOGITS, false>(p, s); break;
    default: throw Error(OHW_E_INVALID_ARG, "dec_gemm: unknown epilogue");
  }
  HIP_CHECK(hipGetLastError());
}

// One wave, one (row, head).  Keys are processed in chunks of 64: lane j owns key (chunk*64 + j) for the
// score, lane = dh for P.V.  Every global load of a chunk (the lane's K row: 8 x 16 B, and the chunk's V
// column slice: 64 x 2 B) is requested before anything waits: one memory round trip per chunk.
// SLOTS (beam search): kv_slot [rows][n_ctx] - the cache row that holds position j of this row's sequence: beams that
// continue another beam share its past through this table instead of copying K/V.  A separate instantiation: the
// per-key lookups cost the greedy path 2 us per launch (5.9 -> 8.0 us) when they were a run-time branch.
// COH: q and the cache rows may have been written by OTHER workgroups of the same launch (the fused form inside the QKV
// launch): every load of them is a write-through-coherent one (buffer loads with sc1); arithmetic and order are the same, so
// the two forms give the same bits.
template <typename T, bool SLOTS, bool COH>
__device__ __forceinline__ void self_attn_row(const T* __restrict__ q, const T* __restrict__ kc, const T* __restrict__ vc, int n_keys, int b, int m, int h,
                                              int n_head, int n_ctx, const int32_t* __restrict__ slots, T* __restrict__ out, float* qs, float* ps,
                                              int lane, unsigned kv_bytes) {
  const int d = n_head * 64;
  const int64_t row_stride = (int64_t)n_head * n_ctx << 6;
  const T* kb = kc + ((int64_t)h * n_ctx << 6);
  const T* vb = vc + ((int64_t)h * n_ctx << 6);
  __amdgpu_buffer_rsrc_t rq, rk, rv;
  if constexpr (COH) {
    rq = __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, 0x7ffffffe, 0x00020000);
    rk = __builtin_amdgcn_make_buffer_rsrc((void*)kc, 0, kv_bytes, 0x00020000);
    rv = __builtin_amdgcn_make_buffer_rsrc((void*)vc, 0, kv_bytes, 0x00020000);
    const unsigned short qr = __builtin_amdgcn_raw_buffer_load_b16(rq, (int)(((int64_t)m * d + h * 64 + lane) * 2), 0, 16);
    qs[lane] = (float)__builtin_bit_cast(T, qr) * 0.125f;
  } else {
    qs[lane] = (float)q[(int64_t)m * d + h * 64 + lane] * 0.125f;
  }
  float m_run = -INFINITY, l_run = 0.f, o = 0.f;
  for (int c0 = 0; c0 < n_keys; c0 += 64) {
    const int nk = n_keys - c0 < 64 ? n_keys - c0 : 64;
    const int jk = lane < nk ? c0 + lane : c0;          // clamped: unconditional loads
    // SLOTS: ONE table load per lane and chunk (its own key's row); the V loads take key j's row from lane j's register
    // (v_readlane) - round 2 loaded the table 64 more times through the scalar cache, a wait each (6.8 us per launch
    // against 2.9 without the table, profiles/r03_streaming_beam5_graphs_on_kernel_stats.csv)
    const int sk = SLOTS ? slots[jk] : b;
    vec8_t<T> kr[8];
    if constexpr (COH) {
      const int kof = (int)((((int64_t)h * n_ctx << 6) + sk * row_stride + ((int64_t)jk << 6)) * 2);
#pragma unroll
      for (int c = 0; c < 8; ++c) kr[c] = __builtin_bit_cast(vec8_t<T>, __builtin_amdgcn_raw_buffer_load_b128(rk, kof + c * 16, 0, 16));
    } else {
      const vec8_t<T>* kp = (const vec8_t<T>*)(kb + sk * row_stride + ((int64_t)jk << 6));
#pragma unroll
      for (int c = 0; c < 8; ++c) kr[c] = kp[c];
    }
    T vr[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const int jj = j < nk ? c0 + j : c0;
      const int sv = SLOTS ? __builtin_amdgcn_readlane(sk, j < nk ? j : 0) : b;
      if constexpr (COH) {
        const unsigned short vraw = __builtin_amdgcn_raw_buffer_load_b16(rv, (int)((((int64_t)h * n_ctx << 6) + sv * row_stride + ((int64_t)jj << 6) + lane) * 2), 0, 16);
        vr[j] = __builtin_bit_cast(T, vraw);
      } else {
        vr[j] = vb[sv * row_stride + ((int64_t)jj << 6) + lane];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // qs (first chunk) / ps of the previous chunk consumed (one wave: program order)
    __builtin_amdgcn_wave_barrier();
    float sc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) sc += qs[c * 8 + e] * (float)kr[c][e];
    if (lane >= nk) sc = -INFINITY;
    const float m_new = fmaxf(m_run, wave_max(sc));
    const float alpha = __expf(m_run - m_new);
    const float pe = lane < nk ? __expf(sc - m_new) : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ps[lane] = pe;
    l_run = l_run * alpha + wave_sum(pe);
    m_run = m_new;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) acc += ps[j] * (float)vr[j];   // masked keys carry p = 0
    o = o * alpha + acc;
  }
  out[act_tiled_offset(m, h * 64 + lane, d)] = (T)(o / l_run);
}

// ------------------------------------------------------------------------------------------------
// masked self-attention over the KV cache: one wave per (row m, head)
// ------------------------------------------------------------------------------------------------
template <typename T, bool SLOTS>
__global__ __launch_bounds__(64) void self_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc, const T* __restrict__ vc,
                                                       const int32_t* __restrict__ n_past, T* __restrict__ out, int n_new,
                                                       int n_head, int n_ctx, const int32_t* __restrict__ kv_slot) {
  __shared__ float qs[64];
  __shared__ float ps[64];
  const int lane = threadIdx.x;
  const int h = blockIdx.x, m = blockIdx.y;
  const int b = m / n_new, i = m % n_new;
  TRACE(2, 0);
  int n_keys = n_past[b] + i + 1;
  if (n_keys > n_ctx) n_keys = n_ctx;
  self_attn_row<T, SLOTS, false>(q, kc, vc, n_keys, b, m, h, n_head, n_ctx, SLOTS ? kv_slot + (int64_t)b * n_ctx : nullptr, out, qs, ps, lane, 0u);
  TRACE(2, 3);
}
template <typename T>
void launch_self_attn(const void* q, const void* k_cache, const void* v_cache, const int32_t* n_past, void* out, int M, int n_new,
                      int n_head, int n_ctx, hipStream_t s, const int32_t* kv_slot) {
  if (kv_slot)
    hipLaunchKernelGGL((self_attn_kernel<T, true>), dim3(n_head, M), dim3(64), 0, s, (const T*)q, (const T*)k_cache, (const T*)v_cache, n_past,
                       (T*)out, n_new, n_head, n_ctx, kv_slot);
  else
    hipLaunchKernelGGL((self_attn_kernel<T, false>), dim3(n_head, M), dim3(64), 0, s, (const T*)q, (const T*)k_cache, (const T*)v_cache, n_past,
                       (T*)out, n_new, n_head, n_ctx, kv_slot);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// cross-attention of one query row against 1500 encoder positions: one workgroup per (row, head).
// 8 lanes share a key (16 B of the 128-B K/V row each), so every wave-instruction covers 8 keys =
// 1 KiB contiguous; the 4 waves interleave 8-key groups.  Each 8-lane group keeps its own online
// softmax state (no cross-lane traffic in the loop beyond the 8-lane dot reduction); the 32 partial
// states are merged once at the end.
// ------------------------------------------------------------------------------------------------
constexpr int XA_THREADS = 256;
constexpr int XA_UNROLL = 4;

template <typename T>
__global__ __launch_bounds__(XA_THREADS) void cross_attn_kernel(const T* __restrict__ q, const T* __restrict__ xk, const T* __restrict__ xv,
                                                                T* __restrict__ out, int n_new, int n_head, int t_len,
                                                                float* __restrict__ partials, unsigned* __restrict__ tickets,
                                                                const int32_t* __restrict__ done) {
  // gridDim.z > 1 (small batches: fewer than a wave of (row, head) pairs): the keys of one (row, head) are cut into
  // gridDim.z contiguous slices on as many CUs - one CU pulls only ~25 GB/s of a 384 KB stream; each slice publishes
  // its (max, sum, 64-vector) state, the workgroup that draws the last ticket merges them in slice order.
  __shared__ float red_m[4], red_l[4];
  __shared__ float red_o[4][64];
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, m = blockIdx.y;
  const int b = m / n_new;
  // a window that has produced its end-of-text token still rides along in the batch: its rows are computed by the GEMMs
  // (free) but its 7.7 MB of cross K/V per layer are not streamed (its output row is never used again)
  if (done && done[b]) return;
  const int d = n_head * 64;
  const int part = lane & 7, slot = lane >> 3;
  const float sc = 0.125f * 1.44269504088896340736f;
  TRACE(3, 0);
  float qv[8];
  {
    const vec8_t<T> qq = *(const vec8_t<T>*)(q + (int64_t)m * d + h * 64 + part * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = (float)qq[e] * sc;
  }
  const T* kb = xk + (((int64_t)b * n_head + h) * t_len << 6) + part * 8;
  const T* vb = xv + (((int64_t)b * n_head + h) * t_len << 6) + part * 8;
  float mrun = -INFINITY, lrun = 0.f;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int n_groups_all = (t_len + 7) / 8;
  const int per_slice = (n_groups_all + gridDim.z - 1) / gridDim.z;
  const int g_lo = blockIdx.z * per_slice;
  const int n_groups = g_lo + per_slice < n_groups_all ? g_lo + per_slice : n_groups_all;
  for (int g0 = g_lo + wave; g0 < n_groups; g0 += 4 * XA_UNROLL) {
    vec8_t<T> kf[XA_UNROLL], vf[XA_UNROLL];
    int keys[XA_UNROLL];
#pragma unroll
    for (int u = 0; u < XA_UNROLL; ++u) {
      const int g = g0 + 4 * u;
      int key = g * 8 + slot;
      keys[u] = (g < n_groups && key < t_len) ? key : -1;
      if (key > t_len - 1) key = t_len - 1;
      kf[u] = __builtin_nontemporal_load((const vec8_t<T>*)(kb + ((int64_t)key << 6)));
      vf[u] = __builtin_nontemporal_load((const vec8_t<T>*)(vb + ((int64_t)key << 6)));
    }
#pragma unroll
    for (int u = 0; u < XA_UNROLL; ++u) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += qv[e] * (float)kf[u][e];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (keys[u] < 0) s = -INFINITY;
      const float mn = fmaxf(mrun, s);
      // mn == -inf only while every key so far was masked: keep the state untouched then
      const float alpha = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(mrun - mn);
      const float pe = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s - mn);
      mrun = mn;
      lrun = lrun * alpha + pe;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = acc[e] * alpha + pe * (float)vf[u][e];
    }
  }
  TRACE(3, 2);
  // merge the 8 slots of this wave (lanes with equal `part`): xor 8, 16, 32
#pragma unroll
  for (int o = 8; o <= 32; o <<= 1) {
    const float m2 = __shfl_xor(mrun, o, 64), l2 = __shfl_xor(lrun, o, 64);
    const float mn = fmaxf(mrun, m2);
    const float a1 = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(mrun - mn);
    const float a2 = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m2 - mn);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float o2 = __shfl_xor(acc[e], o, 64);
      acc[e] = acc[e] * a1 + o2 * a2;
    }
    lrun = lrun * a1 + l2 * a2;
    mrun = mn;
  }
  if (slot == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red_o[wave][part * 8 + e] = acc[e];
    if (part == 0) { red_m[wave] = mrun; red_l[wave] = lrun; }
  }
  __syncthreads();
  float mn = 0.f, l = 0.f, o = 0.f;
  if (tid < 64) {
    mn = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float a = red_m[w] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(red_m[w] - mn);
      l += red_l[w] * a;
      o += red_o[w][tid] * a;
    }
  }
  if (gridDim.z > 1) {
    // publish {o[64], max, sum} of this slice (wave 0 only holds it), ticket, last arriver merges in slice order
    const int KS = gridDim.z;
    unsigned* slot = (unsigned*)partials + (((int64_t)m * n_head + h) * KS + blockIdx.z) * 68;
    if (tid < 64) {
      __hip_atomic_store(slot + tid, __float_as_uint(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 0) {
        __hip_atomic_store(slot + 64, __float_as_uint(mn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 65, __float_as_uint(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) s_ticket = __hip_atomic_fetch_add(tickets + m * n_head + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_ticket != (unsigned)KS - 1 || tid >= 64) return;
    if (tid == 0) __hip_atomic_store(tickets + m * n_head + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every load of the merge is issued before the first is used (sc1 buffer loads; a chain of waited atomic loads
    // would be 3 * KS dependent round trips)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(partials + ((int64_t)m * n_head + h) * KS * 68), 0, KS * 68 * 4, 0x00020000);
    float mk[XA_MAX_SPLIT], lk[XA_MAX_SPLIT], ok[XA_MAX_SPLIT];
#pragma unroll
    for (int k = 0; k < XA_MAX_SPLIT; ++k) {
      const int kk = k < KS ? k : 0;
      mk[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (kk * 68 + 64) * 4, 0, 16));
      lk[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (kk * 68 + 65) * 4, 0, 16));
      ok[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (kk * 68 + tid) * 4, 0, 16));
    }
    float mm = -INFINITY;
#pragma unroll
    for (int k = 0; k < XA_MAX_SPLIT; ++k) if (k < KS) mm = fmaxf(mm, mk[k]);
    l = 0.f; o = 0.f;
#pragma unroll
    for (int k = 0; k < XA_MAX_SPLIT; ++k) {
      if (k < KS) {
        const float a = mk[k] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mk[k] - mm);
        l += lk[k] * a;
        o += ok[k] * a;
      }
    }
  }
  if (tid < 64) out[act_tiled_offset(m, h * 64 + tid, d)] = (T)(o / l);
  TRACE(3, 3);
}
// The prompt pass feeds n_new = 2..4 tokens per window at once: their query rows read the SAME 7.7 MB of cross K/V per
// layer, so one workgroup takes all NQ rows of a (window, head) and streams K/V once (row by row it was read NQ times:
// 3 x 246 MB per layer at 32 windows).  Per row the arithmetic and its order are those of cross_attn_kernel (bit-identical).
template <typename T, int NQ>
__global__ __launch_bounds__(XA_THREADS) void cross_attn_rows_kernel(const T* __restrict__ q, const T* __restrict__ xk, const T* __restrict__ xv,
                                                                     T* __restrict__ out, int n_head, int t_len,
                                                                     const int32_t* __restrict__ done) {
  __shared__ float red_m[NQ][4], red_l[NQ][4];
  __shared__ float red_o[NQ][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  if (done && done[b]) return;
  const int d = n_head * 64;
  const int part = lane & 7, slot = lane >> 3;
  const float sc = 0.125f * 1.44269504088896340736f;
  float qv[NQ][8], mrun[NQ], lrun[NQ], acc[NQ][8];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const vec8_t<T> qq = *(const vec8_t<T>*)(q + (int64_t)(b * NQ + i) * d + h * 64 + part * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { qv[i][e] = (float)qq[e] * sc; acc[i][e] = 0.f; }
    mrun[i] = -INFINITY; lrun[i] = 0.f;
  }
  const T* kb = xk + (((int64_t)b * n_head + h) * t_len << 6) + part * 8;
  const T* vb = xv + (((int64_t)b * n_head + h) * t_len << 6) + part * 8;
  const int n_groups = (t_len + 7) / 8;
  for (int g0 = wave; g0 < n_groups; g0 += 4 * XA_UNROLL) {
    vec8_t<T> kf[XA_UNROLL], vf[XA_UNROLL];
    int keys[XA_UNROLL];
#pragma unroll
    for (int u = 0; u < XA_UNROLL; ++u) {
      const int g = g0 + 4 * u;
      int key = g * 8 + slot;
      keys[u] = (g < n_groups && key < t_len) ? key : -1;
      if (key > t_len - 1) key = t_len - 1;
      kf[u] = __builtin_nontemporal_load((const vec8_t<T>*)(kb + ((int64_t)key << 6)));
      vf[u] = __builtin_nontemporal_load((const vec8_t<T>*)(vb + ((int64_t)key << 6)));
    }
#pragma unroll
    for (int u = 0; u < XA_UNROLL; ++u) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += qv[i][e] * (float)kf[u][e];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (keys[u] < 0) s = -INFINITY;
        const float mn = fmaxf(mrun[i], s);
        const float alpha = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(mrun[i] - mn);
        const float pe = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s - mn);
        mrun[i] = mn;
        lrun[i] = lrun[i] * alpha + pe;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = acc[i][e] * alpha + pe * (float)vf[u][e];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
#pragma unroll
    for (int o = 8; o <= 32; o <<= 1) {
      const float m2 = __shfl_xor(mrun[i], o, 64), l2 = __shfl_xor(lrun[i], o, 64);
      const float mn = fmaxf(mrun[i], m2);
      const float a1 = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(mrun[i] - mn);
      const float a2 = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m2 - mn);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float o2 = __shfl_xor(acc[i][e], o, 64);
        acc[i][e] = acc[i][e] * a1 + o2 * a2;
      }
      lrun[i] = lrun[i] * a1 + l2 * a2;
      mrun[i] = mn;
    }
    if (slot == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red_o[i][wave][part * 8 + e] = acc[i][e];
      if (part == 0) { red_m[i][wave] = mrun[i]; red_l[i][wave] = lrun[i]; }
    }
  }
  __syncthreads();
  // wave i finishes row i (and row i + 4 when NQ = 5)
  for (int i = wave; i < NQ; i += 4) {
    const float mn = fmaxf(fmaxf(red_m[i][0], red_m[i][1]), fmaxf(red_m[i][2], red_m[i][3]));
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float a = red_m[i][w] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(red_m[i][w] - mn);
      l += red_l[i][w] * a;
      o += red_o[i][w][lane] * a;
    }
    out[act_tiled_offset(b * NQ + i, h * 64 + lane, d)] = (T)(o / l);
  }
}

template <typename T>
void launch_cross_attn(const void* q, const void* xk, const void* xv, void* out, int M, int n_new, int n_head, int t_len, float* partials,
                       unsigned* tickets, int max_split_rows, const int32_t* done, hipStream_t s, int kv_group, bool batch_invariant) {
  // batch_invariant: the kernel variant (and with it the order of the softmax reduction) is picked from n_new alone, never
  // from the number of rows, so a window's result does not depend on how many windows share its batch
  // beam search: kv_group consecutive rows are the beams of ONE window and read the same cross K/V: one workgroup per
  // (window, head) streams it once for all of them (done: per window)
  if (kv_group > 1) {
    if (n_new != 1 || M % kv_group != 0 || kv_group > 5) throw Error(OHW_E_INVALID_ARG, "cross-attention: beams are single-token rows, at most 5 per window");
    // a few windows (streaming: ONE): n_head workgroups would each pull a 384-KB stream alone (48.6 us per launch at one
    // window, half of the beam step); the general kernel with n_new = kv_group reads window m / kv_group for row m and
    // cuts the keys of every (row, head) over up to 8 workgroups instead (the K/V bytes are a few MB here)
    if (!batch_invariant && partials && tickets && M <= max_split_rows && (int64_t)(M / kv_group) * n_head < 128) {
      int ks = 1;
      while (ks < XA_MAX_SPLIT && (int64_t)M * n_head * ks < 512 && t_len / (ks * 2) >= 64) ks *= 2;
      hipLaunchKernelGGL((cross_attn_kernel<T>), dim3(n_head, M, ks), dim3(XA_THREADS), 0, s, (const T*)q, (const T*)xk, (const T*)xv, (T*)out, kv_group,
                         n_head, t_len, partials, tickets, done);
      HIP_CHECK(hipGetLastError());
      return;
    }
    const dim3 grid(n_head, M / kv_group);
    switch (kv_group) {
      case 2: hipLaunchKernelGGL((cross_attn_rows_kernel<T, 2>), grid, dim3(XA_THREADS), 0, s, (const T*)q, (const T*)xk, (const T*)xv, (T*)out, n_head, t_len, done); break;
      case 3: hipLaunchKernelGGL((cross_attn_rows_kernel<T, 3>), grid, dim3(XA_THREADS), 0, s, (const T*)q, (const T*)xk, (const T*)xv, (T*)out, n_head, t_len, done); break;
      case 4: hipLaunchKernelGGL((cross_attn_rows_kernel<T, 4>), grid, dim3(XA_THREADS), 0, s, (const T*)q, (const T*)xk, (const T*)xv, (T*)out, n_head, t_len, done); break;
      default: hipLaunchKernelGGL((cross_attn_rows_kernel<T, 5>), grid, dim3(XA_THREADS), 0, s, (const T*)q, (const T*)xk, (const T*)xv, (T*)out, n_head, t_len, done); break;
    }
    HIP_CHECK(hipGetLastError());
    return;
  }
  // fewer (row, head) pairs than two per CU: cut the keys (at most 8 slices, each at least a few hundred keys)
  int ks = 1;
  if (partials && tickets && M <= max_split_rows && !batch_invariant) {
    while (ks < XA_MAX_SPLIT && (int64_t)M * n_head * ks < 512 && t_len / (ks * 2) >= 64) ks *= 2;
  }
  // several new tokens per window and enough windows to fill the chip: one workgroup per (window, head) streams K/V once
  if (n_new >= 2 && n_new <= 4 && M % n_new == 0 && ((int64_t)(M / n_new) * n_head >= 256 || batch_invariant)) {
    const dim3 grid(n_head, M / n_new);
    if (n_new == 2) hipLaunchKernelGGL((cross_attn_rows_kernel<T, 2>), grid, dim3(XA_THREADS), 0, s, (const T*)q, (const T*)xk, (const T*)xv, (T*)out, n_head, t_len, done);
    else if (n_new == 3) hipLaunchKernelGGL((cross_attn_rows_kernel<T, 3>), grid, dim3(XA_THREADS), 0, s, (const T*)q, (const T*)xk, (const T*)xv, (T*)out, n_head, t_len, done);
    else hipLaunchKernelGGL((cross_attn_rows_kernel<T, 4>), grid, dim3(XA_THREADS), 0, s, (const T*)q, (const T*)xk, (const T*)xv, (T*)out, n_head, t_len, done);
    HIP_CHECK(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL((cross_attn_kernel<T>), dim3(n_head, M, ks), dim3(XA_THREADS), 0, s, (const T*)q, (const T*)xk, (const T*)xv, (T*)out, n_new,
                     n_head, t_len, partials, tickets, done);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// logits filter + greedy arg-max on the device: restates oracle ref_process_logits / ref_greedy
// (whisper.cpp defaults).  One workgroup per window.
// ------------------------------------------------------------------------------------------------
constexpr int SP_THREADS = 512;

struct SampState {
  int is_initial, last_ts, penult_ts, last_seen, suppress_eot;
};

__device__ __forceinline__ bool sp_allowed(const SamplerParams& p, const SampState& st, int i) {
  if (i == p.no_ts || i == p.sot || i == p.nosp || i == p.translate || i == p.transcribe || i == p.prev || i == p.solm) return false;
  if (i > p.sot && i <= p.sot + p.n_langs) return false;
  if (st.is_initial && p.suppress_blank && (i == p.eot || i == p.blank)) return false;
  if (st.suppress_eot && i == p.eot) return false;
  if (p.no_timestamps) return i < p.ts_begin;
  if (st.last_ts) {
    if (st.penult_ts) { if (i >= p.ts_begin) return false; }
    else { if (i < p.eot) return false; }
  }
  if (st.is_initial && p.max_initial_ts > 0 && i > p.ts_begin + p.max_initial_ts) return false;
  if (st.last_seen >= 0 && i >= p.ts_begin && i < st.last_seen) return false;
  return true;
}

// the last timestamp token among the n_cur tokens sampled so far (or -1): every thread looks at its share of the history and the
// workgroup keeps the latest hit - ONE memory round trip.  (Each thread walking the history backwards by itself paid a dependent
// round trip per token: 20 us of the sampler's launch and 40 us of the beam top-k's once a window had 50 tokens and no timestamp.)
template <int NTHREADS>
__device__ __forceinline__ int block_last_timestamp(const int32_t* toks, int n_cur, int ts_begin, int tid, int* sh /* [NTHREADS / 64] */) {
  int key = -1;                                     // (position << 16) | token: positions < 2^9, tokens < 2^16
  for (int i = tid; i < n_cur; i += NTHREADS) {
    const int t = toks[i];
    if (t >= ts_begin) key = (i << 16) | t;         // a thread's positions ascend: the last hit stays
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int ok = __shfl_xor(key, o, 64); key = ok > key ? ok : key; }
  if ((tid & 63) == 0) sh[tid >> 6] = key;
  __syncthreads();
  int r = sh[0];
#pragma unroll
  for (int k = 1; k < NTHREADS / 64; ++k) r = sh[k] > r ? sh[k] : r;
  __syncthreads();
  return r >= 0 ? (r & 0xffff) : -1;
}

struct SampAcc {
  float m, s_all, s_ts;     // online log-sum-exp state: sums are relative to m
  float tv; int ti;         // best text token  (value, index)
  float zv; int zi;         // best timestamp token
  float rm, rs;             // first step only: log-sum-exp state of the UNFILTERED row (no-speech probability)
};

__device__ __forceinline__ void samp_merge(SampAcc& a, const SampAcc& b) {
  const float mn = fmaxf(a.m, b.m);
  const float fa = a.m == -INFINITY ? 0.f : expf(a.m - mn);
  const float fb = b.m == -INFINITY ? 0.f : expf(b.m - mn);
  a.s_all = a.s_all * fa + b.s_all * fb;
  a.s_ts = a.s_ts * fa + b.s_ts * fb;
  a.m = mn;
  if (b.tv > a.tv || (b.tv == a.tv && b.ti < a.ti)) { a.tv = b.tv; a.ti = b.ti; }
  if (b.zv > a.zv || (b.zv == a.zv && b.zi < a.zi)) { a.zv = b.zv; a.zi = b.zi; }
  const float rn = fmaxf(a.rm, b.rm);
  const float ra = a.rm == -INFINITY ? 0.f : expf(a.rm - rn);
  const float rb = b.rm == -INFINITY ? 0.f : expf(b.rm - rn);
  a.rs = a.rs * ra + b.rs * rb;
  a.rm = rn;
}

// ONE pass over the logits row: masked online log-sum-exp (all / timestamps) and the best text and
// best timestamp candidates; the timestamp-mass rule then only chooses between the two candidates.
// The scan is VALU-bound (the rule test per token), so a window's row is cut into SAMPLER_SPLIT slices on as many
// CUs; each slice publishes its partial state (agent-scope stores, drained) and takes a ticket, and the workgroup that
// draws the last ticket merges the partials in slice order and applies the decision (no spinning, reproducible).
__global__ __launch_bounds__(SP_THREADS) void sampler_kernel(SamplerParams p) {
  __shared__ SampAcc sh[SP_THREADS / 64];
  const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  if (p.done[b]) return;
  const int np_now = p.n_past[b] + p.advance;
  const float* lg = p.logits + (int64_t)b * p.ld;
  int32_t* toks = p.tokens + (int64_t)b * p.max_tokens;
  const int n_cur = p.n_cur[b];
  SampState st;
  st.is_initial = n_cur == 0;
  st.last_ts = n_cur > 0 && toks[n_cur - 1] >= p.ts_begin;
  st.penult_ts = n_cur < 2 || toks[n_cur - 2] >= p.ts_begin;
  __shared__ int sh_ts[SP_THREADS / 64];
  st.last_seen = block_last_timestamp<SP_THREADS>(toks, n_cur, p.ts_begin, tid, sh_ts);
  st.suppress_eot = p.force_len > 0 && n_cur < p.force_len;
  const int per = (p.n_vocab + SAMPLER_SPLIT - 1) / SAMPLER_SPLIT;
  const int lo = part * per, V = lo + per < p.n_vocab ? lo + per : p.n_vocab;
  const bool want_raw = st.is_initial && p.nosp_prob != nullptr;
  SampAcc a;
  a.m = -INFINITY; a.s_all = 0.f; a.s_ts = 0.f; a.tv = -INFINITY; a.ti = 0x7fffffff; a.zv = -INFINITY; a.zi = 0x7fffffff;
  a.rm = -INFINITY; a.rs = 0.f;
  // batches of SP_BATCH loads per thread, all in flight before the first is used
  constexpr int SP_BATCH = 13;
  TRACE(4, 0);
  for (int i0 = lo + tid; i0 < V; i0 += SP_THREADS * SP_BATCH) {
    float vv[SP_BATCH];
#pragma unroll
    for (int u = 0; u < SP_BATCH; ++u) {
      const int i = i0 + u * SP_THREADS;
      vv[u] = lg[i < V ? i : V - 1];           // unconditional load (clamped address); masked below
    }
    if (p.bias) {
      float bb[SP_BATCH];
#pragma unroll
      for (int u = 0; u < SP_BATCH; ++u) {
        const int i = i0 + u * SP_THREADS;
        bb[u] = p.bias[i < V ? i : V - 1];
      }
#pragma unroll
      for (int u = 0; u < SP_BATCH; ++u) vv[u] += bb[u];
    }
    if (want_raw) {
#pragma unroll
      for (int u = 0; u < SP_BATCH; ++u) {
        const int i = i0 + u * SP_THREADS;
        const float v = vv[u];
        if (i >= V) continue;
        if (v > a.rm) { a.rs *= a.rm == -INFINITY ? 0.f : expf(a.rm - v); a.rm = v; }
        a.rs += expf(v - a.rm);
      }
    }
#pragma unroll
    for (int u = 0; u < SP_BATCH; ++u) {
      const int i = i0 + u * SP_THREADS;
      const float v = vv[u];
      if (i >= V || !sp_allowed(p, st, i)) continue;
      if (v > a.m) {
        const float f = a.m == -INFINITY ? 0.f : expf(a.m - v);
        a.s_all *= f; a.s_ts *= f; a.m = v;
      }
      const float e = expf(v - a.m);
      a.s_all += e;
      if (i >= p.ts_begin) { a.s_ts += e; if (v > a.zv) { a.zv = v; a.zi = i; } }
      else if (v > a.tv) { a.tv = v; a.ti = i; }
    }
  }
  TRACE(4, 2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    SampAcc bb;
    bb.m = __shfl_xor(a.m, o, 64); bb.s_all = __shfl_xor(a.s_all, o, 64); bb.s_ts = __shfl_xor(a.s_ts, o, 64);
    bb.tv = __shfl_xor(a.tv, o, 64); bb.ti = __shfl_xor(a.ti, o, 64); bb.zv = __shfl_xor(a.zv, o, 64); bb.zi = __shfl_xor(a.zi, o, 64);
    bb.rm = __shfl_xor(a.rm, o, 64); bb.rs = __shfl_xor(a.rs, o, 64);
    samp_merge(a, bb);
  }
  if ((tid & 63) == 0) sh[tid >> 6] = a;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < SP_THREADS / 64; ++w) samp_merge(a, sh[w]);
    // publish this slice, take a ticket; only the last arriver goes on (one thread does all of it, in program order)
    constexpr int NW = 9;
    unsigned* slot = (unsigned*)p.partials + ((int64_t)b * SAMPLER_SPLIT + part) * SAMPLER_PART_WORDS;
    const unsigned words[NW] = {__float_as_uint(a.m), __float_as_uint(a.s_all), __float_as_uint(a.s_ts), __float_as_uint(a.tv), (unsigned)a.ti,
                                __float_as_uint(a.zv), (unsigned)a.zi, __float_as_uint(a.rm), __float_as_uint(a.rs)};
#pragma unroll
    for (int k = 0; k < NW; ++k) __hip_atomic_store(slot + k, words[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(p.tickets + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == SAMPLER_SPLIT - 1) {
      __hip_atomic_store(p.tickets + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned* all = (const unsigned*)p.partials + (int64_t)b * SAMPLER_SPLIT * SAMPLER_PART_WORDS;
      unsigned w8[SAMPLER_SPLIT][NW];
#pragma unroll
      for (int q = 0; q < SAMPLER_SPLIT; ++q)
#pragma unroll
        for (int k = 0; k < NW; ++k) w8[q][k] = __hip_atomic_load(all + q * SAMPLER_PART_WORDS + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int q = 0; q < SAMPLER_SPLIT; ++q) {
        SampAcc c;
        c.m = __uint_as_float(w8[q][0]); c.s_all = __uint_as_float(w8[q][1]); c.s_ts = __uint_as_float(w8[q][2]);
        c.tv = __uint_as_float(w8[q][3]); c.ti = (int)w8[q][4]; c.zv = __uint_as_float(w8[q][5]); c.zi = (int)w8[q][6];
        c.rm = __uint_as_float(w8[q][7]); c.rs = __uint_as_float(w8[q][8]);
        if (q == 0) a = c; else samp_merge(a, c);
      }
      if (want_raw) {
        const float v = lg[p.nosp] + (p.bias ? p.bias[p.nosp] : 0.f);
        p.nosp_prob[b] = expf(v - (a.rm + logf(a.rs)));
      }
      const float lse = a.m + logf(a.s_all);
      bool force_ts = false;
      if (!p.no_timestamps && a.s_ts > 0.f) {
        const float ts_lp = a.m + logf(a.s_ts) - lse;
        const float text_lp = a.tv - lse;
        force_ts = ts_lp > text_lp;
      }
      // arg-max over what is left; text indices are below timestamp indices, so ties go to text
      float bv; int bi;
      if (force_ts || a.zv > a.tv) { bv = a.zv; bi = a.zi; } else { bv = a.tv; bi = a.ti; }
      const int n_max = p.force_len > 0 ? p.force_len : p.n_max;
      bool finished = false;
      if (p.tok_lp) p.tok_lp[(int64_t)b * (p.max_tokens + 1) + n_cur] = bv - lse;
      if (bi == p.eot) {
        finished = true;
        p.next_tok[b] = bi;
      } else {
        toks[n_cur] = bi;
        p.n_cur[b] = n_cur + 1;
        p.sum_logprob[b] += bv - lse;
        p.next_tok[b] = bi;
        if (p.advance) p.n_past[b] = np_now;
        const int np = np_now;
        if (n_cur + 1 >= n_max || n_cur + 1 >= p.max_tokens || np + 1 >= p.n_text_ctx) finished = true;
      }
      if (finished) { p.done[b] = 1; atomicAdd(p.n_done, 1); }
    }
  }
  TRACE(4, 3);
}
void launch_sampler(const SamplerParams& p, hipStream_t s) {
  hipLaunchKernelGGL(sampler_kernel, dim3(SAMPLER_SPLIT, p.batch), dim3(SP_THREADS), 0, s, p);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// beam search (BASELINE.json config #5).  The reference never uses beam search (Greedy{best_of:1},
// src/engine/whisper.rs:243); the rule restated here is the published Whisper BeamSearchDecoder (openai/whisper
// decoding.py): every live beam proposes its beam_size + 1 most likely next tokens (after the same logits filter),
// the candidates of a window are ranked by cumulative log-probability, sequences that end in end-of-text move to the
// window's finished pool (at most beam_size of them), the best beam_size others become the new beams.
// Rows: beam j of window w is decoder row w * K + j; the K rows of a window share its cross K/V (cross_attn_rows_kernel)
// and share their common past through the kv_slot table of self_attn_kernel instead of copying K/V.
// ------------------------------------------------------------------------------------------------
constexpr int BEAM_THREADS = 256;
constexpr int BEAM_PER_THREAD = 26;     // 8 slices x 256 threads x 26 >= 51866

// BEAM_SPLIT workgroups per row (round 2: one workgroup of 1024 threads per row, 78 us per launch at 5 rows): a slice computes the
// masked soft-max state of its part of the row and its K + 1 best text and K + 1 best timestamp tokens (ties: lowest index),
// publishes them and takes a ticket; the last arriver merges the slices in slice order, applies the timestamp-mass rule and keeps
// the K + 1 best of what is left - the same candidates as one pass over the whole row gives (a token among the row's K + 1 best
// of its kind is among its slice's).
__global__ __launch_bounds__(BEAM_THREADS) void beam_topk_kernel(SamplerParams p, BeamParams bp, int first) {
  __shared__ float red_v[BEAM_THREADS / 64];
  __shared__ int red_i[BEAM_THREADS / 64];
  __shared__ float sh_cv[2][6];
  __shared__ int sh_ci[2][6];
  const int tid = threadIdx.x, part = blockIdx.x;
  // the first step reads the prompt pass's logits: one row per WINDOW, candidates go to the window's beam 0
  const int lrow = blockIdx.y;
  const int row = first ? lrow * bp.K : lrow;
  const int w = row / bp.K;
  if (bp.win_done[w]) return;
  const int K1 = bp.K + 1;
  const float* lg = p.logits + (int64_t)lrow * p.ld;
  const int32_t* toks = p.tokens + (int64_t)row * p.max_tokens;
  const int n_cur = bp.n_cur[w];
  SampState st;
  st.is_initial = n_cur == 0;
  st.last_ts = n_cur > 0 && toks[n_cur - 1] >= p.ts_begin;
  st.penult_ts = n_cur < 2 || toks[n_cur - 2] >= p.ts_begin;
  __shared__ int sh_ts[BEAM_THREADS / 64];
  st.last_seen = block_last_timestamp<BEAM_THREADS>(toks, n_cur, p.ts_begin, tid, sh_ts);
  st.suppress_eot = 0;
  const int per = (p.n_vocab + BEAM_SPLIT - 1) / BEAM_SPLIT;
  const int lo = part * per, V = lo + per < p.n_vocab ? lo + per : p.n_vocab;
  // two batches of 13 loads per thread, all of a batch in flight before the first is used (the sampler's pattern): unconditional
  // loads (clamped address, masked below), the bias behind ONE test of its pointer per batch.
  float v[BEAM_PER_THREAD];
  float lmax = -INFINITY;
  constexpr int TB = BEAM_PER_THREAD / 2;
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
    float vv[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int i = lo + tid + (hb * TB + u) * BEAM_THREADS;
      vv[u] = lg[i < V ? i : V - 1];
    }
    if (p.bias) {
      float bb[TB];
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int i = lo + tid + (hb * TB + u) * BEAM_THREADS;
        bb[u] = p.bias[i < V ? i : V - 1];
      }
#pragma unroll
      for (int u = 0; u < TB; ++u) vv[u] += bb[u];
    }
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int i = lo + tid + (hb * TB + u) * BEAM_THREADS;
      const float x = (i >= V || !sp_allowed(p, st, i)) ? -INFINITY : vv[u];
      v[hb * TB + u] = x;
      lmax = fmaxf(lmax, x);
    }
  }
  auto block_max = [&](float x) {
    x = wave_max(x);
    if ((tid & 63) == 0) red_v[tid >> 6] = x;
    __syncthreads();
    float r = red_v[0];
    for (int k = 1; k < BEAM_THREADS / 64; ++k) r = fmaxf(r, red_v[k]);
    __syncthreads();
    return r;
  };
  auto block_sum = [&](float x) {
    x = wave_sum(x);
    if ((tid & 63) == 0) red_v[tid >> 6] = x;
    __syncthreads();
    float r = 0.f;
    for (int k = 0; k < BEAM_THREADS / 64; ++k) r += red_v[k];
    __syncthreads();
    return r;
  };
  const float m = block_max(lmax);
  float s_all = 0.f, s_ts = 0.f, t_max = -INFINITY;
#pragma unroll
  for (int u = 0; u < BEAM_PER_THREAD; ++u) {
    const int i = lo + tid + u * BEAM_THREADS;
    if (v[u] > -INFINITY) {
      const float e = expf(v[u] - m);
      s_all += e;
      if (i >= p.ts_begin) s_ts += e; else t_max = fmaxf(t_max, v[u]);
    }
  }
  const float sum_all = block_sum(s_all), sum_ts = block_sum(s_ts), text_max = block_max(t_max);
  // The slice's K + 1 best per kind (0: text, 1: timestamps).  Every LANE first sorts the best six of its own 26 values per kind into
  // registers (one pass, branch-free insertion; equal values keep the lower index in front); a wave's K + 1 best are then K + 1
  // rounds over the lanes' list HEADS - a shuffle tree, the winner pops its list - and wave 0 takes the K + 1 best of the waves'
  // 4 x (K + 1).  (Rounds that re-scanned the 26 values and struck the winner out took 29 of the kernel's 47 us.)
  __shared__ float sh_wv[2][BEAM_THREADS / 64][6];
  __shared__ int sh_wi[2][BEAM_THREADS / 64][6];
  const int wv = tid >> 6, ln = tid & 63;
  float lv[2][6]; int li[2][6];
#pragma unroll
  for (int kind = 0; kind < 2; ++kind)
#pragma unroll
    for (int k = 0; k < 6; ++k) { lv[kind][k] = -INFINITY; li[kind][k] = 0x7fffffff; }
#pragma unroll
  for (int u = 0; u < BEAM_PER_THREAD; ++u) {
    const int i = lo + tid + u * BEAM_THREADS;
    const bool is_ts = i >= p.ts_begin;
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
      float x = (kind == 1) == is_ts ? v[u] : -INFINITY; int xi = i;
#pragma unroll
      for (int k = 0; k < 6; ++k) {                       // strict >: an equal value with a higher index stays behind
        const bool up = x > lv[kind][k];
        const float tv = up ? lv[kind][k] : x; const int ti = up ? li[kind][k] : xi;
        lv[kind][k] = up ? x : lv[kind][k]; li[kind][k] = up ? xi : li[kind][k];
        x = tv; xi = ti;
      }
    }
  }
  for (int c = 0; c < K1; ++c) {
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
      float bv = lv[kind][0]; int bi = bv > -INFINITY ? li[kind][0] : 0x7fffffff;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      const bool got = bv > -INFINITY;
      if (got && lv[kind][0] == bv && li[kind][0] == bi) {        // the winner pops its list
#pragma unroll
        for (int k = 0; k < 5; ++k) { lv[kind][k] = lv[kind][k + 1]; li[kind][k] = li[kind][k + 1]; }
        lv[kind][5] = -INFINITY; li[kind][5] = 0x7fffffff;
      }
      if (ln == 0) { sh_wv[kind][wv][c] = bv; sh_wi[kind][wv][c] = got ? bi : -1; }
    }
  }
  __syncthreads();
  if (wv == 0) {
    for (int kind = 0; kind < 2; ++kind) {
      float cv = -INFINITY; int ci = -1;
      if (ln < (BEAM_THREADS / 64) * K1) { cv = sh_wv[kind][ln / K1][ln % K1]; ci = sh_wi[kind][ln / K1][ln % K1]; }
      for (int c = 0; c < K1; ++c) {
        float bv = ci >= 0 ? cv : -INFINITY; int bi = ci >= 0 ? ci : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        const bool got = bv > -INFINITY;
        if (ln == 0) { sh_cv[kind][c] = bv; sh_ci[kind][c] = got ? bi : -1; }
        if (got && ci == bi) { cv = -INFINITY; ci = -1; }
      }
    }
  }
  __syncthreads();
  __shared__ int sh_last;
  __shared__ unsigned sh_w[BEAM_SPLIT][28];
  if (tid == 0) {
    // publish the slice, take a ticket; only the last arriver's workgroup goes on
    unsigned* slot = bp.part + ((int64_t)row * BEAM_SPLIT + part) * BEAM_PART_WORDS;
    unsigned words[28];
    words[0] = __float_as_uint(m); words[1] = __float_as_uint(sum_all); words[2] = __float_as_uint(sum_ts); words[3] = __float_as_uint(text_max);
#pragma unroll
    for (int kind = 0; kind < 2; ++kind)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        words[4 + (kind * 6 + c) * 2] = __float_as_uint(c < K1 ? sh_cv[kind][c] : -INFINITY);
        words[5 + (kind * 6 + c) * 2] = (unsigned)(c < K1 ? sh_ci[kind][c] : -1);
      }
#pragma unroll
    for (int k = 0; k < 28; ++k) __hip_atomic_store(slot + k, words[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(bp.tickets + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_last = ticket == BEAM_SPLIT - 1;
    if (sh_last) __hip_atomic_store(bp.tickets + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!sh_last) return;
  // the last arriver: every slice's state in ONE round trip (a thread per word, agent-scope loads), then the first 96 threads
  // hold one candidate each and K + 1 rounds of arg-max run over them
  if (tid < BEAM_SPLIT * 28) {
    const unsigned* all = bp.part + (int64_t)row * BEAM_SPLIT * BEAM_PART_WORDS;
    sh_w[tid / 28][tid % 28] = __hip_atomic_load(all + (tid / 28) * BEAM_PART_WORDS + tid % 28, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int q = 0; q < BEAM_SPLIT; ++q) M = fmaxf(M, __uint_as_float(sh_w[q][0]));
  float S = 0.f, Sts = 0.f, tmax = -INFINITY;
#pragma unroll
  for (int q = 0; q < BEAM_SPLIT; ++q) {                  // slice order: every thread computes the same sums
    const float mq = __uint_as_float(sh_w[q][0]);
    const float f = mq == -INFINITY ? 0.f : expf(mq - M);
    S += __uint_as_float(sh_w[q][1]) * f;
    Sts += __uint_as_float(sh_w[q][2]) * f;
    tmax = fmaxf(tmax, __uint_as_float(sh_w[q][3]));
  }
  const float lse = M + logf(S);
  // the timestamp-mass rule: when the timestamps together outweigh every text token, only timestamps remain
  const bool force_ts = !p.no_timestamps && Sts > 0.f && (M + logf(Sts) - lse) > (tmax - lse);
  float cv = -INFINITY; int ci = -1;
  if (tid < BEAM_SPLIT * 12) {
    const int q = tid / 12, k = tid % 12;
    ci = (int)sh_w[q][5 + 2 * k];
    cv = __uint_as_float(sh_w[q][4 + 2 * k]);
    if (ci < 0 || (force_ts && k < 6)) { cv = -INFINITY; ci = -1; }
  }
  for (int c = 0; c < K1; ++c) {
    float bv = ci >= 0 ? cv : -INFINITY; int bi = ci >= 0 ? ci : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
    __syncthreads();
    float rv = red_v[0]; int ri = red_i[0];
    for (int k = 1; k < BEAM_THREADS / 64; ++k) if (red_v[k] > rv || (red_v[k] == rv && red_i[k] < ri)) { rv = red_v[k]; ri = red_i[k]; }
    __syncthreads();
    const bool got = rv > -INFINITY;
    if (tid == 0) {
      bp.cand_lp[(int64_t)row * K1 + c] = got ? rv - lse : -INFINITY;
      bp.cand_tok[(int64_t)row * K1 + c] = got ? ri : -1;
    }
    if (got && ci == ri) { cv = -INFINITY; ci = -1; }        // struck out (an index appears once: the slices are disjoint)
  }
}

// one workgroup per window: rank the candidates, retire finished sequences, reorder the beams
__global__ __launch_bounds__(256) void beam_update_kernel(SamplerParams p, BeamParams bp, int first) {
  constexpr int MAXK = 5, MAXC = MAXK * (MAXK + 1);
  __shared__ int s_src[MAXK], s_tok[MAXK], s_nfin, s_fin_src[MAXC], s_fin_slot[MAXC];
  __shared__ float s_sum[MAXK], s_fin_sum[MAXC];
  __shared__ int s_done;
  const int w = blockIdx.x, tid = threadIdx.x, K = bp.K;
  if (bp.win_done[w]) return;
  const int n_cur = bp.n_cur[w];
  const int P = bp.n_past_w[w];          // position the step that just ran wrote (prompt pass: n_prompt - 1)
  // the window's K x (K + 1) candidates in ONE memory round trip (thread per candidate); one thread reading them in its
  // ranking loop paid a dependent round trip per candidate (34 us per launch)
  __shared__ int s_ctok[MAXC];
  __shared__ float s_csc[MAXC];
  const int nb = first ? 1 : K;          // first step: the K beams are identical - only beam 0 proposes
  if (tid < nb * (K + 1)) {
    const int j = tid / (K + 1), c = tid % (K + 1), r = w * K + j;
    s_ctok[tid] = bp.cand_tok[(int64_t)r * (K + 1) + c];
    s_csc[tid] = (first ? 0.f : bp.beam_sum[r]) + bp.cand_lp[(int64_t)r * (K + 1) + c];
  }
  __syncthreads();
  // rank of a candidate = how many candidates come before it (score descending; ties: the earlier beam, then the more likely token =
  // the lower candidate index): a thread per candidate, all reads from LDS (one thread sorting private arrays took 30 us: they
  // live in scratch memory)
  __shared__ int s_ord[MAXC];
  const int n_all = nb * (K + 1);
  if (tid < MAXC) s_ord[tid] = -1;
  __syncthreads();
  if (tid < n_all && s_ctok[tid] >= 0) {
    int rank = 0;
    const float mine = s_csc[tid];
    for (int o = 0; o < n_all; ++o)
      if (s_ctok[o] >= 0 && (s_csc[o] > mine || (s_csc[o] == mine && o < tid))) ++rank;
    s_ord[rank] = tid;
  }
  __syncthreads();
  if (tid == 0) {
    int saved = 0, nfin = 0, fin_cnt = bp.fin_cnt[w];
    for (int a = 0; a < n_all && saved < K; ++a) {
      const int o = s_ord[a];
      if (o < 0) break;                    // ranks are dense: the first gap ends the list
      const int src_o = o / (K + 1);
      if (s_ctok[o] == p.eot) {
        // newly finished, best first; the pool takes them while it has room (max_candidates = beam size)
        if (fin_cnt < K) { s_fin_src[nfin] = src_o; s_fin_sum[nfin] = s_csc[o]; s_fin_slot[nfin] = fin_cnt; ++nfin; ++fin_cnt; }
      } else {
        s_src[saved] = src_o; s_tok[saved] = s_ctok[o]; s_sum[saved] = s_csc[o]; ++saved;
      }
    }
    // fewer live continuations than beams (everything else was end-of-text or forbidden): repeat the last one
    for (int j = saved; j < K; ++j) { s_src[j] = saved ? s_src[saved - 1] : 0; s_tok[j] = saved ? s_tok[saved - 1] : p.eot; s_sum[j] = saved ? -INFINITY : -INFINITY; }
    s_nfin = nfin;
    bp.fin_cnt[w] = fin_cnt;
    const bool full = fin_cnt >= K;
    const bool out_of_room = n_cur + 1 >= p.n_max || n_cur + 1 >= p.max_tokens || P + 2 >= p.n_text_ctx || saved == 0;
    s_done = (full || out_of_room) ? 1 : 0;
  }
  __syncthreads();
  // finished sequences: the source beam's tokens (end-of-text itself is not stored); one flat index over (sequence, position)
  // so that every copy of the launch is ONE round trip (a loop over the sequences was a dependent round trip each)
  for (int idx = tid; idx < s_nfin * n_cur; idx += 256) {
    const int f = idx / n_cur, i = idx - f * n_cur;
    bp.fin_tok[((int64_t)w * K + s_fin_slot[f]) * p.max_tokens + i] = p.tokens[(int64_t)(w * K + s_fin_src[f]) * p.max_tokens + i];
  }
  if (tid < s_nfin) { bp.fin_len[w * K + s_fin_slot[tid]] = n_cur; bp.fin_sum[w * K + s_fin_slot[tid]] = s_fin_sum[tid]; }
  // new beams: history and kv_slot rows are gathered from the source beams into the other half of the double buffers
  for (int idx = tid; idx < K * n_cur; idx += 256) {
    const int j = idx / n_cur, i = idx - j * n_cur;
    bp.tokens_next[(int64_t)(w * K + j) * p.max_tokens + i] = p.tokens[(int64_t)(w * K + s_src[j]) * p.max_tokens + i];
  }
  for (int idx = tid; idx < K * (P + 1); idx += 256) {
    const int j = idx / (P + 1), i = idx - j * (P + 1);
    bp.kv_slot_next[(int64_t)(w * K + j) * p.n_text_ctx + i] = first ? w : bp.kv_slot[(int64_t)(w * K + s_src[j]) * p.n_text_ctx + i];   // after the prompt pass every position lives in slot w
  }
  if (tid < K) {
    const int j = tid, r = w * K + j;
    bp.tokens_next[(int64_t)r * p.max_tokens + n_cur] = s_tok[j];
    bp.kv_slot_next[(int64_t)r * p.n_text_ctx + P + 1] = r;      // the next step writes position P + 1 of this beam into its own row
    bp.beam_sum[r] = s_sum[j];
    p.next_tok[r] = s_tok[j];
    p.n_past[r] = P + 1;
  }
  if (tid == 0) {
    bp.n_cur[w] = n_cur + 1;
    bp.n_past_w[w] = P + 1;
    if (s_done) { bp.win_done[w] = 1; atomicAdd(p.n_done, 1); }
  }
}

void launch_beam_step(const SamplerParams& p, const BeamParams& bp, int n_windows, int first, hipStream_t s) {
  if (bp.K < 2 || bp.K > 5) throw Error(OHW_E_INVALID_ARG, "beam search: beam size must be in 2..5");
  if (p.n_vocab > BEAM_SPLIT * BEAM_THREADS * BEAM_PER_THREAD) throw Error(OHW_E_INVALID_ARG, "beam search: vocabulary too large");
  if (!bp.part || !bp.tickets) throw Error(OHW_E_INVALID_ARG, "beam search: the split top-k needs its slice buffers");
  hipLaunchKernelGGL(beam_topk_kernel, dim3(BEAM_SPLIT, first ? n_windows : n_windows * bp.K), dim3(BEAM_THREADS), 0, s, p, bp, first);
  hipLaunchKernelGGL(beam_update_kernel, dim3(n_windows), dim3(256), 0, s, p, bp, first);
  HIP_CHECK(hipGetLastError());
}

#define INST(T) \
  template void launch_dec_gemm<T>(const DecGemmParams&, int, hipStream_t); \
  template void launch_embed<T>(const void*, const float*, const int32_t*, const int32_t*, float*, void*, float*, int, int, int, hipStream_t); \
  template void launch_self_attn<T>(const void*, const void*, const void*, const int32_t*, void*, int, int, int, int, hipStream_t, const int32_t*); \
  template void launch_cross_attn<T>(const void*, const void*, const void*, void*, int, int, int, int, float*, unsigned*, int, const int32_t*, hipStream_t, int, bool);
INST(bf16_t)
INST(f16_t)
#undef INST

}  // namespace ohw

#ifdef OHW_TRACE
// instrumented build only: copy the timeline out and reset it (returns the number of records)
extern "C" int ohw_dbg_trace_read(unsigned long long* out, int cap) {
  unsigned n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(ohw::ohw_trace_n), sizeof(n)) != hipSuccess) return -1;
  if (n > (1u << 18)) n = 1u << 18;
  if ((int)n > cap) n = cap;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(ohw::ohw_trace_buf), (size_t)n * 8) != hipSuccess) return -1;
  const unsigned zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(ohw::ohw_trace_n), &zero, sizeof(zero)) != hipSuccess) return -1;
  return (int)n;
}
#endif
