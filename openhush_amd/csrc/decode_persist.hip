// decode_persist.hip — ONE launch for the 32 decoder layers of a single-token step at small batch (at most 16 rows:
// the daemon's one utterance, BASELINE config #2, the beam rows of config #5), gfx950.
//
// Why: at one row the step is a chain of 8 dependent kernels per layer - 260 launches of 3 - 5 us with 1.5 - 1.8 us between
// them (tools/dec_trace.py: 1.44 ms per token at large-v3 against 0.23 ms of weight bytes).  A dependent hop between
// workgroups of ONE launch costs 1.6 - 2.2 us on this chip when the payload is its own flag (tools/probes/persist_probe.hip,
// profiles/r03_persist_probe.txt), and the weights of the next GEMM can be in flight while the hop is waited for.
//
// Shape: one workgroup per CU (all resident: grid <= CUs, 96 KiB of LDS each), 8 waves.
//   waves 0-3 ("MFMA waves")  stream a task's weight fragments straight into registers BEFORE the task's inputs exist, run
//                             the 16x16x32 MFMAs against the activation tile in LDS, leave 4 partial tiles in LDS; in the
//                             cross-attention phase they stream the K/V slice instead
//   waves 4-7 ("IO waves")    gather a task's inputs from the previous phase's granules (LayerNorm fused here), write the LDS
//                             tile, reduce the partial tiles in fixed order, apply the epilogue and publish; they also run
//                             the per-(row, head) phases (self-attention, cross-attention merge, mlp.2 reduce), one wave per task
// Ten phases per layer: A LN1+QKV | B self-attention | C out-projection (+x) | D LNx + cross query | E cross-attention partials
// over key slices | F merge of the slices | G cross out-projection (+x) | H LN2 + mlp.0 + GELU | I mlp.2 over K slices |
// J x += bias + the K-slice partials.  Phase p's tasks are dealt round-robin to the workgroups; nothing is waited for but data.
//
// Hand-off form (cdna_hip_programming.md Guideline 16, R2; MI355X_MICROARCH.md "Valid forms"): every value that crosses
// workgroups is an 8-byte granule {32-bit payload | 32-bit tag} written by ONE sc1 store and polled with sc1 loads until the
// tag matches; tag = epoch * 1024 + phase + 1, where the epoch is a device word this kernel bumps when it completes - so a
// granule left by the previous launch (same phase number) never matches, nothing is zeroed between launches and the launch
// replays from a hipGraph unchanged.  Every spin is bounded; a workgroup that gives up sets abort_word and the launch drains.
// Reductions have a fixed order (K-slices of the 4 MFMA waves, key slices, mlp.2 slices): a row's result does not depend on
// the other rows of the batch.
#include <type_traits>

#include "kernels.hpp"

namespace ohw {

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef unsigned long long u64;

constexpr int PS_THREADS = 512;
constexpr int PS_NB = 14;                  // weight k-blocks an MFMA wave holds in registers: a task's K slice <= 4 * 14 * 32
constexpr int PS_TROW = 56 * 64 + 16;      // bytes per row of the LDS activation tile (<= 56 k-blocks, + 16 against bank conflicts)
constexpr int PS_NQ = 5;                   // rows that share one window's cross K/V (beam size), at most
constexpr int PS_SPIN = 1 << 19;           // polls before a waiter gives up (about half a second)
constexpr int PS_LDS = 96 * 1024;          // more than half a CU's LDS: one workgroup per CU

// LDS carve-up (bytes)
constexpr int L_TILE = 0;                                  // [16][PS_TROW]
constexpr int L_PART = L_TILE + 16 * PS_TROW;              // f32x4 [4 waves][64]
constexpr int L_QS = L_PART + 4 * 64 * 16;                 // f32 [PS_NQ][64]   cross-attention queries (scaled)
constexpr int L_RM = L_QS + PS_NQ * 64 * 4;                // f32 [PS_NQ][4]
constexpr int L_RL = L_RM + PS_NQ * 4 * 4;                 // f32 [PS_NQ][4]
constexpr int L_RO = L_RL + PS_NQ * 4 * 4;                 // f32 [PS_NQ][4][64]
constexpr int L_FAIL = L_RO + PS_NQ * 4 * 64 * 4;          // int
constexpr int L_END = L_FAIL + 16;
static_assert(L_END <= PS_LDS, "LDS budget");

__device__ __forceinline__ void gst(u64* p, unsigned tag, unsigned data) {
  __hip_atomic_store((gu64*)p, ((u64)tag << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gld(const u64* p) { return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// granule `idx` of the arena: a zero-extended 32-bit byte offset from the (uniform) base - one address register per load
__device__ __forceinline__ const u64* gat(const u64* base, int idx) { return (const u64*)((const char*)base + (size_t)((unsigned)idx * 8u)); }

// an opaque copy: index arithmetic built on it cannot be hoisted out of the phase loops (LICM otherwise parks a hundred
// loop-invariant granule offsets in registers for the whole launch and the kernel spills)
__device__ __forceinline__ int opq(int x) { asm volatile("" : "+v"(x)); return x; }

struct Ctl {
  volatile int* fail;          // LDS: this workgroup has given up (or seen another give up)
  unsigned* abort_word;        // global
  unsigned code;               // phase + 1 of the current wait (diagnostics)
};

// poll `n` (<= N) granules g[idx[i]] until every tag matches; data out.  Per lane, no wave-uniform requirement.
template <int N>
__device__ __forceinline__ bool poll(const u64* g, const int (&idx)[N], int n, unsigned tag, unsigned (&out)[N], const Ctl& c) {
  if (*c.fail) return false;
  for (int spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i < n) {
        const u64 v = gld(gat(g, idx[i]));
        out[i] = (unsigned)v;
        ok &= (unsigned)(v >> 32) == tag;
      }
    }
    if (ok) return true;
    if (spins > PS_SPIN || ((spins & 63) == 63 && __hip_atomic_load((__attribute__((address_space(1))) unsigned*)c.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      if (spins > PS_SPIN) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)c.abort_word, c.code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *c.fail = 1;
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

#ifdef OHW_TRACE
// instrumented build only (tools/persist_trace.py): 100 MHz stamps of one layer, per workgroup and phase.  Slots 0-3 by the first
// IO wave (phase entered, inputs gathered, partial tiles ready, published), 4-7 by the first MFMA wave (weights requested,
// inputs ready, weights landed, products done); the first task of a phase only
__device__ unsigned long long ps_trace_buf[256 * 10 * 8];
#define PTRACE(cond, ph, k) do { if ((cond) && (tid & 255) == 0 && wg < 256) ps_trace_buf[(wg * 10 + (ph)) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PTRACE(cond, ph, k) do { } while (0)
#endif

}  // namespace

int64_t persist_layout(PersistParams* p) {
  const int64_t d = p->d, H = p->H;
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += (n + 15) / 16 * 16; return (int32_t)at; };   // regions start on 128-byte lines
  p->o_x = take(16 * d);
  p->o_q = take(16 * d / 2);
  p->o_kv = take(16 * 2 * d / 2);
  p->o_a = take(16 * d / 2);
  p->o_qx = take(16 * d / 2);
  p->o_h = take(16 * 2 * d);
  p->o_xp = take(16 * H * 16 * 66);
  p->o_mp = take(4 * 16 * d);
  return o;
}

template <typename T>
__global__ __launch_bounds__(PS_THREADS, 2) void persist_step_kernel(PersistParams p) {
  using Ops = TypeOps<T>;
  using vec8 = typename Ops::vec8;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* tile = lds + L_TILE;
  f32x4* part = (f32x4*)(lds + L_PART);
  float* qs = (float*)(lds + L_QS);
  float* red_m = (float*)(lds + L_RM);
  float* red_l = (float*)(lds + L_RL);
  float* red_o = (float*)(lds + L_RO);
  volatile int* fail = (volatile int*)(lds + L_FAIL);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool io = wave >= 4;
  const int iw = wave & 3;                 // index inside the wave's group
  const int it = tid & 255;                // thread index inside the group
  const int wg = blockIdx.x, G = gridDim.x;
  const int M = p.M, d = p.d, H = p.H, W = p.M / p.group;
  const int nt_d = d / 16;

  if (tid == 0) *fail = 0;
  // rows >= M of the activation tile stay zero for the whole launch (they feed MFMA columns that are never stored)
  for (int i = tid; i < 16 * PS_TROW / 16; i += PS_THREADS) ((u32x4*)tile)[i] = (u32x4){0u, 0u, 0u, 0u};
  const unsigned epoch = *p.epoch & 0x3fffffu;
  __syncthreads();
  auto tag_of = [&](int layer, int k) { return (epoch << 10) + (unsigned)(layer * 10 + k + 1); };

  // the GEMM phases: kind 0: A (LN1 + QKV)  1: C (out-proj, +x)  2: D (LNx + xq)  3: G (xo, +x)  4: H (LN2 + mlp.0 + GELU)  5: I (mlp.2 K slice)
  // phase index k of a kind inside its layer (the tag it publishes under): A 0, C 2, D 3, G 6, H 7, I 8
  struct Desc { const void* w; const float* bias; int n_tiles, KB, nsl; };
  auto desc_of = [&](const PersistLayer& lw, int kind) {
    Desc q;
    switch (kind) {
      case 0: q = Desc{lw.wqkv, lw.bqkv, 3 * nt_d, d / 32, 1}; break;
      case 1: q = Desc{lw.wo, lw.bo, nt_d, d / 32, 1}; break;
      case 2: q = Desc{lw.wxq, lw.bxq, nt_d, d / 32, 1}; break;
      case 3: q = Desc{lw.wxo, lw.bxo, nt_d, d / 32, 1}; break;
      case 4: q = Desc{lw.w1, lw.b1, 4 * nt_d, d / 32, 1}; break;
      default: q = Desc{lw.w2, nullptr, nt_d, 4 * d / 32, p.nsplit}; break;
    }
    return q;
  };
  // cross-attention key slices
  const int S = p.S, NQ = p.group;
  const int n_groups_all = (p.t_len + 7) / 8;
  const int per_slice = (n_groups_all + S - 1) / S;
  const int part8 = lane & 7, slot = lane >> 3;

  if (!io) {
    // =====================================================================================================================
    // MFMA waves: weights (or the K/V slice) are requested before the barrier that says the inputs are in LDS
    // =====================================================================================================================
    vec8 wreg[PS_NB];
    for (int layer = 0; layer < p.L; ++layer) {
      const PersistLayer lw = p.layers[layer];
      const T* xk = (const T*)p.xkv + (int64_t)(2 * layer) * p.xkv_slab;
      const T* xv = xk + p.xkv_slab;
#pragma unroll 1
      for (int ph = 0; ph < 7; ++ph) {
        if (ph != 3) {
          const int kind = ph < 3 ? ph : ph - 1;                     // A C D | E | G H I
          const Desc q = desc_of(lw, kind);
          const int n_tasks = q.n_tiles * q.nsl;
          const int lph = ph == 0 ? 0 : ph == 1 ? 2 : ph == 2 ? 3 : ph == 4 ? 6 : ph == 5 ? 7 : 8;
          (void)lph;
          for (int t = wg; t < n_tasks; t += G) {
            const int nt = t / q.nsl, sl = t % q.nsl;
            const int kb0 = (int)((int64_t)q.KB * sl / q.nsl), kb1 = (int)((int64_t)q.KB * (sl + 1) / q.nsl);
            const vec8* wt = (const vec8*)q.w + ((int64_t)nt * q.KB) * 64 + lane;
#pragma unroll
            for (int u = 0; u < PS_NB; ++u) {
              int kk = kb0 + iw + 4 * u;
              if (kk > kb1 - 1) kk = kb1 - 1;                    // clamped: unconditional loads, masked below
              wreg[u] = __builtin_nontemporal_load(&wt[(int64_t)kk * 64]);
            }
            PTRACE(layer == p.L / 2 && t == wg, lph, 4);
            __syncthreads();                                      // #1: the tile is complete
            PTRACE(layer == p.L / 2 && t == wg, lph, 5);
#ifdef OHW_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PTRACE(layer == p.L / 2 && t == wg, lph, 6);
#endif
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const unsigned char* y0 = tile + (lane & 15) * PS_TROW + (lane >> 4) * 16;
#pragma unroll
            for (int u = 0; u < PS_NB; ++u) {
              const int kk = kb0 + iw + 4 * u;
              const int kc = kk < kb1 ? kk - kb0 : 0;
              vec8 a = *(const vec8*)(y0 + kc * 64);
              if (kk >= kb1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = 0;
              }
              acc = Ops::mfma16(wreg[u], a, acc);
            }
            part[iw * 64 + lane] = acc;
            PTRACE(layer == p.L / 2 && t == wg, lph, 7);
            __syncthreads();                                      // #2: the four partial tiles are in LDS
          }
        } else {
          // ---- E: cross-attention partials of (window, head, key slice) for the window's NQ rows
          const int n_tasks = W * H * S;
          for (int t = wg; t < n_tasks; t += G) {
            const int w = t / (H * S), h = (t / S) % H, sl = t % S;
            const bool skip = p.done && p.done[w];
            const int g_lo = sl * per_slice;
            const int g_hi = g_lo + per_slice < n_groups_all ? g_lo + per_slice : n_groups_all;
            constexpr int XU = 4;
            vec8 kf[XU], vf[XU];
            int keys[XU];
            const T* kb = xk + (((int64_t)w * H + h) * p.t_len << 6) + part8 * 8;
            const T* vb = xv + (((int64_t)w * H + h) * p.t_len << 6) + part8 * 8;
            auto load_kv = [&](int g0) {
#pragma unroll
              for (int u = 0; u < XU; ++u) {
                const int g = g0 + 4 * u;
                int key = g * 8 + slot;
                keys[u] = (g < g_hi && key < p.t_len) ? key : -1;
                if (key > p.t_len - 1) key = p.t_len - 1;
                kf[u] = __builtin_nontemporal_load((const vec8*)(kb + ((int64_t)key << 6)));
                vf[u] = __builtin_nontemporal_load((const vec8*)(vb + ((int64_t)key << 6)));
              }
            };
            if (!skip) load_kv(g_lo + iw);                       // in flight before the queries exist
            PTRACE(layer == p.L / 2 && t == wg, 4, 4);
            __syncthreads();                                      // #1: the queries are in LDS
            PTRACE(layer == p.L / 2 && t == wg, 4, 5);
            if (!skip) {
              float qv[PS_NQ][8], mrun[PS_NQ], lrun[PS_NQ], acc[PS_NQ][8];
#pragma unroll
              for (int i = 0; i < PS_NQ; ++i) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { qv[i][e] = i < NQ ? qs[i * 64 + part8 * 8 + e] : 0.f; acc[i][e] = 0.f; }
                mrun[i] = -INFINITY; lrun[i] = 0.f;
              }
              for (int g0 = g_lo + iw; g0 < g_hi; g0 += 4 * XU) {
                if (g0 != g_lo + iw) load_kv(g0);
#pragma unroll
                for (int u = 0; u < XU; ++u) {
#pragma unroll
                  for (int i = 0; i < PS_NQ; ++i) {
                    if (i < NQ) {
                      float sc = 0.f;
#pragma unroll
                      for (int e = 0; e < 8; ++e) sc += qv[i][e] * (float)kf[u][e];
                      sc += __shfl_xor(sc, 1, 64);
                      sc += __shfl_xor(sc, 2, 64);
                      sc += __shfl_xor(sc, 4, 64);
                      if (keys[u] < 0) sc = -INFINITY;
                      const float mn = fmaxf(mrun[i], sc);
                      const float alpha = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(mrun[i] - mn);
                      const float pe = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sc - mn);
                      mrun[i] = mn;
                      lrun[i] = lrun[i] * alpha + pe;
#pragma unroll
                      for (int e = 0; e < 8; ++e) acc[i][e] = acc[i][e] * alpha + pe * (float)vf[u][e];
                    }
                  }
                }
              }
#pragma unroll
              for (int i = 0; i < PS_NQ; ++i) {
                if (i < NQ) {
#pragma unroll
                  for (int o2 = 8; o2 <= 32; o2 <<= 1) {
                    const float m2 = __shfl_xor(mrun[i], o2, 64), l2 = __shfl_xor(lrun[i], o2, 64);
                    const float mn = fmaxf(mrun[i], m2);
                    const float a1 = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(mrun[i] - mn);
                    const float a2 = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m2 - mn);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                      const float ox = __shfl_xor(acc[i][e], o2, 64);
                      acc[i][e] = acc[i][e] * a1 + ox * a2;
                    }
                    lrun[i] = lrun[i] * a1 + l2 * a2;
                    mrun[i] = mn;
                  }
                  if (slot == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) red_o[(i * 4 + iw) * 64 + part8 * 8 + e] = acc[i][e];
                    if (part8 == 0) { red_m[i * 4 + iw] = mrun[i]; red_l[i * 4 + iw] = lrun[i]; }
                  }
                }
              }
            }
            PTRACE(layer == p.L / 2 && t == wg, 4, 7);
            __syncthreads();                                      // #2: the four waves' states are in LDS
          }
        }
      }
    }
  } else {
    // =====================================================================================================================
    // IO waves
    // =====================================================================================================================
    Ctl ctl{fail, p.abort_word, 0u};
    // row r's LayerNorm input from the residual granules (or, layer 0, the plain embedding output) -> (x - mean) * rstd as 16-bit
    // into tile row r.  One wave per row; lane l owns columns l, l + 64, ... (d <= 1280: 20 per lane).
    auto ln_row = [&](int r, unsigned tag, bool plain) -> bool {
      constexpr int NV = 20;
      float v[NV];
      const int nv = d / 64;
      const int lane = opq(tid & 63);
      if (plain) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = i < nv ? p.x_in[(int64_t)r * d + lane + 64 * i] : 0.f;
      } else {
        int idx[NV]; unsigned raw[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) idx[i] = p.o_x + r * d + lane + 64 * (i < nv ? i : 0);
        if (!poll<NV>(p.g, idx, nv, tag, raw, ctl)) return false;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = i < nv ? __uint_as_float(raw[i]) : 0.f;
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) s += v[i];
      const float mean = wave_sum(s) / (float)d;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) { const float c = i < nv ? v[i] - mean : 0.f; q += c * c; }
      const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (i < nv) *(T*)(tile + r * PS_TROW + (lane + 64 * i) * 2) = (T)((v[i] - mean) * rstd);
      return true;
    };
    // row r of a 16-bit granule region (two values per granule), columns [c0, c1) (even bounds) -> tile row r
    auto gather_row16 = [&](int region, int row_granules, int r, int c0, int c1, unsigned tag) -> bool {
      constexpr int NV = 14;
      const int n = (c1 - c0) / 2;
      const int lane = opq(tid & 63);
      int idx[NV]; unsigned raw[NV];
      int nv = 0;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        idx[i] = region + r * row_granules + c0 / 2 + (j < n ? j : 0);
        if (j < n) nv = i + 1;
      }
      if (!poll<NV>(p.g, idx, nv, tag, raw, ctl)) return false;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < n) *(unsigned*)(tile + r * PS_TROW + j * 4) = raw[i];
      }
      return true;
    };

    for (int layer = 0; layer < p.L; ++layer) {
      const PersistLayer lw = p.layers[layer];
      T* kc = (T*)p.self_kv + (int64_t)(2 * layer) * p.kv_layer;
      T* vc = kc + p.kv_layer;
      const unsigned tA = tag_of(layer, 0), tB = tag_of(layer, 1), tC = tag_of(layer, 2), tD = tag_of(layer, 3), tE = tag_of(layer, 4),
                     tF = tag_of(layer, 5), tG = tag_of(layer, 6), tH = tag_of(layer, 7), tI = tag_of(layer, 8), tJ = tag_of(layer, 9);
      const unsigned tXin = layer == 0 ? 0u : tag_of(layer - 1, 9);     // the residual stream as the previous layer left it

#pragma unroll 1
      for (int ph = 0; ph < 10; ++ph) {
        ctl.code = (unsigned)(layer * 10 + ph + 1);
        if (ph == 0 || ph == 2 || ph == 3 || ph == 6 || ph == 7 || ph == 8) {
          // ---------------- a GEMM phase: gather (+ LayerNorm) -> tile | #1 | #2 | reduce + epilogue + publish ----------------
          const int kind = ph == 0 ? 0 : ph == 2 ? 1 : ph == 3 ? 2 : ph == 6 ? 3 : ph == 7 ? 4 : 5;
          const Desc q = desc_of(lw, kind);
          const unsigned t_in = kind == 0 ? tXin : kind == 1 ? tB : kind == 2 ? tC : kind == 3 ? tF : kind == 4 ? tG : tH;
          const unsigned t_out = kind == 0 ? tA : kind == 1 ? tC : kind == 2 ? tD : kind == 3 ? tG : kind == 4 ? tH : tI;
          const int n_tasks = q.n_tiles * q.nsl;
          for (int t = wg; t < n_tasks; t += G) {
            const int nt = t / q.nsl, sl = t % q.nsl;
            const int kb0 = (int)((int64_t)q.KB * sl / q.nsl), kb1 = (int)((int64_t)q.KB * (sl + 1) / q.nsl);
            PTRACE(layer == p.L / 2 && t == wg, ph, 0);
            for (int r = iw; r < M; r += 4) {
              bool ok;
              if (kind == 0 || kind == 2 || kind == 4) ok = ln_row(r, t_in, kind == 0 && layer == 0);
              else if (kind == 1 || kind == 3) ok = gather_row16(p.o_a, d / 2, r, 0, d, t_in);
              else ok = gather_row16(p.o_h, 2 * d, r, kb0 * 32, kb1 * 32, t_in);
              if (!ok) break;
            }
            PTRACE(layer == p.L / 2 && t == wg, ph, 1);
            __syncthreads();                                  // #1
            __syncthreads();                                  // #2
            PTRACE(layer == p.L / 2 && t == wg, ph, 2);
            // thread = (lane' = it >> 2, reg = it & 3): D[n = 4 * (lane' >> 4) + reg][m = lane' & 15]
            const int ll = opq(it >> 2), reg = opq(it & 3);
            const float* pp = (const float*)part;
            float v = pp[(0 * 64 + ll) * 4 + reg];
#pragma unroll
            for (int w4 = 1; w4 < 4; ++w4) v += pp[(w4 * 64 + ll) * 4 + reg];
            const int m = ll & 15, n = nt * 16 + 4 * (ll >> 4) + reg;
            if (q.bias) v += q.bias[n];
            if (kind == 4) v = gelu_erf(v);
            const float vn = __shfl_down(v, 1, 64);          // the next column's value (for 16-bit pairs)
            const bool live = m < M && !*fail;
            if (kind == 0) {
              // q -> its granules; k, v -> granules (this step's self-attention) and the cache row (the steps to come)
              if (live) {
                if (n < d) {
                  if ((reg & 1) == 0) gst(p.g + p.o_q + m * (d / 2) + n / 2, t_out, pack2<T>(v, vn));
                } else {
                  const int which = n < 2 * d ? 0 : 1, nn = n - (which + 1) * d;
                  if ((reg & 1) == 0) gst(p.g + p.o_kv + (m * 2 + which) * (d / 2) + nn / 2, t_out, pack2<T>(v, vn));
                  const int pos = p.n_past[m];
                  if (pos < p.n_ctx) (which ? vc : kc)[(int64_t)m * p.kv_row + (((int64_t)(nn >> 6) * p.n_ctx + pos) << 6) + (nn & 63)] = (T)v;
                }
                // layer 0: the residual stream enters the granule form here (the columns of the first d / 16 tasks)
                if (layer == 0 && nt < nt_d) gst(p.g + p.o_x + m * d + n, t_out, __float_as_uint(p.x_in[(int64_t)m * d + n]));
              }
            } else if (kind == 1 || kind == 3) {
              // x[m][n] += v: the element's one owner rewrites its granule under this phase's tag
              const unsigned t_x = kind == 1 ? (layer == 0 ? tA : tXin) : tC;
              int idx[1] = {p.o_x + (m < M ? m * d + n : 0)};
              unsigned raw[1];
              if (live && poll<1>(p.g, idx, 1, t_x, raw, ctl)) gst(p.g + idx[0], t_out, __float_as_uint(__uint_as_float(raw[0]) + v));
            } else if (kind == 2) {
              if (live && (reg & 1) == 0) gst(p.g + p.o_qx + m * (d / 2) + n / 2, t_out, pack2<T>(v, vn));
            } else if (kind == 4) {
              if (live && (reg & 1) == 0) gst(p.g + p.o_h + m * (2 * d) + n / 2, t_out, pack2<T>(v, vn));
            } else {
              if (live) gst(p.g + p.o_mp + (sl * 16 + m) * d + n, t_out, __float_as_uint(v));
            }
            PTRACE(layer == p.L / 2 && t == wg, ph, 3);
          }
        } else if (ph == 1) {
          // ---------------- B: self-attention, one IO wave per (row, head) ----------------
          // Layout of the cross-attention kernels: 8 lanes share a key (16 B of its 128-byte K / V row each), a wave-instruction
          // covers 8 keys, every 8-lane group keeps its own online-softmax state, merged at the end.  The cached positions come
          // from the K/V cache (through kv_slot for beams); this step's own key arrives as granules.
          const int n_tasks = M * H;
          for (int t = wg + G * iw; t < n_tasks; t += 4 * G) {
            const int m = t / H, h = t % H;
            const int part8 = opq(tid & 7), slot = opq((tid & 63) >> 3);
            PTRACE(layer == p.L / 2 && t == wg, 1, 0);
            int n_old = p.n_past[m];                                         // cached positions 0 .. n_old - 1
            if (n_old > p.n_ctx - 1) n_old = p.n_ctx - 1;
            const char* kb = (const char*)(kc + ((int64_t)h * p.n_ctx << 6) + part8 * 8);
            const char* vb = (const char*)(vc + ((int64_t)h * p.n_ctx << 6) + part8 * 8);
            const unsigned row_bytes = (unsigned)(p.kv_row * 2);           // rows x row bytes < 4 GiB per layer (checked at launch)
            const int32_t* slots = p.kv_slot ? p.kv_slot + (int64_t)m * p.n_ctx : nullptr;
            constexpr int SU = 4;                                            // 8 keys x 4 instructions = a chunk of 32 keys (registers: the first chunk is held across the wait)
            vec8 kf[SU], vf[SU];
            int keys[SU];
            auto load_chunk = [&](int c0) {
#pragma unroll
              for (int u = 0; u < SU; ++u) {
                int key = c0 + 8 * u + slot;
                keys[u] = key < n_old ? key : -1;
                if (key > n_old - 1) key = n_old > 0 ? n_old - 1 : 0;
                const int row = slots ? slots[key] : m;
                const unsigned off = (unsigned)row * row_bytes + ((unsigned)key << 7);
                kf[u] = *(const vec8*)(kb + (size_t)off);
                vf[u] = *(const vec8*)(vb + (size_t)off);
              }
            };
            if (n_old > 0) load_chunk(0);                                    // the cache does not depend on this step: in flight first
            // this step's q, k, v of the head: 4 granules of each per lane (dims part8 * 8 .. + 8), one round trip for all twelve
            int i12[12]; unsigned r12[12];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              i12[e] = p.o_q + m * (d / 2) + h * 32 + part8 * 4 + e;
              i12[4 + e] = p.o_kv + (m * 2 + 0) * (d / 2) + h * 32 + part8 * 4 + e;
              i12[8 + e] = p.o_kv + (m * 2 + 1) * (d / 2) + h * 32 + part8 * 4 + e;
            }
            if (!poll<12>(p.g, i12, 12, tA, r12, ctl)) break;
            PTRACE(layer == p.L / 2 && t == wg, 1, 1);
            float qv[8], kn[8], vn8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              unpack2<T>(r12[e], qv[2 * e], qv[2 * e + 1]);
              unpack2<T>(r12[4 + e], kn[2 * e], kn[2 * e + 1]);
              unpack2<T>(r12[8 + e], vn8[2 * e], vn8[2 * e + 1]);
            }
            const float scl = 0.125f * 1.44269504088896340736f;             // 1 / sqrt(64), exp2 domain
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] *= scl;
            float mrun = -INFINITY, lrun = 0.f, acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int c0 = 0; c0 < n_old; c0 += 8 * SU) {
              if (c0) load_chunk(c0);
#pragma unroll
              for (int u = 0; u < SU; ++u) {
                float sc = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sc += qv[e] * (float)kf[u][e];
                sc += __shfl_xor(sc, 1, 64);
                sc += __shfl_xor(sc, 2, 64);
                sc += __shfl_xor(sc, 4, 64);
                if (keys[u] < 0) sc = -INFINITY;
                const float mn = fmaxf(mrun, sc);
                const float alpha = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(mrun - mn);
                const float pe = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sc - mn);
                mrun = mn;
                lrun = lrun * alpha + pe;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = acc[e] * alpha + pe * (float)vf[u][e];
              }
            }
            // merge the 8 key groups of the wave (lanes with equal part8): xor 8, 16, 32
#pragma unroll
            for (int o2 = 8; o2 <= 32; o2 <<= 1) {
              const float m2 = __shfl_xor(mrun, o2, 64), l2 = __shfl_xor(lrun, o2, 64);
              const float mn = fmaxf(mrun, m2);
              const float a1 = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(mrun - mn);
              const float a2 = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m2 - mn);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float ox = __shfl_xor(acc[e], o2, 64);
                acc[e] = acc[e] * a1 + ox * a2;
              }
              lrun = lrun * a1 + l2 * a2;
              mrun = mn;
            }
            {  // this step's own key, last
              float sn = 0.f;
#pragma unroll
              for (int e = 0; e < 8; ++e) sn += qv[e] * kn[e];
              sn += __shfl_xor(sn, 1, 64);
              sn += __shfl_xor(sn, 2, 64);
              sn += __shfl_xor(sn, 4, 64);
              const float mn = fmaxf(mrun, sn);
              const float alpha = mrun == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mrun - mn);
              const float pe = __builtin_amdgcn_exp2f(sn - mn);
              lrun = lrun * alpha + pe;
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[e] = acc[e] * alpha + pe * vn8[e];
            }
            // lanes 0..7 (slot 0) hold dims part8 * 8 .. + 8: four granules each
            if (slot == 0 && !*fail) {
              const float inv = 1.0f / lrun;
#pragma unroll
              for (int e = 0; e < 4; ++e) gst(p.g + p.o_a + m * (d / 2) + h * 32 + part8 * 4 + e, tB, pack2<T>(acc[2 * e] * inv, acc[2 * e + 1] * inv));
            }
            PTRACE(layer == p.L / 2 && t == wg, 1, 3);
          }
        } else if (ph == 4) {
          // ---------------- E: queries in, partial states out (the MFMA waves stream the K/V slice) ----------------
          const int n_tasks = W * H * S;
          for (int t = wg; t < n_tasks; t += G) {
            const int w = t / (H * S), h = (t / S) % H, sl = t % S;
            const bool skip = p.done && p.done[w];
            PTRACE(layer == p.L / 2 && t == wg, 4, 0);
            if (!skip) {
              // IO wave i takes rows i, i + 4 of the window: 32 granules per row -> 64 scaled floats in LDS
              for (int i = iw; i < NQ; i += 4) {
                const int m = w * NQ + i;
                int i1[1] = {p.o_qx + m * (d / 2) + h * 32 + opq(lane & 31)}; unsigned r1[1];
                if (!poll<1>(p.g, i1, 1, tD, r1, ctl)) break;
                float lo, hi;
                unpack2<T>(r1[0], lo, hi);
                const float scl = 0.125f * 1.44269504088896340736f;
                if (lane < 32) { qs[i * 64 + 2 * lane] = lo * scl; qs[i * 64 + 2 * lane + 1] = hi * scl; }
              }
            }
            PTRACE(layer == p.L / 2 && t == wg, 4, 1);
            __syncthreads();                                  // #1
            __syncthreads();                                  // #2
            PTRACE(layer == p.L / 2 && t == wg, 4, 2);
            if (!skip) {
              for (int i = iw; i < NQ; i += 4) {
                const float mn = fmaxf(fmaxf(red_m[i * 4 + 0], red_m[i * 4 + 1]), fmaxf(red_m[i * 4 + 2], red_m[i * 4 + 3]));
                float l = 0.f, o = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                  const float a = red_m[i * 4 + w4] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(red_m[i * 4 + w4] - mn);
                  l += red_l[i * 4 + w4] * a;
                  o += red_o[(i * 4 + w4) * 64 + lane] * a;
                }
                u64* dst = p.g + p.o_xp + ((((w * NQ + i) * H + h) * S + sl) * 66);
                if (!*fail) {
                  gst(dst + lane, tE, __float_as_uint(o));
                  if (lane == 0) { gst(dst + 64, tE, __float_as_uint(mn)); gst(dst + 65, tE, __float_as_uint(l)); }
                }
              }
            }
            PTRACE(layer == p.L / 2 && t == wg, 4, 3);
          }
        } else if (ph == 5) {
          // ---------------- F: merge of the key slices, one IO wave per (row, head) ----------------
          const int n_tasks = M * H;
          for (int t = wg + G * iw; t < n_tasks; t += 4 * G) {
            const int m = t / H, h = t % H;
            const int lane = opq(tid & 63);
            PTRACE(layer == p.L / 2 && t == wg, 5, 0);
            float res = 0.f;
            if (!(p.done && p.done[m / p.group])) {
              constexpr int MS = 16;
              // ONE round trip: o[lane] of every slice (all lanes), max and sum of slice k by lane k
              int ix[MS + 2]; unsigned rx[MS + 2];
              const int base = p.o_xp + ((m * H + h) * S) * 66;
#pragma unroll
              for (int k = 0; k < MS; ++k) ix[k] = base + (k < S ? k : 0) * 66 + lane;
              const int ks = lane < S ? lane : 0;
              ix[MS] = base + ks * 66 + 64; ix[MS + 1] = base + ks * 66 + 65;
              if (!poll<MS + 2>(p.g, ix, MS + 2, tE, rx, ctl)) break;
              const float my_m = lane < S ? __uint_as_float(rx[MS]) : -INFINITY, my_l = __uint_as_float(rx[MS + 1]);
              const float mm = wave_max(my_m);
              float l = 0.f, o = 0.f;
#pragma unroll
              for (int k = 0; k < MS; ++k) {
                if (k < S) {
                  const float mk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_m), k));
                  const float lk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_l), k));
                  const float a = mk == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mk - mm);
                  l += lk * a;
                  o += __uint_as_float(rx[k]) * a;
                }
              }
              res = o / l;
            } else {
              // a finished window: nothing to merge, but the ORDER the merge gives must stay - o_a still holds the self-attention
              // output phase C gathers, and G (which only waits for F) rewrites the residual granules phase D normalises.  The
              // head's cross query under D's tag says every C task has run (D's LayerNorm read all of C's output)
              int i1[1] = {p.o_qx + m * (d / 2) + h * 32 + (lane & 31)}; unsigned r1[1];
              if (!poll<1>(p.g, i1, 1, tD, r1, ctl)) break;
            }
            const float res_n = __shfl_down(res, 1, 64);
            if ((lane & 1) == 0 && !*fail) gst(p.g + p.o_a + m * (d / 2) + h * 32 + lane / 2, tF, pack2<T>(res, res_n));
            PTRACE(layer == p.L / 2 && t == wg, 5, 3);
          }
        } else {
          // ---------------- J: x += b2 + the slices' partial sums (fixed order), one IO wave per 16 columns ----------------
          const bool last = layer == p.L - 1;
          for (int t = wg + G * iw; t < nt_d; t += 4 * G) {
            PTRACE(layer == p.L / 2 && t == wg, 9, 0);
            for (int e = opq(lane); e < 16 * M; e += 64) {
              const int m = e >> 4, n = t * 16 + (e & 15);
              // the partials carry I's tag, the residual G's: the partials arrive last, so they are waited for first
              int ip[4]; unsigned rp[4];
#pragma unroll
              for (int sl = 0; sl < 4; ++sl) ip[sl] = p.o_mp + ((sl < p.nsplit ? sl : 0) * 16 + m) * d + n;
              if (!poll<4>(p.g, ip, p.nsplit, tI, rp, ctl)) break;
              int i1[1] = {p.o_x + m * d + n}; unsigned r1[1];
              if (!poll<1>(p.g, i1, 1, tG, r1, ctl)) break;
              float x = __uint_as_float(r1[0]) + lw.b2[n];
#pragma unroll
              for (int sl = 0; sl < 4; ++sl) if (sl < p.nsplit) x += __uint_as_float(rp[sl]);
              if (!*fail) {
                gst(p.g + p.o_x + m * d + n, tJ, __float_as_uint(x));
                if (last) p.x_out[(int64_t)m * d + n] = x;
              }
            }
            PTRACE(layer == p.L / 2 && t == wg, 9, 3);
          }
        }
      }
    }
  }
  // the launch is complete for this workgroup; workgroup 0 advances the epoch for the next launch (every workgroup that had
  // a task read it before its first task, and workgroup 0's last task depends on all of them)
  __syncthreads();
  if (wg == 0 && tid == 0) *p.epoch = (epoch + 1u) & 0x3fffffu;
}

template <typename T>
void launch_persist_step(const PersistParams& p, int grid, hipStream_t s) {
  if (p.M < 1 || p.M > 16 || p.group < 1 || p.group > PS_NQ || p.M % p.group != 0) throw Error(OHW_E_INVALID_ARG, "persistent step: 1..16 rows, at most 5 per window");
  if (p.d % 64 != 0 || p.d > 1280 || p.H * 64 != p.d) throw Error(OHW_E_INVALID_ARG, "persistent step: d_model must be a multiple of 64, at most 1280");
  if (p.S < 1 || p.S > 16 || p.nsplit < 1 || p.nsplit > 4) throw Error(OHW_E_INVALID_ARG, "persistent step: slices");
  const int kb_i = (4 * p.d / 32 + p.nsplit - 1) / p.nsplit + 1;
  if (kb_i > 56 || (kb_i + 3) / 4 > PS_NB || (p.d / 32 + 3) / 4 > PS_NB) throw Error(OHW_E_INVALID_ARG, "persistent step: K slice too long for the register file");
  ensure_dynamic_lds((const void*)persist_step_kernel<T>, PS_LDS);
  hipLaunchKernelGGL((persist_step_kernel<T>), dim3(grid), dim3(PS_THREADS), PS_LDS, s, p);
  HIP_CHECK(hipGetLastError());
}

template void launch_persist_step<bf16_t>(const PersistParams&, int, hipStream_t);
template void launch_persist_step<f16_t>(const PersistParams&, int, hipStream_t);

}  // namespace ohw

#ifdef OHW_TRACE
extern "C" int ohw_dbg_persist_trace_read(unsigned long long* out, int cap) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  const int n = cap < 256 * 10 * 8 ? cap : 256 * 10 * 8;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ohw::ps_trace_buf), (size_t)n * 8) != hipSuccess) return -1;
  return n;
}
#endif
