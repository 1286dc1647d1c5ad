// What does one dependency hop of a persistent decoder-step kernel cost on this chip, with the layer's weight stream running
// beside it?  (round 3, before building openhush_amd/csrc/decode_persist.hip)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/persist_probe.hip -o tools/probes/persist_probe
// One workgroup per CU, all resident.  A "token step" = LAYERS x the phase list below; in phase p every producing workgroup
// publishes out_per_wg values per row as 8-byte {tag, value} granules (one sc1 store each: MI355X_MICROARCH.md, R2), every
// workgroup sweeps the first `gather` granules of the phase until all carry the phase's tag, then "computes" on its share of
// the phase's weights (real nt loads of the real byte count, prefetched one phase ahead into registers).  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Phase { int producers, out_per_wg, gather, wbytes_per_wg; };
constexpr int MAXP = 12;
struct Params {
  Phase ph[MAXP];
  int n_phase, layers, rows, n_wg;
  unsigned long long* gran[2];   // [2][max granules]
  const u32x4* weights;          // streamed region
  size_t weights_elems;          // in u32x4
  unsigned* timeout;             // set on a spin timeout
  unsigned* sink;
  unsigned long long* stamps;    // [n_wg] total clocks in gather
  int do_stream;
};

constexpr int THREADS = 256;
constexpr int MAXLOAD = 8;   // 16-B loads per thread per phase kept in flight

__global__ __launch_bounds__(THREADS, 1) void persist(Params p) {
  __shared__ float vec[16 * 1024];
  __shared__ int fail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
  if (tid == 0) fail = 0;
  __syncthreads();
  unsigned acc = 0;
  size_t wpos = (size_t)wg * 4096;      // this CU's cursor in the weight stream (u32x4 units)
  u32x4 wreg[MAXLOAD];
  int nload_next = 0;
  auto issue = [&](const Phase& ph) {
    nload_next = p.do_stream ? min(MAXLOAD, (ph.wbytes_per_wg / 16 + THREADS - 1) / THREADS) : 0;
#pragma unroll
    for (int u = 0; u < MAXLOAD; ++u)
      if (u < nload_next) {
        size_t idx = (wpos + (size_t)u * THREADS + tid) % p.weights_elems;
        wreg[u] = __builtin_nontemporal_load(&p.weights[idx]);
      }
    wpos += (size_t)nload_next * THREADS * p.n_wg;
  };
  issue(p.ph[0]);
  unsigned long long t_gather = 0;
  unsigned epoch = 0;
  for (int layer = 0; layer < p.layers; ++layer) {
    for (int q = 0; q < p.n_phase; ++q) {
      const Phase ph = p.ph[q];
      ++epoch;
      // ---- gather the previous phase's outputs (phase 0 of layer 0 has nothing to wait for)
      if (epoch > 1 && wave == 0) {
        const Phase prev = p.ph[(q + p.n_phase - 1) % p.n_phase];
        const int n = prev.gather * p.rows;
        gu64* g = (gu64*)p.gran[(epoch - 1) & 1];
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        for (int base = 0; base < n; base += 64 * 8) {
          for (;;) {
            bool ok = true;
            unsigned v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int i = base + k * 64 + lane;
              const unsigned long long x = i < n ? __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)(epoch - 1) << 32);
              v[k] = (unsigned)x; ok &= (unsigned)(x >> 32) == epoch - 1;
            }
            if (__all(ok)) {
#pragma unroll
              for (int k = 0; k < 8; ++k) { const int i = base + k * 64 + lane; if (i < n && i < 16 * 1024) vec[i] = __uint_as_float(v[k]); }
              break;
            }
            if (++spins > (1u << 18)) { if (lane == 0) { fail = 1; atomicExch(p.timeout, epoch); } break; }
            __builtin_amdgcn_s_sleep(2);
          }
          if (fail) break;
        }
        t_gather += wall_clock64() - t0;
      }
      __syncthreads();
      if (fail || *(volatile unsigned*)p.timeout) { if (tid == 0) p.sink[wg] = acc; return; }
      // ---- "compute": consume this phase's weights, prefetch the next phase's
      const int nl = nload_next;
#pragma unroll
      for (int u = 0; u < MAXLOAD; ++u) if (u < nl) acc += wreg[u].x ^ wreg[u].y ^ wreg[u].z ^ wreg[u].w;
      acc += __float_as_uint(vec[tid & 1023]);
      {
        const int qn = (q + 1) % p.n_phase;
        issue(p.ph[qn]);
      }
      // ---- publish
      if (wg < ph.producers && wave == 0) {
        gu64* g = (gu64*)p.gran[epoch & 1];
        const int n = ph.out_per_wg * p.rows;
        for (int i = lane; i < n; i += 64)
          __hip_atomic_store(g + (size_t)wg * n + i, ((unsigned long long)epoch << 32) | (acc & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (tid == 0) { p.sink[wg] = acc; p.stamps[wg] = t_gather; }
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("%s CUs=%d\n", prop.gcnArchName, n_cu);
  const size_t wbytes = (size_t)1600 << 20;     // 1.6 GB: the decoder's weights
  void* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
  unsigned long long* gran[2];
  const size_t gmax = (size_t)512 * 1024;
  for (auto& g : gran) { CK(hipMalloc(&g, gmax * 8)); }
  unsigned *tmo, *sink; unsigned long long* stamps;
  CK(hipMalloc(&tmo, 64)); CK(hipMalloc(&sink, 4096 * 4)); CK(hipMalloc(&stamps, 4096 * 8));
  struct Cfg { const char* name; std::vector<Phase> ph; };
  const int d = 1280;
  // bytes per WG per phase: that GEMM's weights / 256
  auto wb = [&](long n, long k) { return (int)(n * k * 2 / 256); };
  std::vector<Cfg> cfgs = {
    {"8 all-to-all hops per layer (large-v3 shapes)", {
      {240, 16, 192, wb(3 * d, d)},      // LN1 + QKV -> heads
      {20, 64, 1280, 0},                 // self-attention -> out-proj
      {80, 16, 1280, wb(d, d)},          // out-proj (+resid) -> LNx + xq
      {80, 16, 64, wb(d, d)},            // xq -> cross-attention slices
      {240, 66, 12 * 66, 7680 * 1024 / 256},   // cross-attention partials (K/V stream) -> merge
      {20, 64, 1280, 0},                 // merge -> xo
      {80, 16, 1280, wb(d, d)},          // xo (+resid) -> LN2 + mlp.0
      {256, 20, 5120, wb(4 * d, d)},     // mlp.0 -> mlp.2
      {240, 16, 3 * 1280, wb(d, 4 * d)}, // mlp.2 partials -> next layer
    }},
    {"1 hop per layer, 1280 values from 80 producers", {{80, 16, 1280, wb(d, d)}}},
    {"1 hop per layer, 5120 values from 256 producers", {{256, 20, 5120, wb(4 * d, d)}}},
    {"1 hop per layer, 64 values from 4 producers", {{4, 16, 64, wb(d, d)}}},
  };
  for (int rows : {1, 5, 16})
    for (int stream : {1, 0})
      for (auto& c : cfgs) {
        Params p{};
        p.n_phase = (int)c.ph.size(); p.layers = 32 * (c.ph.size() == 1 ? 8 : 1); p.rows = rows; p.n_wg = n_cu;
        for (int i = 0; i < p.n_phase; ++i) p.ph[i] = c.ph[i];
        size_t need = 0;
        for (auto& ph : c.ph) need = std::max(need, (size_t)std::max(ph.producers * ph.out_per_wg, ph.gather) * rows);
        if (need > gmax) { printf("skip %s rows %d\n", c.name, rows); continue; }
        p.gran[0] = gran[0]; p.gran[1] = gran[1];
        p.weights = (const u32x4*)w; p.weights_elems = wbytes / 16; p.timeout = tmo; p.sink = sink; p.stamps = stamps; p.do_stream = stream;
        float best = 1e9f;
        unsigned tm = 0;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipMemset(gran[0], 0, gmax * 8)); CK(hipMemset(gran[1], 0, gmax * 8)); CK(hipMemset(tmo, 0, 64));
          hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(persist, dim3(n_cu), dim3(THREADS), 0, 0, p);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          CK(hipMemcpy(&tm, tmo, 4, hipMemcpyDeviceToHost));
          if (tm) break;
          best = std::min(best, ms);
          CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        const int hops = p.layers * p.n_phase;
        if (tm) printf("rows %2d stream %d  %-52s TIMEOUT at epoch %u\n", rows, stream, c.name, tm);
        else printf("rows %2d stream %d  %-52s %8.1f us per step  = %6.2f us per hop (%d hops)\n", rows, stream, c.name, best * 1e3f, best * 1e3f / hops, hops);
        fflush(stdout);
      }
  return 0;
}
