// gemm256.hip — 256x256x64 MFMA GEMM with direct-to-LDS loads kept in flight across barriers (gfx950).
//
// Same contract and epilogues as gemm.hip (C = A * W^T, swapped MFMA operands, 16 contiguous output
// columns per lane), for the large encoder / cross-K/V GEMMs (N % 256 == 0).
//
// Structure (after cdna_hip_programming.md section 5, "Pipelining across barriers" / the 8-phase idea):
//   * 512 threads = 8 waves (2 along m x 4 along n), each wave a 128x64 output tile = 8x4 MFMA tiles of
//     16x16x32 (128 accumulator VGPRs); one workgroup per CU.
//   * LDS: ONE 160-KiB array = 3 stages of the A tile (256 rows x 64 k) + 2 stages of the W tile.  Rows are 128 B = one whole
//     cache line of the operand: a global_load_lds_dwordx4 wave-instruction brings 8 rows x 128 B (the first version staged
//     k-halves with 64-byte rows, i.e. every line was fetched in two half-used pieces by two different instructions).  The eight
//     16-byte chunks of a row are XOR-swizzled with (row >> 1) & 7, which makes every ds_read_b128 fragment read
//     conflict-free; the LDS destination of the DMA is lane-linear, so the swizzle goes on the per-lane SOURCE
//     address and on the read.
//   * a K-tile is consumed in two phases (k 0..31, then k 32..63), 32 MFMAs per wave each.  The two waves of
//     a SIMD are staggered by one barrier interval: one reads its fragments for the next phase while the
//     other runs MFMAs on fragments already in registers.  In K-tile kt's first read interval a wave requests its share of
//     W(kt + 1) and then of A(kt + 2) (4 DMA instructions each), into the stages K-tile kt - 1 was read from; the wait in front
//     of K-tile kt + 1 is s_waitcnt vmcnt(4) - everything but that A tile - + raw s_barrier.  The activation panels come from
//     beyond L2 for the first tile of an XCD that touches them (all of them when N = 1280: five n-tiles per panel), and one
//     K-tile time (1.5 us) does not cover that latency: the third A stage took 2 - 4.5 % off the N = 1280 launches (round 3,
//     tools/gemm_probe.py: 515 / 819 / 1505 us at K = 1280 / 2560 / 5120 against 527 / 844 / 1575), 0 - 1.5 % off the others.
// Roofline: MFMA.
#include <algorithm>
#include <cstdlib>

#include "gemm.hpp"
#include "gemm_epilogue.hpp"

namespace ohw {

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64;
constexpr int G2_THREADS = 512;
constexpr int G2_OPSTAGE = 32768;  // bytes of one operand's K-tile: 256 rows x 128 B
constexpr int G2_NA = 3;           // A stages at 0, 32 KiB, 64 KiB: the activation panels mostly come from beyond L2 (one use per m-tile and XCD)
constexpr int G2_WBASE = G2_NA * G2_OPSTAGE;   // 2 W stages behind them (the weight panels stay in the L2s)
constexpr int G2_LDS = G2_WBASE + 2 * G2_OPSTAGE;   // 160 KiB

__device__ __forceinline__ int g2_key(int row) { return (row >> 1) & 7; }

#ifdef OHW_TRACE
// instrumented build only (tools/gemm_trace.py): per workgroup {tile id, XCC_ID | CU id, entry, first K-tile landed, main loop
// done, exit} on the 100 MHz clock
constexpr unsigned G2_TRACE_CAP = 1u << 16;
__device__ unsigned long long g2_trace_buf[G2_TRACE_CAP * 12];
__device__ unsigned g2_trace_n;
#define G2T(k) do { if (tid == 0 && g2_slot < G2_TRACE_CAP) g2_trace_buf[g2_slot * 12 + (k)] = wall_clock64(); } while (0)
#else
#define G2T(k) do { } while (0)
#endif

template <typename T, int EPI>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm256_kernel(GemmParams p) {
  using Ops = TypeOps<T>;
  using vec8 = typename Ops::vec8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
#ifdef OHW_TRACE
  unsigned g2_slot = G2_TRACE_CAP;
  if (tid == 0) {
    g2_slot = atomicAdd(&g2_trace_n, 1u);
    if (g2_slot < G2_TRACE_CAP) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g2_trace_buf[g2_slot * 12 + 0] = blockIdx.x;
      g2_trace_buf[g2_slot * 12 + 1] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    }
  }
  G2T(2);
#endif

  const unsigned n_tiles_n = (unsigned)(p.N / G2_BN);
  const unsigned n_tiles_m = (unsigned)((p.M + G2_BM - 1) / G2_BM);
  const unsigned nwg = n_tiles_n * n_tiles_m;
  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;

  // ---- staging: per stage each operand is 32 blocks of 1 KiB (8 rows x 128 B); wave w fills blocks 4w .. 4w+3.
  // lane -> row r = lane>>3 of the block, LDS chunk c = lane&7, holding data chunk c ^ key(row).
  const int sr = lane >> 3, sc = lane & 7;
  unsigned a_off[4], w_off[4];       // BYTE offsets of this lane's four source rows from A / W (< 4 GiB, checked by the launcher): 8 registers, not 16
  int64_t m0 = 0, n0 = 0;
  auto tile_setup = [&](unsigned d) {
    const unsigned lid = xcd_remap(d, nwg);
    // tile order inside an XCD's contiguous range: groups of GM m-tiles (6; 8 in round 1), m fastest, so that the ~32 workgroups
    // an XCD runs at a time cover GM m-tiles x 32 / GM n-tiles (about 12 operand panels through its 4-MiB L2 instead of 17)
    const unsigned GM = (unsigned)p.group_m;
    const unsigned per_group = GM * n_tiles_n;
    const unsigned grp_id = lid / per_group, in_grp = lid % per_group;
    const unsigned g_first = grp_id * GM;
    const unsigned g_size = n_tiles_m - g_first < GM ? n_tiles_m - g_first : GM;
    m0 = (int64_t)(g_first + in_grp % g_size) * G2_BM;
    n0 = (int64_t)(in_grp / g_size) * G2_BN;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int R = (wave * 4 + j) * 8 + sr;          // LDS row 0..255
      const int dchunk = sc ^ g2_key(R);
      int64_t m = m0 + R;
      if (m > p.M - 1) m = p.M - 1;
      int64_t b = 0, rr = m;
      if (p.rows_per_batch < p.M) { const unsigned bb = (unsigned)m / (unsigned)p.rows_per_batch; b = bb; rr = m - (int64_t)bb * p.rows_per_batch; }
      a_off[j] = (unsigned)((b * p.a_batch_stride + rr * p.lda + dchunk * 8) * 2);
      // LDS row rho (inside each 64-row block: rho = ni*16 + q*4 + jj) holds W row q*16 + ni*4 + jj
      const int rl = R & 63;
      const int nl = (((rl >> 2) & 3) << 4) + ((rl >> 4) << 2) + (rl & 3);
      w_off[j] = (unsigned)(((n0 + (R & ~63) + nl) * p.K + dchunk * 8) * 2);
    }
  };
  const int KT = (int)(p.K / G2_BK);

  // the DMA instructions of this wave for K-tile kt: 4 for its share of the A tile, 4 for the W tile
  auto issue_A = [&](int kt) {
    const unsigned koff = (unsigned)(kt * G2_BK * 2);
    const int buf = (kt % G2_NA) * G2_OPSTAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)A + (size_t)(a_off[j] + koff)),
                                       (__attribute__((address_space(3))) void*)(smem + buf + (wave * 4 + j) * 1024), 16, 0, 0);
  };
  auto issue_W = [&](int kt) {
    const unsigned koff = (unsigned)(kt * G2_BK * 2);
    const int buf = G2_WBASE + (kt & 1) * G2_OPSTAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)W + (size_t)(w_off[j] + koff)),
                                       (__attribute__((address_space(3))) void*)(smem + buf + (wave * 4 + j) * 1024), 16, 0, 0);
  };
  // K-tile kt's read interval: W(kt + 1) first, then A(kt + 2) - the wait in front of K-tile kt + 1 leaves exactly the four
  // youngest operations (that A tile) in flight
  auto issue_ahead = [&](int kt) {
    if (kt + 1 < KT) issue_W(kt + 1);
    if (kt + 2 < KT) issue_A(kt + 2);
  };

  // fragment read offsets inside a stage: row = tile row + (lane & 15), chunk (4h + (lane >> 4)) ^ key(row);
  // tile rows start at multiples of 16, so key(row) depends on (lane & 15) only
  const int fr = lane & 15, fq = lane >> 4;
  const int fkey = g2_key(fr);
  const int a_row = (wm * 128 + fr) * 128;
  const int w_row = G2_WBASE + (wn * 64 + fr) * 128;

  // A "phase" P = 2*kt + h consumes k-half h of K-tile kt (32 MFMAs per wave) from stage kt & 1.
  //
  // The two waves that share a SIMD (wave w and w + 4, i.e. wm = 0 / 1) run the SAME program shifted by one
  // barrier interval: in every interval one of them issues its 12 fragment reads for the next phase while
  // the other runs the 32 MFMAs of the fragments it read one interval earlier, so the MFMA pipe of a SIMD
  // always has a wave with operands in registers (MI355X_MICROARCH.md "Two waves per SIMD", item 9):
  //     interval i, wave group g (0/1), j = i - g:   j even -> read phase j/2      j odd -> compute phase j/2
  // Stage kt & 1 is last read in phase 2kt + 1 (intervals 4kt + 2 and 4kt + 3) and refilled with tile kt + 2: each wave
  // issues its share in its read interval of phase 2kt + 2 (4kt + 4 for group 0, 4kt + 5 for group 1), never while it
  // computes; the data is first read in interval 4kt + 8, behind vmcnt(0) at the barrier that opens it (each wave's
  // youngest DMA group at that point is that tile's).
  const int NP = 2 * KT;
  f32x4 acc[8][4];
  vec8 fw[4], fa[8];
  auto read_frags = [&](int P) {
    const int kt = P >> 1;
    const int acur = (kt % G2_NA) * G2_OPSTAGE, wcur = (kt & 1) * G2_OPSTAGE;
    const int ch = (((P & 1) * 4 + fq) ^ fkey) << 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fw[ni] = *(const vec8*)(smem + wcur + w_row + ch + ni * 2048);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) fa[mi] = *(const vec8*)(smem + acur + a_row + ch + mi * 2048);
  };
  auto compute = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Ops::mfma16(fw[ni], fa[mi], acc[mi][ni]);
    __builtin_amdgcn_s_setprio(0);
  };
  // barrier that opens an EVEN interval 2*Pn: when Pn starts a K-tile k, every wave's share of A(k) and W(k) must have landed; the A tile
  // for k + 1 (this wave's four youngest operations, if there is one) stays in flight
  auto open_even = [&](int Pn) {
    if (Pn < NP && (Pn & 1) == 0) {
      if ((Pn >> 1) + 1 < KT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto open_odd = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto raw_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // One tile per workgroup.  (Round 3 also built the persistent form - a workgroup per CU walking tiles d, d + G, ... with the next
  // tile's first K-tile requested inside the epilogue; it took the 1.4 us of first bytes and the 0.9 us of dispatch out of a tile
  // and gave them back to the imbalance of a static tile assignment: DESIGN.md Appendix A.  Removed with the third A stage.)
  const unsigned d = blockIdx.x;
  tile_setup(d);
  issue_A(0); issue_W(0);
  if (KT > 1) issue_A(1);
  {
    // this tile's place (tile_setup's m0 / n0 move on to the NEXT tile inside the epilogue)
    const int64_t m0c = m0, n0c = n0;
    // the lane's 16 bias values, requested before the main loop (four 16-byte loads): at the head of the epilogue they were 16 branchy
    // dword loads and a full memory round trip in every tile (gemm_trace: stores issued 4.2 us after the last MFMA)
    const int64_t nb = n0c + wn * 64 + fq * 16;
    f32x4 bias4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bias4[j] = *(const f32x4*)(p.bias + nb + 4 * j);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (wm == 0) {
      // group 0: interval 2P reads phase P, interval 2P+1 computes it
      for (int P = 0; P < NP; ++P) {
        open_even(P);
#ifdef OHW_TRACE
        if (P == 0) G2T(3);
#endif
        read_frags(P);
        if ((P & 1) == 0) issue_ahead(P >> 1);   // into the stages K-tile kt - 1 was read from: their readers are behind the barrier above
        __builtin_amdgcn_sched_barrier(0);
        open_odd();
        compute();
        __builtin_amdgcn_sched_barrier(0);
      }
      open_even(NP);   // the partner group's last compute interval
    } else {
      // group 1: one interval behind: interval 2P+1 reads phase P, interval 2P+2 computes it
      open_even(0);
      for (int P = 0; P < NP; ++P) {
        open_odd();
        read_frags(P);
        if ((P & 1) == 0) issue_ahead(P >> 1);   // this wave's share, one interval after the partner group's
        __builtin_amdgcn_sched_barrier(0);
        open_even(P + 1);
        compute();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    G2T(4);

    // ---- epilogue: lane (fq, fr) holds, for each mi, columns n0 + wn*64 + fq_e*16 + [0,16) of row m ----
    // (its addresses are built on an opaque copy of the lane id: they are not to be computed ahead of the main loop and parked in
    // registers through it)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int fr_e = lane_e & 15, fq_e = lane_e >> 4;
    float bias[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bias[j] = bias4[j >> 2][j & 3];
    bool stored = false;
    if constexpr (EPI == EPI_BIAS_RESID_F32) {
      // fp32 residual read-modify-write through LDS.  In the MFMA layout a lane owns 64 B of a row and a wave-instruction
      // touches 16 rows x 4 x 16 B: 32 half-used cache lines per instruction - a tile's 512 KB of residual traffic took
      // ~35 us that way (the K = 1280 out-projection ran at half the rate of the same shape with a 16-bit store).  The
      // finished tile goes to the (now free) LDS, 128 rows at a time, chunk-swizzled by row, and every wave then walks
      // whole rows: 64 lanes x 16 B = one 1-KiB row of the tile = 8 full lines per instruction.
      float* outf = (float*)p.out;
      // Two rounds; in round r EVERY wave sends its accumulator rows mi = 4r .. 4r + 3 to LDS (local row wm * 64 + q * 16 + fr_e of 128,
      // 1 KiB each), then wave w walks local rows w * 16 + [0, 16) whole: tile row (w >> 2) * 128 + r * 64 + (w & 3) * 16 + i.
      // The old values of a round's 16 rows are requested in ONE go - round 0's before the tile goes to LDS, round 1's as soon as
      // round 0's accumulator registers are free - behind raw barriers that leave them in flight (a __syncthreads drains every
      // outstanding load).  Measured (tools/gemm_trace.py, 96 windows): 15 us per tile either way - the 512 KB a tile reads and
      // writes here move at the 35 GB/s ONE CU gets from beyond its L2 while the other CUs run their main loops, whatever the
      // order of the requests; moving the add into the following LayerNorm launch would cost that launch more (it runs at HBM
      // rate on every CU: + 1.1 GB per launch at 96 windows = 0.18 ms against 0.11 ms saved here).
      auto row_ptr = [&](int r, int i) -> float* {
        int64_t m = m0c + (wave >> 2) * 128 + r * 64 + (wave & 3) * 16 + i;
        if (m > p.M - 1) m = p.M - 1;                 // unconditional loads (clamped row), masked stores
        int64_t bb = 0, rr = m;
        if (p.rows_per_batch < p.M) { const unsigned q = (unsigned)m / (unsigned)p.rows_per_batch; bb = q; rr = m - (int64_t)q * p.rows_per_batch; }
        return outf + bb * p.c_batch_stride + rr * p.ldc + n0c + lane_e * 4;
      };
      auto to_lds = [&](int r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = wm * 64 + q * 16 + fr_e;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const int chunk = (wn * 16 + fq_e * 4 + ni) ^ (row & 15);
            f32x4 v4 = acc[4 * r + q][ni];
            v4.x += bias[4 * ni]; v4.y += bias[4 * ni + 1]; v4.z += bias[4 * ni + 2]; v4.w += bias[4 * ni + 3];
            *(f32x4*)(smem + row * 1024 + chunk * 16) = v4;
          }
        }
      };
      f32x4 old0[16], old1[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) old0[i] = *(const f32x4*)row_ptr(0, i);
      G2T(6);
      to_lds(0);
      raw_barrier();
      G2T(7);
#pragma unroll
      for (int i = 0; i < 16; ++i) old1[i] = *(const f32x4*)row_ptr(1, i);      // into the registers acc[0..3] no longer need
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = wave * 16 + i;
        if (m0c + (wave >> 2) * 128 + (wave & 3) * 16 + i >= p.M) continue;
        const f32x4 v4 = *(const f32x4*)(smem + lr * 1024 + ((lane_e ^ (lr & 15)) << 4));
        f32x4 o4 = old0[i];
        o4.x += v4.x; o4.y += v4.y; o4.z += v4.z; o4.w += v4.w;
        *(f32x4*)row_ptr(0, i) = o4;
      }
      G2T(8);
      raw_barrier();                         // round 0 has been read out of LDS
      to_lds(1);
      raw_barrier();
      G2T(9);
      // round 1: LDS -> registers (the accumulator's are all free now)
      f32x4 v1[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lr = wave * 16 + i;
        v1[i] = *(const f32x4*)(smem + lr * 1024 + ((lane_e ^ (lr & 15)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (m0c + (wave >> 2) * 128 + 64 + (wave & 3) * 16 + i >= p.M) continue;
        f32x4 o4 = old1[i];
        o4.x += v1[i].x; o4.y += v1[i].y; o4.z += v1[i].z; o4.w += v1[i].w;
        *(f32x4*)row_ptr(1, i) = o4;
      }
      G2T(10);
      G2T(11);
      stored = true;
    }
    if constexpr (EPI == EPI_BIAS_T || EPI == EPI_BIAS_GELU_T) {
      {
        // 16-bit row-major output through the (now free) LDS: in the MFMA layout a store instruction writes 16 rows x 4 separate
        // 16-byte pieces; here every instruction writes two whole 512-byte rows of the tile.  Row pitch 512 B, the 32 chunks of a
        // row XOR-swizzled with row & 15 (conflict-free both ways).
        G2T(9);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const int row = wm * 128 + mi * 16 + fr_e;
          float v[16];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[ni * 4 + j] = acc[mi][ni][j] + bias[ni * 4 + j];
          if constexpr (EPI == EPI_BIAS_GELU_T) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = gelu_erf(v[j]);
          }
          u32x4 lo, hi;
          lo.x = pack2<T>(v[0], v[1]); lo.y = pack2<T>(v[2], v[3]); lo.z = pack2<T>(v[4], v[5]); lo.w = pack2<T>(v[6], v[7]);
          hi.x = pack2<T>(v[8], v[9]); hi.y = pack2<T>(v[10], v[11]); hi.z = pack2<T>(v[12], v[13]); hi.w = pack2<T>(v[14], v[15]);
          const int c0 = wn * 8 + fq_e * 2;
          *(u32x4*)(smem + row * 512 + ((c0 ^ fr_e) << 4)) = lo;
          *(u32x4*)(smem + row * 512 + (((c0 + 1) ^ fr_e) << 4)) = hi;
        }
        G2T(7);
        raw_barrier();
        G2T(8);
        T* outp = (T*)p.out;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int r = half * 128 + wave * 16 + 2 * i + (lane_e >> 5);
            const int64_t m = m0c + r;
            if (m >= p.M) continue;
            int64_t bb = 0, rr = m;
            if (p.rows_per_batch < p.M) { const unsigned q = (unsigned)m / (unsigned)p.rows_per_batch; bb = q; rr = m - (int64_t)q * p.rows_per_batch; }
            const u32x4 d4 = *(const u32x4*)(smem + r * 512 + (((lane_e & 31) ^ (r & 15)) << 4));
            *(u32x4*)(outp + bb * p.c_batch_stride + rr * p.ldc + n0c + (lane_e & 31) * 8) = d4;
          }
        }
        stored = true;
      }
    }
    if (!stored) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int64_t m = m0c + wm * 128 + mi * 16 + fr_e;
        if (m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[ni * 4 + j] = acc[mi][ni][j] + bias[ni * 4 + j];
        gemm_store_row<T, EPI>(p, m, n0c + wn * 64 + fq_e * 16, v);
      }
    }
#ifdef OHW_TRACE
    G2T(6 + (EPI == EPI_BIAS_RESID_F32 ? 5 : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2T(5);
#endif
  }
}

template <typename T, int EPI>
static void launch256_one(const GemmParams& p, hipStream_t stream) {
  const unsigned nwg = (unsigned)((p.N / G2_BN) * ((p.M + G2_BM - 1) / G2_BM));
  ensure_dynamic_lds((const void*)gemm256_kernel<T, EPI>, G2_LDS);
  GemmParams q = p;
  static const int gm_env = [] { const char* e = getenv("OHW_GEMM_GM"); return e ? atoi(e) : 0; }();
  q.group_m = gm_env > 0 ? gm_env : 6;   // 6: 1 % faster than 8 at 32 and at 96 windows per pass (round 2 sweep: 4, 6, 8, 16)
  hipLaunchKernelGGL((gemm256_kernel<T, EPI>), dim3(nwg), dim3(G2_THREADS), G2_LDS, stream, q);
  HIP_CHECK(hipGetLastError());
}

template <typename T>
void launch_gemm256(const GemmParams& p, int epilogue, hipStream_t stream) {
  if (p.M <= 0) return;
  if (p.N % G2_BN != 0 || p.K % G2_BK != 0 || p.lda % 8 != 0 || p.a_batch_stride % 8 != 0 || p.rows_per_batch <= 0 || p.M >= ((int64_t)1 << 31) || p.N >= ((int64_t)1 << 31))
    throw Error(OHW_E_INVALID_ARG, "gemm256: N must be a multiple of 256, K of 64, row strides of 8 elements");
  {  // the kernel addresses its operands by 32-bit byte offsets
    const int64_t nb_ = (p.M + p.rows_per_batch - 1) / p.rows_per_batch;
    const int64_t a_bytes = ((nb_ - 1) * p.a_batch_stride + (std::min(p.M, p.rows_per_batch) + 2) * p.lda + p.K) * 2, w_bytes = p.N * p.K * 2;
    if (a_bytes >= ((int64_t)1 << 32) || w_bytes >= ((int64_t)1 << 32)) throw Error(OHW_E_INVALID_ARG, "gemm256: an operand spans 4 GiB or more");
  }
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

#ifdef OHW_TRACE
// instrumented build only: copy the per-workgroup records out and reset (returns the number of records)
extern "C" int ohw_dbg_gemm_trace_read(unsigned long long* out, int cap_records) {
  unsigned n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(ohw::g2_trace_n), sizeof(n)) != hipSuccess) return -1;
  if (n > ohw::G2_TRACE_CAP) n = ohw::G2_TRACE_CAP;
  if ((int)n > cap_records) n = (unsigned)cap_records;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(ohw::g2_trace_buf), (size_t)n * 96) != hipSuccess) return -1;
  const unsigned zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(ohw::g2_trace_n), &zero, sizeof(zero)) != hipSuccess) return -1;
  return (int)n;
}
#endif
