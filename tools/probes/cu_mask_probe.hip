// Does hipExtStreamCreateWithCUMask partition the CUs on this box, and how do mask bits map to (XCC, CU)?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cu_mask_probe.hip -o tools/probes/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void where(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the workgroup alive for a while so that the grid spreads over every CU it may use
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("%s CUs=%d\n", p.gcnArchName, p.multiProcessorCount);
  unsigned* d; const int N = 2048; CK(hipMalloc(&d, N * 4));
  std::vector<unsigned> h(N);
  for (int frac = 0; frac < 4; ++frac) {
    hipStream_t s;
    std::vector<uint32_t> mask(8, 0);
    const char* name;
    if (frac == 0) { for (auto& m : mask) m = 0xffffffffu; name = "all"; }
    else if (frac == 1) { for (int b = 0; b < 64; ++b) mask[b / 32] |= 1u << (b % 32); name = "bits 0..63"; }
    else if (frac == 2) { for (int b = 64; b < 256; ++b) mask[b / 32] |= 1u << (b % 32); name = "bits 64..255"; }
    else { for (int b = 0; b < 256; b += 4) mask[b / 32] |= 1u << (b % 32); name = "every 4th bit"; }
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
    if (e != hipSuccess) { printf("mask %-14s: create failed: %s\n", name, hipGetErrorString(e)); continue; }
    CK(hipMemsetAsync(d, 0xff, N * 4, s));
    hipLaunchKernelGGL(where, dim3(N), dim3(64), 0, s, d, 2000);   // 20 us each
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, N * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus; int per_xcc[16] = {0};
    for (unsigned v : h) { unsigned xcc = v >> 16, hw = v & 0xffff; unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      unsigned key = xcc << 12 | se << 8 | sh << 4 | cu; if (cus.insert(key).second) per_xcc[xcc]++; }
    printf("mask %-14s: %zu distinct CUs; per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  return 0;
}
