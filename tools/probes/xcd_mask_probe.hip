// Which XCDs do the bits of a hipExtStreamCreateWithCUMask mask select?  Launches a kernel on streams with different masks and
// counts the workgroups per XCC id.   hipcc --offload-arch=gfx950 -O2 tools/probes/xcd_mask_probe.hip -o tools/probes/xcd_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void who(int* xcc, int n) {
  if (threadIdx.x == 0 && (int)blockIdx.x < n) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    xcc[blockIdx.x] = (int)v;
  }
  // some work so that blocks spread over the CUs
  float a = threadIdx.x;
  for (int i = 0; i < 20000; ++i) a = a * 1.0001f + 0.5f;
  if (a == 123.f) xcc[0] = -1;
}

static void run(const char* name, const std::vector<int>& bits) {
  uint32_t mask[8] = {0};
  for (int b : bits) mask[b / 32] |= 1u << (b % 32);
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
  const int n = 4096;
  int* d;
  hipMalloc(&d, n * 4);
  hipMemset(d, 0xff, n * 4);
  hipLaunchKernelGGL(who, dim3(n), dim3(64), 0, s, d, n);
  hipStreamSynchronize(s);
  std::vector<int> h(n);
  hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  int cnt[16] = {0};
  for (int v : h) if (v >= 0 && v < 16) cnt[v]++;
  printf("%-28s (%3zu bits): workgroups per XCC:", name, bits.size());
  for (int i = 0; i < 8; ++i) printf(" %5d", cnt[i]);
  printf("\n");
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  std::vector<int> a, b, c, e;
  for (int i = 0; i < 64; ++i) a.push_back(i);                  // bits 0..63
  for (int i = 0; i < 256; ++i) if (i % 8 < 2) b.push_back(i);  // bits = 0,1 mod 8
  for (int i = 0; i < 32; ++i) c.push_back(i);                  // bits 0..31
  for (int i = 0; i < 256; ++i) if (i % 8 == 3) e.push_back(i); // bits = 3 mod 8
  run("bits 0..63", a);
  run("bits with i % 8 in {0, 1}", b);
  run("bits 0..31", c);
  run("bits with i % 8 == 3", e);
  return 0;
}
