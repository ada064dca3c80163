// Developer probe: SIMD placement of the waves of 256-thread workgroups when two of them share a CU (78 KB LDS each).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(int* out, int spin) {
  extern __shared__ int lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (int)id;
  // stay resident for a while so that workgroups pile up two per CU
  long long t0 = clock64();
  while (clock64() - t0 < spin) { if (threadIdx.x == 9999) lds[0] = 1; }
}
int main() {
  const int NB = 1024;
  int* d;
  (void)hipMalloc(&d, NB * 4 * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
  static int h[NB * 4];
  hipLaunchKernelGGL(k, dim3(NB), dim3(256), 78 * 1024, 0, d, 200000);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int distinct4 = 0, two = 0, other = 0;
  for (int b = 0; b < NB; ++b) {
    int cnt[4] = {0, 0, 0, 0};
    for (int w = 0; w < 4; ++w) cnt[(h[b * 4 + w] >> 4) & 3]++;
    int mx = 0, nz = 0;
    for (int s = 0; s < 4; ++s) { if (cnt[s] > mx) mx = cnt[s]; nz += cnt[s] > 0; }
    if (nz == 4) distinct4++; else if (nz == 2 && mx == 2) two++; else other++;
  }
  printf("workgroups with 4 waves on 4 distinct SIMDs: %d, on 2 SIMDs (2+2): %d, other: %d (of %d)\n", distinct4, two, other, NB);
  for (int b = 0; b < 8; ++b) {
    printf("  wg %d: simd", b);
    for (int w = 0; w < 4; ++w) printf(" %d", (h[b * 4 + w] >> 4) & 3);
    printf("   cu %d se %d  wave slots", (h[b * 4] >> 8) & 15, (h[b * 4] >> 13) & 7);
    for (int w = 0; w < 4; ++w) printf(" %d", h[b * 4 + w] & 15);
    printf("\n");
  }
  return 0;
}
