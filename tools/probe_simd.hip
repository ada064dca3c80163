// Developer probe: which SIMD each wave of a 512-thread workgroup lands on (HW_REG_HW_ID bits 5:4), with and without
// a large dynamic LDS allocation.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(int* out) {
  extern __shared__ int lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = (int)id;
  if (threadIdx.x == 9999) lds[0] = 1;
}
int main() {
  int* d;
  hipMalloc(&d, 4096 * 8 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  int h[4096 * 8];
  for (int lds : {0, 152 * 1024}) {
    hipLaunchKernelGGL(k, dim3(4096), dim3(512), lds, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < 4096; ++b)
      for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    printf("lds %d: wave -> SIMD histogram over 4096 workgroups\n", lds);
    for (int w = 0; w < 8; ++w) printf("  wave %d: %5d %5d %5d %5d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    int same = 0;
    for (int b = 0; b < 4096; ++b) {
      bool ok = true;
      for (int w = 0; w < 4; ++w) ok = ok && (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3));
      same += ok;
    }
    printf("  workgroups where wave w and w+4 share a SIMD for all w: %d / 4096\n", same);
    for (int b = 0; b < 3; ++b) {
      printf("  wg %d:", b);
      for (int w = 0; w < 8; ++w) printf(" %d", (h[b * 8 + w] >> 4) & 3);
      printf("\n");
    }
  }
  return 0;
}
