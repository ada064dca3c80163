// Same-address fp32/int atomic throughput across workgroups (memory-side serialisation) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probe_atomics.hip -o tools/probe_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_same(float* p, int naddr, int per_wg, int stride = 32) {
  // thread t < per_wg of every workgroup adds to address (t % naddr)
  if (threadIdx.x < per_wg) atomicAdd(p + (threadIdx.x % naddr) * stride, 1.0f);
}
__global__ void k_ret(int* p, int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = atomicAdd(p, 1);
}
int main() {
  float* p; hipMalloc(&p, 1 << 20); hipMemset(p, 0, 1 << 20);
  int* out; hipMalloc(&out, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {64, 256, 1024, 4096})
    for (int naddr : {1, 16, 256})
      for (int per : {1, 16, 256}) {
        if (naddr > per) continue;
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0); hipLaunchKernelGGL(k_same, dim3(wgs), dim3(256), 0, 0, p, naddr, per); hipEventRecord(e1);
          hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("wgs %5d  addresses %4d  atomics/wg %4d: %8.2f us  (%.1f ns per same-address op)\n", wgs, naddr, per, best * 1e3,
               best * 1e6 / ((double)wgs * per / naddr));
      }
  for (int stride : {1, 2, 4, 8, 16, 32, 64})
    for (int naddr : {16, 256}) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_same, dim3(1024), dim3(256), 0, 0, p, naddr, 256, stride); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("1024 wgs x 256 atomics over %3d addresses, stride %2d floats: %8.2f us\n", naddr, stride, best * 1e3);
    }
  for (int wgs : {64, 256, 1024}) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(k_ret, dim3(wgs), dim3(64), 0, 0, (int*)p, out); hipEventRecord(e1);
      hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("returning int atomic, 1 per wg, wgs %5d: %8.2f us\n", wgs, best * 1e3);
  }
  return 0;
}
