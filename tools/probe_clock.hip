// Developer probe: shader clock (s_memtime) against the constant 100 MHz counter (s_memrealtime) under three loads:
// one idle-ish workgroup, every CU spinning on VALU, every CU issuing back-to-back bf16 MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_clock.hip -o tools/probe_clock ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void k(int mode, int iters, unsigned long long* out, float* sink) {
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[4] = {};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {(short)threadIdx.x, 1, 1, 1, 1, 1, 1, 1};
  float v = threadIdx.x;
  if (mode == 3) {            // random bf16 operands in [-2, 2): every multiplier input toggles from lane to lane
    unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    for (int j = 0; j < 8; ++j) {
      h ^= h << 13; h ^= h >> 17; h ^= h << 5;
      a[j] = (short)((h & 0x807f) | 0x3f80);          // sign + mantissa random, exponent 0 or 1
      h ^= h << 13; h ^= h >> 17; h ^= h << 5;
      b[j] = (short)((h & 0x807f) | 0x3f80);
    }
  }
  for (int i = 0; i < iters; ++i) {
    if (mode >= 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) v = v * 1.0001f + 0.5f;
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
  if (v == 123.f || acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 77.f) sink[0] = v;
}

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 16); hipMalloc(&sink, 4);
  const char* names[4] = {"one workgroup, VALU", "1024 workgroups x 256 threads, VALU", "1024 workgroups, MFMA 32x32x16 bf16, constant operands",
                          "1024 workgroups, MFMA 32x32x16 bf16, random operands"};
  for (int mode = 0; mode < 4; ++mode) {
    const int grid = mode == 0 ? 1 : 1024, iters = mode >= 2 ? 400000 : 200000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ev_ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, mode, iters, out, sink);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ev_ms, e0, e1);
    }
    unsigned long long h[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double ms = h[1] / 100e6 * 1e3;
    printf("%-52s %10llu s_memtime ticks, %9llu s_memrealtime ticks (%.3f ms at 100 MHz), events %.3f ms -> %.3f GHz by events", names[mode],
           h[0], h[1], ms, ev_ms, h[0] / (ev_ms * 1e-3) / 1e9);
    if (mode >= 2) printf("   %.0f TFLOP/s by events", 1024.0 * 4 * iters * 4 * 32768.0 / (ev_ms * 1e-3) / 1e12);
    printf("\n");
  }
  return 0;
}
