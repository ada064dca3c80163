// What does a long kernel on a second stream cost a chain of small dependent kernels on the first one?  (The step timeline shows
// 40-50 us between consecutive encoder-backward kernels while the decoder's streaming 3x3 weight gradients run on the late stream,
// 0 us otherwise.)  Main stream: a captured graph of NCHAIN small kernels (52 workgroups x 256 threads, 32 KB of LDS, a few
// dependent loads each).  Second stream: ONE long kernel of G workgroups x T threads with L KB of LDS that either spins on the
// clock or streams memory for `us` microseconds.  Prints the chain's time alone and next to each flavour of long kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_overlap.hip -o /tmp/probe_overlap && /tmp/probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_small(const float* in, float* out, int n) {
  extern __shared__ float sm[];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v = in[i % n];
  sm[threadIdx.x] = v;
  __syncthreads();
  v += sm[(threadIdx.x * 7) & 255];
  const int j = ((int)(v * 1e-30f) + i * 17) % n;            // dependent second load
  v += in[j];
  out[i % n] = v;
}

template <int T>
__global__ __launch_bounds__(T) void k_long(float* buf, long long nfloat, long long ticks, int stream_mem, int wg_us, int* counter, int total_items) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();                         // 100 MHz
  float acc = 0.f;
  if (wg_us > 0) {
    // dynamic work items of wg_us microseconds each (workgroups retire and new ones start: gridDim.x = total_items)
    while (wall_clock64() - t0 < (long long)wg_us * 100) acc += sm[(threadIdx.x + (int)acc) & (T - 1)] * 1e-9f;
  } else if (stream_mem) {
    long long i = ((long long)blockIdx.x * T + threadIdx.x) * 4;
    while (wall_clock64() - t0 < ticks) {
      const float4 v = *reinterpret_cast<const float4*>(buf + (i % (nfloat - 4)));
      acc += v.x + v.y + v.z + v.w;
      i += (long long)gridDim.x * T * 4;
    }
  } else {
    while (wall_clock64() - t0 < ticks) acc += sm[(threadIdx.x + (int)acc) & (T - 1)] * 1e-9f;
  }
  if (acc == 123.456f) buf[0] = acc;
}

int main() {
  const int NCHAIN = 200, n = 1 << 20;
  float *in, *out, *big;
  const long long nbig = 256ll << 20;                        // 1 GiB of floats
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&big, nbig * 4));
  CK(hipMemset(in, 0, n * 4)); CK(hipMemset(big, 0, nbig * 4));
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_long<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_long<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < NCHAIN; ++i) hipLaunchKernelGGL(k_small, dim3(52), dim3(256), 32 * 1024, s1, in, out, n);
  CK(hipStreamEndCapture(s1, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto chain_ms = [&](auto&& launch_long) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipDeviceSynchronize());
      launch_long();
      CK(hipEventRecord(e0, s1));
      CK(hipGraphLaunch(ge, s1));
      CK(hipEventRecord(e1, s1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    return best;
  };
  const float alone = chain_ms([] {});
  printf("chain of %d small kernels alone: %.3f ms (%.2f us per kernel)\n", NCHAIN, alone, alone * 1e3f / NCHAIN);
  const long long ticks = 5000 * 100;                        // 5 ms of long kernel: covers the whole chain
  struct Cfg { int G, T, Lkb, mem, wg_us, items; };
  const std::vector<Cfg> cfgs = {
      {16, 512, 1, 0, 0, 0},    {160, 512, 1, 0, 0, 0},   {256, 512, 1, 0, 0, 0},   {160, 512, 64, 0, 0, 0},  {160, 512, 120, 0, 0, 0},
      {160, 512, 150, 0, 0, 0}, {256, 512, 120, 0, 0, 0}, {512, 512, 64, 0, 0, 0},  {160, 512, 120, 1, 0, 0}, {160, 256, 1, 1, 0, 0},
      {1024, 256, 1, 1, 0, 0},  {160, 512, 120, 0, 50, 160 * 100}, {256, 512, 120, 0, 50, 256 * 100}, {512, 512, 64, 0, 50, 512 * 100}};
  for (const Cfg& c : cfgs) {
    const float ms = chain_ms([&] {
      const int grid = c.wg_us ? c.items : c.G;
      if (c.T == 512) hipLaunchKernelGGL(k_long<512>, dim3(grid), dim3(512), c.Lkb * 1024, s2, big, nbig, ticks, c.mem, c.wg_us, nullptr, 0);
      else hipLaunchKernelGGL(k_long<256>, dim3(grid), dim3(256), c.Lkb * 1024, s2, big, nbig, ticks, c.mem, c.wg_us, nullptr, 0);
    });
    printf("next to %5d WGs x %3d thr, %3d KB LDS, %s%s: chain %.3f ms (%.2f us per kernel, +%.2f)\n", c.wg_us ? c.items : c.G, c.T, c.Lkb,
           c.mem ? "streaming memory" : "spinning", c.wg_us ? " (50-us workgroups, grid >> chip)" : "", ms, ms * 1e3f / NCHAIN,
           (ms - alone) * 1e3f / NCHAIN);
  }
  // several long kernels queued behind each other on the second stream (the head of that queue is then a packet waiting on its
  // predecessor), directly and as a captured graph
  for (int as_graph = 0; as_graph < 2; ++as_graph) {
    hipGraph_t g2 = nullptr; hipGraphExec_t ge2 = nullptr;
    if (as_graph) {
      CK(hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_long<512>, dim3(160), dim3(512), 120 * 1024, s2, big, nbig, 1000 * 100ll, 0, 0, nullptr, 0);
      CK(hipStreamEndCapture(s2, &g2));
      CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    }
    const float ms = chain_ms([&] {
      if (as_graph) { CK(hipGraphLaunch(ge2, s2)); }
      else for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_long<512>, dim3(160), dim3(512), 120 * 1024, s2, big, nbig, 1000 * 100ll, 0, 0, nullptr, 0);
    });
    printf("next to 5 queued 1-ms kernels (160 WGs, 120 KB) %s: chain %.3f ms (%.2f us per kernel, +%.2f)\n", as_graph ? "as a graph" : "launched directly",
           ms, ms * 1e3f / NCHAIN, (ms - alone) * 1e3f / NCHAIN);
  }
  return 0;
}
