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
#ifdef WITH_LIB
#include "camradepth_hip.h"
#endif
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
  } else if (stream_mem == 2) {
    // 16 x 16-byte loads in flight per lane (what a wave of the streaming weight-gradient kernel keeps outstanding)
    long long i = ((long long)blockIdx.x * T + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * T * 4;
    while (wall_clock64() - t0 < ticks) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(buf + ((i + u * step) % (nfloat - 4)));
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
      i += 16 * step;
    }
  } else if (stream_mem == 4 || stream_mem == 5) {
    // L2-resident working set (2 MB per XCD's worth of lines: all hits after the first pass), 16 loads in flight per lane;
    // 5: the same through LDS-DMA (buffer_load ... lds), as the streaming kernels issue them
    const long long span = 512 * 1024;                       // floats = 2 MB
    long long i = ((long long)blockIdx.x * T + threadIdx.x) * 4;
    const long long step = (long long)T * 4;
    if (stream_mem == 4) {
      while (wall_clock64() - t0 < ticks) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(buf + ((i + u * step) % span));
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        i += 16 * step;
      }
    } else {
      typedef __attribute__((address_space(3))) void* lds_ptr;
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)(span * 4), 0x00020000);
      const int wv = threadIdx.x >> 6;
      while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sm + wv * 256 * 16 + u * 256), 16, (unsigned)(((i + u * step) % span) * 4), 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        i += 16 * step;
      }
    }
  } else if (stream_mem == 3) {
    // the same with stores as well (dirty lines in L2 for the end-of-kernel write-back of the chain's kernels)
    long long i = ((long long)blockIdx.x * T + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * T * 4;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
      for (int u = 0; u < 16; ++u) *reinterpret_cast<float4*>(buf + ((i + u * step) % (nfloat - 4))) = make_float4(acc, 1.f, 2.f, 3.f);
      i += 16 * step;
    }
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
      {1024, 256, 1, 1, 0, 0},  {160, 512, 120, 2, 0, 0}, {64, 512, 120, 2, 0, 0}, {160, 512, 120, 3, 0, 0}, {160, 512, 120, 4, 0, 0}, {160, 512, 140, 5, 0, 0}, {64, 512, 140, 5, 0, 0}, {160, 512, 120, 0, 50, 160 * 100}, {256, 512, 120, 0, 50, 256 * 100}, {512, 512, 64, 0, 50, 512 * 100}};
  for (const Cfg& c : cfgs) {
    const float ms = chain_ms([&] {
      const int grid = c.wg_us ? c.items : c.G;
      if (c.T == 512) hipLaunchKernelGGL(k_long<512>, dim3(grid), dim3(512), c.Lkb * 1024, s2, big, nbig, ticks, c.mem, c.wg_us, nullptr, 0);
      else hipLaunchKernelGGL(k_long<256>, dim3(grid), dim3(256), c.Lkb * 1024, s2, big, nbig, ticks, c.mem, c.wg_us, nullptr, 0);
    });
    printf("next to %5d WGs x %3d thr, %3d KB LDS, %s%s: chain %.3f ms (%.2f us per kernel, +%.2f)\n", c.wg_us ? c.items : c.G, c.T, c.Lkb,
           c.mem == 5 ? "L2-resident LDS-DMA loads, 16 deep" : c.mem == 4 ? "L2-resident loads, 16 deep" : c.mem == 3 ? "streaming stores, 16 deep" : (c.mem == 2 ? "streaming loads, 16 deep" : (c.mem ? "streaming memory" : "spinning")), c.wg_us ? " (50-us workgroups, grid >> chip)" : "", ms, ms * 1e3f / NCHAIN,
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
#ifdef WITH_LIB
  // the real thing: the streaming 3x3 weight gradient of the largest decoder layer (8 x 256 x 416, 304 -> 128) from
  // libcamradepth_hip.so on the second stream, with the late-stream workgroup budget `cap` (build: -DWITH_LIB -Iinclude + the .so)
  {
    const int B = 8, H = 256, W = 416, Cin = 304, Cout = 128;
    void *x, *dy; float* parts;
    CK(hipMalloc(&x, (size_t)B * H * W * Cin * 2)); CK(hipMalloc(&dy, (size_t)B * H * W * Cout * 2));
    CK(hipMemset(x, 0, (size_t)B * H * W * Cin * 2)); CK(hipMemset(dy, 0, (size_t)B * H * W * Cout * 2));
    CK(hipMalloc(&parts, (size_t)64 * Cout * 9 * Cin * 4));
    crd_sum_t* dwbuf; CK(hipMalloc(&dwbuf, (size_t)Cout * 9 * Cin * 8));
    // chains of REAL encoder-sized kernels: crd_gn_apply on a stage-4 / stage-3 / stage-1 hidden tensor (8 x 104 x 1024, 8 x 416 x 640,
    // 8 x 6656 x 512 bf16), 100 launches captured as one graph
    struct RealChain { int P, C; hipGraphExec_t ge; float alone; };
    std::vector<RealChain> rcs = {{104, 1024, nullptr, 0.f}, {416, 640, nullptr, 0.f}, {6656, 512, nullptr, 0.f}};
    void *gx, *gy; crd_sum_t* gst; float *gga, *gbe;
    CK(hipMalloc(&gx, (size_t)8 * 6656 * 1024 * 2)); CK(hipMalloc(&gy, (size_t)8 * 6656 * 1024 * 2));
    CK(hipMemset(gx, 0, (size_t)8 * 6656 * 1024 * 2));
    CK(hipMalloc(&gst, 8 * 64 * 2 * 8)); CK(hipMemset(gst, 0, 8 * 64 * 2 * 8));
    CK(hipMalloc(&gga, 1024 * 4)); CK(hipMalloc(&gbe, 1024 * 4)); CK(hipMemset(gga, 0, 4096)); CK(hipMemset(gbe, 0, 4096));
    auto time_graph = [&](hipGraphExec_t gexec, auto&& launch_long) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipDeviceSynchronize());
        launch_long();
        CK(hipEventRecord(e0, s1)); CK(hipGraphLaunch(gexec, s1)); CK(hipEventRecord(e1, s1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      return best;
    };
    for (RealChain& rc : rcs) {
      hipGraph_t gg;
      CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 100; ++i)
        if (crd_gn_apply(gx, 0, rc.C, 0, 8, rc.P, rc.C, gst, 1, gga, gbe, 1, nullptr, gy, 0, rc.C, 0, (crd_stream_t)s1)) { printf("gn_apply: %s\n", crd_last_error()); exit(1); }
      CK(hipStreamEndCapture(s1, &gg));
      CK(hipGraphInstantiate(&rc.ge, gg, nullptr, nullptr, 0));
      rc.alone = time_graph(rc.ge, [] {});
      printf("chain of 100 crd_gn_apply(+GELU) 8 x %d x %d alone: %.3f ms (%.2f us per kernel)\n", rc.P, rc.C, rc.alone, rc.alone * 10.f);
    }
    for (int cap : {51, 32, 12}) {
      crd_wgrad_desc d = {};
      d.x = x; d.x_ld = Cin; d.B = B; d.IH = H; d.IW = W; d.Cin = Cin; d.dy = dy; d.dy_ld = Cout; d.OH = H; d.OW = W; d.Cout = Cout;
      d.KH = d.KW = 3; d.stride = 1; d.pad = 1; d.dw = dwbuf; d.dw_partials = parts; d.dw_partial_capacity = cap;
      const int S = crd_conv_wgrad_splits(&d);
      float t_alone = 0.f;
      { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, s2)); for (int i = 0; i < 3; ++i) crd_conv_wgrad(&d, (crd_stream_t)s2); CK(hipEventRecord(e1, s2));
        CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&t_alone, e0, e1)); }
      const float ms = chain_ms([&] { for (int i = 0; i < 3; ++i) if (crd_conv_wgrad(&d, (crd_stream_t)s2)) { printf("wgrad: %s\n", crd_last_error()); exit(1); } });
      printf("next to 3 x crd_conv_wgrad 3x3 304->128 (%d splits x 5 chunks = %d WGs, %.0f us each alone): chain %.3f ms (%.2f us per kernel, +%.2f)\n",
             S, S * 5, t_alone * 1e3f / 3, ms, ms * 1e3f / NCHAIN, (ms - alone) * 1e3f / NCHAIN);
      for (RealChain& rc : rcs) {
        const float m2 = time_graph(rc.ge, [&] { for (int i = 0; i < 6; ++i) crd_conv_wgrad(&d, (crd_stream_t)s2); });
        printf("    chain of 100 crd_gn_apply 8 x %d x %d next to it: %.3f ms (%.2f us per kernel, +%.2f)\n", rc.P, rc.C, m2, m2 * 10.f, (m2 - rc.alone) * 10.f);
      }
    }
  }
#endif
  return 0;
}
