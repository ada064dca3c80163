// Launch-overhead probe: N dependent tiny kernels on one stream, eager vs captured hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_launch.hip -o tools/probe_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_tiny(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
    float* p; CK(hipMalloc(&p, 1 << 24)); CK(hipMemset(p, 0, 1 << 24));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int n : {256, 1 << 16, 1 << 20}) {
        const int N = 2000; int blocks = (n + 255) / 256;
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, p, n);
            auto t1 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("n=%8d eager: %.2f us/kernel GPU, host issue %.2f us/kernel\n", n, ms * 1e3 / N,
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, p, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("n=%8d graph: %.2f us/kernel\n", n, ms * 1e3 / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
