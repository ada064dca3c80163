// Error reporting and version entry points of libcamradepth_hip.so.
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void crd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* crd_last_error(void) { return g_err; }
extern "C" int crd_version(void) { return CRD_ABI_VERSION; }
extern "C" const char* crd_arch(void) { return "gfx950"; }

// ---- sticky non-finite indicator of the fixed-point sums (common.h: to_fx) ----
// One flag per translation unit (a static __device__ word: no relocatable device code needed); their device addresses are gathered
// once PER DEVICE, and a status query is ONE 64-thread launch on the CALLER's stream + ONE 4-byte asynchronous copy + a wait for that
// stream (round 6, ADVICE r5: the legacy null stream does not order against non-blocking streams, so a query could read and reset the
// flag before the kernels that set it had run), whatever the number of translation units.
constexpr int NF_MAX = 64;
static void* (*g_nf_addr_fns[NF_MAX])();
static int g_nf_count = 0;
static bool g_nf_overflow = false;
void crd_register_nonfinite_flag(void* (*addr_of_flag)()) {
  if (g_nf_count < NF_MAX) g_nf_addr_fns[g_nf_count++] = addr_of_flag;
  else g_nf_overflow = true;                       // reported by crd_nonfinite_status: never silently dropped
}
struct NfPtrs { int* p[NF_MAX]; };
static __device__ int g_nf_any;
__global__ __launch_bounds__(64) void k_nf_gather(NfPtrs ptrs, int n, int reset) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  if ((int)threadIdx.x < n && *ptrs.p[threadIdx.x]) {
    any = 1;
    if (reset) *ptrs.p[threadIdx.x] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) g_nf_any = any;
}
extern "C" int crd_nonfinite_status(int32_t reset, crd_stream_t stream) {
  // symbol addresses are per device: one resolved table per device ordinal
  constexpr int MAX_DEV = 16;
  static NfPtrs ptrs[MAX_DEV];
  static int* any_addr[MAX_DEV];
  static bool resolved[MAX_DEV];
  if (g_nf_overflow) { crd_set_error("crd_nonfinite_status: more than %d translation units registered a flag", NF_MAX); return CRD_E_LAUNCH; }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) { crd_set_error("crd_nonfinite_status: no current device"); return CRD_E_LAUNCH; }
  if (!resolved[dev]) {
    for (int i = 0; i < g_nf_count; ++i) {
      ptrs[dev].p[i] = reinterpret_cast<int*>(g_nf_addr_fns[i]());
      if (!ptrs[dev].p[i]) { crd_set_error("crd_nonfinite_status: cannot resolve the device flag of translation unit %d", i); return CRD_E_LAUNCH; }
    }
    void* a = nullptr;
    if (hipGetSymbolAddress(&a, HIP_SYMBOL(g_nf_any)) != hipSuccess || !a) { crd_set_error("crd_nonfinite_status: cannot resolve the result word"); return CRD_E_LAUNCH; }
    any_addr[dev] = reinterpret_cast<int*>(a);
    resolved[dev] = true;
  }
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_nf_gather, dim3(1), dim3(64), 0, st, ptrs[dev], g_nf_count, reset ? 1 : 0);
  if (reset == 2) {                                 // clear only: asynchronous, nothing read back
    if (hipGetLastError() != hipSuccess) { crd_set_error("crd_nonfinite_status: launch failed"); return CRD_E_LAUNCH; }
    return 0;
  }
  int v = 0;
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&v, any_addr[dev], sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    crd_set_error("crd_nonfinite_status: cannot read the device flag");
    return CRD_E_LAUNCH;
  }
  return v ? 1 : 0;
}

// ---- refused dynamic-LDS reservations (common.h: crd_reserve_lds) ----
static thread_local char g_attr_err[256] = "";
void crd_note_attr_failure(const char* kernel, int bytes, int hip_err) {
  snprintf(g_attr_err, sizeof(g_attr_err), "cannot reserve %d bytes of dynamic LDS for %s (hip error %d: %s)", bytes, kernel, hip_err,
           hipGetErrorString((hipError_t)hip_err));
}
int crd_report_attr_failure(const char* entry_point) {
  if (g_attr_err[0] == 0) return 0;
  crd_set_error("%s: %s", entry_point, g_attr_err);
  g_attr_err[0] = 0;
  return 1;
}
