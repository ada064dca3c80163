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
extern "C" int crd_version(void) { return 1; }
extern "C" const char* crd_arch(void) { return "gfx950"; }

// ---- sticky non-finite indicator of the fixed-point sums (common.h: to_fx) ----
static int (*g_nf_readers[64])(int);
static int g_nf_count = 0;
void crd_register_nonfinite_reader(int (*reader)(int)) {
  if (g_nf_count < 64) g_nf_readers[g_nf_count++] = reader;
}
extern "C" int crd_nonfinite_status(int32_t reset) {
  int any = 0;
  for (int i = 0; i < g_nf_count; ++i) {
    const int v = g_nf_readers[i](reset);
    if (v < 0) { crd_set_error("crd_nonfinite_status: cannot read the device flag"); return CRD_E_LAUNCH; }
    any |= v;
  }
  return any ? 1 : 0;
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
