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
