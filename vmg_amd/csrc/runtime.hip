// Error reporting and library identification for libvmg_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vmg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vmg_last_error(void) { return g_err; }
extern "C" int vmg_version(void) { return 100; }
extern "C" int vmg_max_lds_bytes(void) { return 160 * 1024; }
