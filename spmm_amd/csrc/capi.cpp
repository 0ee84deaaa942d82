// Error reporting + version for the C ABI (include/spmm_hip.h).
#include "../../include/spmm_hip.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

extern "C" void spmm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* spmm_last_error(void) { return g_err; }
extern "C" int spmm_version(void) { return 100; }
