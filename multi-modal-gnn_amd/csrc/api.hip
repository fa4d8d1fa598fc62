// Error plumbing + version of libmmgnn's C ABI (include/mmgnn.h).
#include "common.h"

static thread_local char g_err[512] = "";

void mmg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mmg_version(void) { return 100; }   // 0.1.0
extern "C" const char* mmg_last_error(void) { return g_err; }
