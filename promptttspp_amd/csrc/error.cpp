// Thread-local error message + version for the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ptpp.h"

static thread_local char g_err[512] = "";

void ptpp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ptpp_last_error(void) { return g_err; }
extern "C" int ptpp_version(void) { return 1; }
