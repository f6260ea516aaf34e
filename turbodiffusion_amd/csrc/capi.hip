// C-ABI plumbing shared by every entry point: version + thread-local error string.
#include "td_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void td_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int td_abi_version(void) { return TD_ABI_VERSION; }
extern "C" const char* td_last_error(void) { return g_err; }
