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

static int g_tuning[TD_TUNE_COUNT] = {0};
int td_tuning(int key) { return (key >= 0 && key < TD_TUNE_COUNT) ? g_tuning[key] : 0; }
extern "C" int td_set_tuning(int key, int value) {
  TD_REQUIRE(key >= 0 && key < TD_TUNE_COUNT, TD_ERR_INVALID, "td_set_tuning: key %d", key);
  g_tuning[key] = value;
  return TD_OK;
}
