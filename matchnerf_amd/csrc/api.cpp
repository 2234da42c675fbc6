// libmnerf_hip.so — error channel and ABI version (see include/mnerf.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mnerf.h"

static thread_local char g_err[512] = "";

void mnerf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mnerf_abi_version(void) { return MNERF_ABI_VERSION; }
extern "C" const char* mnerf_last_error(void) { return g_err; }
