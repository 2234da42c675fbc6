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

// sizeof() of the by-value argument structs as compiled into the library, so that a binding
// in another language can verify its own struct mirrors (0 view, 1 rays, 2 scene, 3 decoder).
extern "C" int64_t mnerf_struct_size(int32_t which) {
  switch (which) {
    case 0: return (int64_t)sizeof(mnerf_view);
    case 1: return (int64_t)sizeof(mnerf_rays);
    case 2: return (int64_t)sizeof(mnerf_scene);
    case 3: return (int64_t)sizeof(mnerf_decoder);
    default: return -1;
  }
}
