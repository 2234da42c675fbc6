// libmnerf_hip.so — error channel and ABI version (see include/mnerf.h).
#include <string.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "common.hpp"

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
    case 4: return (int64_t)sizeof(mnerf_encoder_layer);
    case 5: return (int64_t)sizeof(mnerf_conv);
    case 6: return (int64_t)sizeof(mnerf_decoder_train);
    case 7: return (int64_t)sizeof(mnerf_encoder_layer_train);
    default: return -1;
  }
}

// ---- debug / tuning knobs: the environment is read ONCE, when the library is loaded (a static initialiser), into a
// table that launches only read (mnerf_debug_set_knob below is the one writer, for tests).  Nothing in a launch path calls getenv or keeps mutable state; the only
// per-process bookkeeping left is "has this kernel's LDS attribute been set on this device" (mnerf_once_per_device).
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

static mnerf_tuning read_tuning() {
  mnerf_tuning t;
  t.decoder_grid = env_int("MNERF_DECODER_GRID", 512);          // persistent: 2 workgroups per CU x 256 CUs
  t.decoder_stagger = env_int("MNERF_DECODER_STAGGER", 16);     // ~130k cycles ~ half a tile
  t.decoder_stagger_mode = env_int("MNERF_DECODER_STAGGER_MODE", 0);
  t.cv_variant = env_int("MNERF_CV_VARIANT", 3);  // 3 / 4 = segment walk with 16 / 8 lanes per sample; 5 = texel tiles in LDS (slower, kept: cost_volume.hip); 0 = plain
  t.cv_mm = env_int("MNERF_CV_MM", 1);            // matrix form of the cost volume where it applies (cost_volume_mm.hip)
  t.cv_mm_spw = env_int("MNERF_CV_MM_SPW", 4);
  t.cv_uvpair = env_int("MNERF_CV_UVPAIR", -1);
  t.cv_pair_block = env_int("MNERF_CV_PAIR_BLOCK", -1);
  t.cv_grid = env_int("MNERF_CV_GRID", 0);        // 0 = the variant's default cap
  t.wa_min4 = env_int("MNERF_WA_MIN4", 200);      // 128-query workgroups once they (nearly) fill the 256 CUs
  t.wa_xcd = env_int("MNERF_WA_XCD", 1);          // query blocks of a window share an XCD (its L2 holds the K / V images)
  t.render_fused = env_int("MNERF_RENDER_FUSED", 0);  // 1: mnerf_render_chunk takes the one-launch form where it applies
  t.decoder_pp = env_int("MNERF_DECODER_PP", 1);
  t.decoder_pp_grid = env_int("MNERF_DECODER_PP_GRID", 256);
  t.decoder_pp_max_s = env_int("MNERF_DECODER_PP_MAX_S", 256);
  return t;
}

// The table is read-only in production (filled from the environment at load).  The one writer is the test hook below; it and
// every reader go through g_tuning_lock, and a reader takes a COPY: a launch on another thread sees the table before or after a
// knob change, never a torn one (13 ints under an uncontended mutex per launch).
static mnerf_tuning g_tuning = read_tuning();
static std::mutex g_tuning_lock;
mnerf_tuning mnerf_tune() {
  std::lock_guard<std::mutex> hold(g_tuning_lock);
  return g_tuning;
}

// Test / diagnosis hook: change one knob of the table after load (what its environment variable would have set); the previous
// value comes back through *old_value, the return value is only the status.  Serialised with the launches' reads (above).
extern "C" int mnerf_debug_set_knob(const char* name, int value, int* old_value) {
  std::lock_guard<std::mutex> hold(g_tuning_lock);
  struct Knob { const char* name; int* slot; };
  const Knob knobs[] = {{"decoder_pp", &g_tuning.decoder_pp}, {"decoder_pp_grid", &g_tuning.decoder_pp_grid},
                        {"decoder_pp_max_s", &g_tuning.decoder_pp_max_s}, {"decoder_grid", &g_tuning.decoder_grid},
                        {"cv_variant", &g_tuning.cv_variant}, {"cv_mm", &g_tuning.cv_mm}, {"cv_mm_spw", &g_tuning.cv_mm_spw}, {"cv_uvpair", &g_tuning.cv_uvpair}, {"cv_pair_block", &g_tuning.cv_pair_block}, {"cv_grid", &g_tuning.cv_grid},
                        {"render_fused", &g_tuning.render_fused}};
  for (const Knob& k : knobs)
    if (name && strcmp(name, k.name) == 0) {
      if (old_value) *old_value = *k.slot;
      *k.slot = value;
      return MNERF_OK;
    }
  mnerf_set_error("mnerf_debug_set_knob: unknown knob '%s'", name ? name : "(null)");
  return MNERF_E_RANGE;
}

bool mnerf_once_per_device(std::atomic<unsigned long long>& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  return (mask.fetch_or(bit, std::memory_order_relaxed) & bit) == 0;
}
