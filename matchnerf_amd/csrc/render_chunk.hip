// a7 — one render chunk: MatchNeRF.render (/root/reference/models/matchnerf.py:88-143)
// = cost volume (K1+K2) -> conditioning vectors in `workspace` -> decoder + compositing (K3-K5).
// Staged form: the hand-off is cond_stride floats per sample through HBM (96 B at 3 views,
// against 24.7 KB of gathered features per sample), both launches on the caller's stream.
#include "common.hpp"

extern "C" int64_t mnerf_render_workspace_bytes(int32_t n_rays, int32_t n_samples,
                                                int32_t cond_stride) {
  if (n_rays < 0 || n_samples < 1 || cond_stride < 1) return -1;
  return (int64_t)n_rays * n_samples * cond_stride * (int64_t)sizeof(float);
}

extern "C" int mnerf_render_chunk(const mnerf_scene* scene, const mnerf_decoder* dec,
                                  const mnerf_rays* rays, void* workspace, float* rgb,
                                  float* depth, float* opacity, void* stream) {
  MNERF_REQUIRE(scene && dec && rays, MNERF_E_NULL, "mnerf_render_chunk: NULL argument struct");
  MNERF_REQUIRE(workspace, MNERF_E_NULL, "mnerf_render_chunk: workspace is NULL");
  MNERF_REQUIRE(mnerf_aligned16(workspace), MNERF_E_ALIGN, "mnerf_render_chunk: workspace not 16B aligned");
  MNERF_REQUIRE(dec->n_views == scene->n_views, MNERF_E_RANGE,
                "mnerf_render_chunk: decoder packed for %d views, scene has %d", dec->n_views,
                scene->n_views);
  const int sumG = scene->n_group[0] + (scene->n_scales > 1 ? scene->n_group[1] : 0);
  MNERF_REQUIRE(dec->cond_dim == sumG + 4 * scene->n_views, MNERF_E_RANGE,
                "mnerf_render_chunk: cond_dim=%d != sum(cos_n_group)+4V=%d", dec->cond_dim,
                sumG + 4 * scene->n_views);
  float* cond = (float*)workspace;
  int rc = mnerf_cost_volume(scene, rays, dec->cond_stride, cond, stream);
  if (rc) return rc;
  return mnerf_decoder_chunk(dec, &scene->views[0], rays, cond, rgb, depth, opacity, nullptr,
                             nullptr, stream);
}
