// a7 — one render chunk: MatchNeRF.render (/root/reference/models/matchnerf.py:88-143)
// = cost volume (K1+K2) -> conditioning vectors -> decoder + compositing (K3-K5).
// Two forms, same results bit for bit:
//   staged  mnerf_cost_volume -> [rays*S, cond_stride] rows in `workspace` (HBM) -> mnerf_decoder_chunk, two launches
//           on the caller's stream.  The default: measured 2x FASTER than the fused form on MI355X —
//           the register-quad walk of the cost volume needs ~16 waves per CU to hide its load latency, which the
//           stand-alone kernel has and four producer waves inside a 256-VGPR MFMA workgroup do not;
//   fused   ONE launch of the ray-chunk kernel (decoder.hip, CVF = 1): every workgroup produces the conditioning rows
//           of its own tile in LDS and consumes them there; `workspace` is not touched.  mnerf_render_chunk_fused,
//           or mnerf_render_chunk with MNERF_RENDER_FUSED=1 in the environment at load time, wherever the
//           configuration fits (split-fp16 stream, S <= 128, <= 5 views: mnerf_render_chunk_is_fused).
#include "common.hpp"

extern "C" int64_t mnerf_render_workspace_bytes(int32_t n_rays, int32_t n_samples,
                                                int32_t cond_stride) {
  if (n_rays < 0 || n_samples < 1 || cond_stride < 1) return -1;
  return (int64_t)n_rays * n_samples * cond_stride * (int64_t)sizeof(float);
}

static int check_render_args(const mnerf_scene* scene, const mnerf_decoder* dec, const mnerf_rays* rays) {
  MNERF_REQUIRE(scene && dec && rays, MNERF_E_NULL, "mnerf_render_chunk: NULL argument struct");
  MNERF_REQUIRE(dec->n_views == scene->n_views, MNERF_E_RANGE,
                "mnerf_render_chunk: decoder packed for %d views, scene has %d", dec->n_views,
                scene->n_views);
  const int sumG = scene->n_group[0] + (scene->n_scales > 1 ? scene->n_group[1] : 0);
  MNERF_REQUIRE(dec->cond_dim == sumG + 4 * scene->n_views, MNERF_E_RANGE,
                "mnerf_render_chunk: cond_dim=%d != sum(cos_n_group)+4V=%d", dec->cond_dim,
                sumG + 4 * scene->n_views);
  return MNERF_OK;
}

extern "C" int32_t mnerf_render_chunk_is_fused(const mnerf_scene* scene, const mnerf_decoder* dec,
                                               const mnerf_rays* rays) {
  if (check_render_args(scene, dec, rays)) return 0;
  return mnerf_fused_render_applies(scene, dec, rays) ? 1 : 0;
}

extern "C" int32_t mnerf_render_takes_pose_table(const mnerf_scene* scene, const mnerf_decoder* dec, int32_t n_samples,
                                                 int32_t rays_per_pose) {
  if (!scene || !dec || n_samples < 1 || rays_per_pose <= 0 || rays_per_pose % 64) return 0;
  if (scene->n_views < 2 || scene->n_views > MNERF_MAX_VIEWS || scene->n_scales < 1 || scene->n_scales > 2) return 0;
  return mnerf_cost_volume_takes_pose_table(scene) && mnerf_decoder_takes_pose_table(dec, n_samples) ? 1 : 0;
}

extern "C" int mnerf_render_chunk_fused(const mnerf_scene* scene, const mnerf_decoder* dec,
                                        const mnerf_rays* rays, float* rgb, float* depth,
                                        float* opacity, void* stream) {
  int rc = check_render_args(scene, dec, rays);
  if (rc) return rc;
  MNERF_REQUIRE(rgb && depth && opacity, MNERF_E_NULL, "mnerf_render_chunk_fused: NULL output buffer");
  MNERF_REQUIRE(mnerf_fused_render_applies(scene, dec, rays), MNERF_E_UNSUPPORTED,
                "mnerf_render_chunk_fused: the one-launch form needs the split-fp16 stream, S <= 128, <= 32 conditioning "
                "inputs and cos_n_group entries >= 2 (use mnerf_render_chunk)");
  rc = mnerf_scene_check(scene, rays, "mnerf_render_chunk_fused");
  if (rc) return rc;
  return mnerf_fused_render_launch(scene, dec, rays, rgb, depth, opacity, stream);
}

extern "C" int mnerf_render_chunk(const mnerf_scene* scene, const mnerf_decoder* dec,
                                  const mnerf_rays* rays, void* workspace, float* rgb,
                                  float* depth, float* opacity, void* stream) {
  int rc = check_render_args(scene, dec, rays);
  if (rc) return rc;
  MNERF_REQUIRE(rgb && depth && opacity, MNERF_E_NULL, "mnerf_render_chunk: NULL output buffer");
  if (mnerf_tune().render_fused && mnerf_fused_render_applies(scene, dec, rays))
    return mnerf_render_chunk_fused(scene, dec, rays, rgb, depth, opacity, stream);
  MNERF_REQUIRE(workspace, MNERF_E_NULL, "mnerf_render_chunk: workspace is NULL");
  MNERF_REQUIRE(mnerf_aligned16(workspace), MNERF_E_ALIGN, "mnerf_render_chunk: workspace not 16B aligned");
  float* cond = (float*)workspace;
  rc = mnerf_cost_volume(scene, rays, dec->cond_stride, cond, stream);
  if (rc) return rc;
  return mnerf_decoder_chunk(dec, &scene->views[0], rays, cond, rgb, depth, opacity, nullptr,
                             nullptr, stream);
}
