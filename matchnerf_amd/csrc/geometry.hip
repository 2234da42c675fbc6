// a8-a10 as a stand-alone op: target rays -> depth samples -> world points -> (u,v,z) in one
// source view.  Replaces camera.get_center_and_ray (/root/reference/misc/camera.py:255-278),
// MatchNeRF.sample_depth (models/matchnerf.py:163-181), get_3D_points_from_depth
// (camera.py:281-286) and get_coord_ref_ndc (camera.py:351-379).  The fused kernels inline the
// same device helpers (common.hpp); this entry point exists so that the geometry can be
// checked on its own — it is bit-exact against the reference's CPU path.
#include "common.hpp"

__global__ __launch_bounds__(256) void ray_samples_kernel(mnerf_rays R, mnerf_view V,
                                                          float* __restrict__ pts,
                                                          float* __restrict__ ndc,
                                                          float* __restrict__ depth) {
  const long long total = (long long)R.n_rays * R.n_samples;
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ray = (int)(i / R.n_samples);
    const int j = (int)(i - (long long)ray * R.n_samples);
    const RayGeom g = make_ray(R, ray);
    const float d = sample_depth(R, ray, j);
    float px, py, pz;
    ray_point(g, d, px, py, pz);
    if (pts) {
      pts[i * 3 + 0] = px;
      pts[i * 3 + 1] = py;
      pts[i * 3 + 2] = pz;
    }
    if (depth) depth[i] = d;
    if (ndc) {
      float u, v, z;
      project(V, px, py, pz, wm1, hm1, u, v, z);
      ndc[i * 3 + 0] = u;
      ndc[i * 3 + 1] = v;
      ndc[i * 3 + 2] = z;
    }
  }
}

extern "C" int mnerf_ray_samples(const mnerf_rays* rays, const mnerf_view* view, float* pts,
                                 float* ndc, float* depth, void* stream) {
  MNERF_REQUIRE(rays, MNERF_E_NULL, "mnerf_ray_samples: rays is NULL");
  MNERF_REQUIRE(rays->n_rays >= 0 && rays->n_samples >= 1, MNERF_E_RANGE,
                "mnerf_ray_samples: n_rays=%d S=%d", rays->n_rays, rays->n_samples);
  MNERF_REQUIRE(!ndc || view, MNERF_E_NULL, "mnerf_ray_samples: view required for ndc output");
  MNERF_REQUIRE(!rays->pose_table && rays->rays_per_pose == 0, MNERF_E_UNSUPPORTED,
                "mnerf_ray_samples: pose tables are a rendering-launch feature (mnerf_cost_volume / mnerf_decoder_chunk)");
  if (rays->n_rays == 0) return MNERF_OK;
  mnerf_view v = {};
  if (view) v = *view;
  const long long total = (long long)rays->n_rays * rays->n_samples;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ray_samples_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     *rays, v, pts, ndc, depth);
  return mnerf_check_launch("mnerf_ray_samples");
}
