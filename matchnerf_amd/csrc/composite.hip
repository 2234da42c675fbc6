// K5 — volume-rendering quadrature, one wavefront per ray.
//
// Replaces NeRF.composite (/root/reference/models/rfdecoder/nerf.py:101-124):
//   alpha_j = 1 - exp(-sd_j),  T_j = exp(-sum_{i<j} sd_i),  w_j = T_j alpha_j
//   rgb = sum w_j c_j, depth = sum w_j d_j, opacity = sum w_j  (+ white background)
// with sd_j = sigma_j (wo_render_interval) or sigma_j * (d_{j+1}-d_j) * |ray| (last 1e10).
//
// Mapping: lane = sample.  The exclusive prefix sum of sd is a wave64 scan built from
// __shfl_up steps (5 DPP row/bank shifts + 1 cross-row on gfx950); rays with S > 64 are
// walked in 64-sample blocks with a scalar carry.  HBM traffic is the compulsory
// 5 floats/sample in + 5 floats/ray out, so this stand-alone form is bandwidth-bound; in the
// fused ray-chunk kernel (decoder.hip) the same routine runs on LDS-resident samples.
#include "common.hpp"

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    float t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void composite_kernel(
    int n_rays, int S, const float* __restrict__ rgb_s, const float* __restrict__ sigma,
    const float* __restrict__ depth_s, const float* __restrict__ ray_len, int wo_interval,
    int setbg, float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ opacity,
    float* __restrict__ prob) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < n_rays; r += n_waves) {
    const size_t base = (size_t)r * S;
    const float rl = wo_interval ? 1.0f : ray_len[r];
    float carry = 0.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_o = 0.f;
    for (int j0 = 0; j0 < S; j0 += 64) {
      const int j = j0 + lane;
      const bool ok = j < S;
      float sd = 0.f, d = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
      if (ok) {
        sd = sigma[base + j];
        d = depth_s[base + j];
        cr = rgb_s[(base + j) * 3 + 0];
        cg = rgb_s[(base + j) * 3 + 1];
        cb = rgb_s[(base + j) * 3 + 2];
        if (!wo_interval) {
          float intv = (j + 1 < S) ? (depth_s[base + j + 1] - d) : 1e10f;
          sd = sd * (intv * rl);
        }
      }
      // exclusive prefix = inclusive scan of the values shifted down one lane (never
      // "inclusive - self": the last interval is 1e10 and would cancel the prefix away)
      float prev = __shfl_up(sd, 1, 64);
      if (lane == 0) prev = 0.f;
      const float excl = carry + wave_incl_scan(prev, lane);
      const float w = ok ? expf(-excl) * (1.0f - expf(-sd)) : 0.f;
      if (prob && ok) prob[base + j] = w;
      acc_r += w * cr;
      acc_g += w * cg;
      acc_b += w * cb;
      acc_d += w * d;
      acc_o += w;
      carry = __shfl(excl + sd, 63, 64);
    }
    acc_r = wave_sum(acc_r);
    acc_g = wave_sum(acc_g);
    acc_b = wave_sum(acc_b);
    acc_d = wave_sum(acc_d);
    acc_o = wave_sum(acc_o);
    if (lane == 0) {
      const float bg = setbg ? (1.0f - acc_o) : 0.f;
      rgb[(size_t)r * 3 + 0] = acc_r + bg;
      rgb[(size_t)r * 3 + 1] = acc_g + bg;
      rgb[(size_t)r * 3 + 2] = acc_b + bg;
      depth[r] = acc_d;
      opacity[r] = acc_o;
    }
  }
}

extern "C" int mnerf_composite(int32_t n_rays, int32_t n_samples, const float* rgb_s,
                               const float* sigma, const float* depth_s, const float* ray_len,
                               int32_t wo_render_interval, int32_t setbg_opaque, float* rgb,
                               float* depth, float* opacity, float* prob, void* stream) {
  MNERF_REQUIRE(n_rays >= 0 && n_samples >= 1, MNERF_E_RANGE,
                "mnerf_composite: n_rays=%d n_samples=%d", n_rays, n_samples);
  if (n_rays == 0) return MNERF_OK;  // empty chunk: nothing to read or write
  MNERF_REQUIRE(rgb_s && sigma && depth_s && rgb && depth && opacity, MNERF_E_NULL,
                "mnerf_composite: NULL buffer");
  MNERF_REQUIRE(wo_render_interval || ray_len, MNERF_E_NULL,
                "mnerf_composite: ray_len required when wo_render_interval == 0");
  int blocks = (n_rays + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(composite_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_rays,
                     n_samples, rgb_s, sigma, depth_s, ray_len, wo_render_interval, setbg_opaque,
                     rgb, depth, opacity, prob);
  return mnerf_check_launch("mnerf_composite");
}
