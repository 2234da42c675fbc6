// Backward kernels of the ray chunk: training through the HIP path
// (/root/reference/coach.py:215-243 back-propagates through the eager op chain).
//
//   mnerf_composite_backward     K5: d(rgb, depth, opacity) -> d(rgb_s, sigma)      (nerf.py:101-124)
//   mnerf_cost_volume_backward   K1+K2: d(cond) -> d(feature maps), scatter-add      (matchnerf.py:209-293)
//
// The conditional MLP + ray transformer in between (K3+K4) is re-evaluated with torch ops from the saved
// conditioning rows (matchnerf_amd/autograd.py): at training ray counts (rand_rays_train = 4096) it is a few
// milliseconds; its hand-written backward is the remaining part of that row.
#include <stdlib.h>

#include "cv_walk.hpp"

// ------------------------------------------------------------------ K5 backward, one wavefront per ray
// w_j = T_j a_j, a_j = 1 - exp(-sd_j), T_j = exp(-sum_{i<j} sd_i), sd_j = sigma_j * k_j (k_j = 1 or interval * |ray|)
//   out_rgb = sum w_j c_j + bg (1 - sum w_j), depth = sum w_j d_j, opacity = sum w_j
// With G_j = dL/dw_j = <g_rgb, c_j> + g_depth d_j + g_opacity - bg sum(g_rgb):
//   dL/dc_j  = w_j g_rgb
//   dL/dsd_j = G_j T_j (1 - a_j) - sum_{k>j} G_k w_k          (w_k depends on sd_j through T_k only)
__device__ __forceinline__ float wave_incl_scan_up(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    float t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

__global__ __launch_bounds__(256) void composite_backward_kernel(
    int n_rays, int S, const float* __restrict__ rgb_s, const float* __restrict__ sigma,
    const float* __restrict__ depth_s, const float* __restrict__ ray_len, int wo_interval, int setbg,
    const float* __restrict__ g_rgb, const float* __restrict__ g_depth, const float* __restrict__ g_opacity,
    float* __restrict__ g_rgb_s, float* __restrict__ g_sigma) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < n_rays; r += n_waves) {
    const size_t base = (size_t)r * S;
    const float rl = wo_interval ? 1.0f : ray_len[r];
    const float gr = g_rgb[(size_t)r * 3 + 0], gg = g_rgb[(size_t)r * 3 + 1], gb = g_rgb[(size_t)r * 3 + 2];
    const float gd = g_depth ? g_depth[r] : 0.0f;
    const float go = (g_opacity ? g_opacity[r] : 0.0f) - (setbg ? (gr + gg + gb) : 0.0f);
    const int n_blk = (S + 63) / 64;
    // pass 1 (front to back): total of G_k w_k over the ray, needed for the suffix sums
    float carry = 0.0f, total = 0.0f;
    for (int blk = 0; blk < n_blk; ++blk) {
      const int j = blk * 64 + lane;
      const bool ok = j < S;
      float sd = 0.0f, G = 0.0f;
      if (ok) {
        const float d = depth_s[base + j];
        float k = 1.0f;
        if (!wo_interval) k = ((j + 1 < S) ? (depth_s[base + j + 1] - d) : 1e10f) * rl;
        sd = sigma[base + j] * k;
        G = gr * rgb_s[(base + j) * 3 + 0] + gg * rgb_s[(base + j) * 3 + 1] + gb * rgb_s[(base + j) * 3 + 2] + gd * d + go;
      }
      float prev = __shfl_up(sd, 1, 64);
      if (lane == 0) prev = 0.0f;
      const float excl = carry + wave_incl_scan_up(prev, lane);
      const float w = ok ? expf(-excl) * (1.0f - expf(-sd)) : 0.0f;
      float gw = G * w;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) gw += __shfl_xor(gw, off, 64);
      total += gw;
      carry = __shfl(excl + sd, 63, 64);
    }
    // pass 2: gradients; suffix_j = total - prefix_inclusive_j
    carry = 0.0f;
    float gw_before = 0.0f;
    for (int blk = 0; blk < n_blk; ++blk) {
      const int j = blk * 64 + lane;
      const bool ok = j < S;
      float sd = 0.0f, G = 0.0f, k = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f;
      if (ok) {
        const float d = depth_s[base + j];
        if (!wo_interval) k = ((j + 1 < S) ? (depth_s[base + j + 1] - d) : 1e10f) * rl;
        sd = sigma[base + j] * k;
        cr = rgb_s[(base + j) * 3 + 0];
        cg = rgb_s[(base + j) * 3 + 1];
        cb = rgb_s[(base + j) * 3 + 2];
        G = gr * cr + gg * cg + gb * cb + gd * d + go;
      }
      float prev = __shfl_up(sd, 1, 64);
      if (lane == 0) prev = 0.0f;
      const float excl = carry + wave_incl_scan_up(prev, lane);
      const float T = expf(-excl), e = expf(-sd);
      const float w = ok ? T * (1.0f - e) : 0.0f;
      const float gw = G * w;
      const float incl = gw_before + wave_incl_scan_up(gw, lane);   // sum_{k<=j} G_k w_k
      if (ok) {
        g_rgb_s[(base + j) * 3 + 0] = w * gr;
        g_rgb_s[(base + j) * 3 + 1] = w * gg;
        g_rgb_s[(base + j) * 3 + 2] = w * gb;
        g_sigma[base + j] = (G * T * e - (total - incl)) * k;
      }
      gw_before = __shfl(incl, 63, 64);
      carry = __shfl(excl + sd, 63, 64);
    }
  }
}

extern "C" int mnerf_composite_backward(int32_t n_rays, int32_t n_samples, const float* rgb_s, const float* sigma,
                                        const float* depth_s, const float* ray_len, int32_t wo_render_interval,
                                        int32_t setbg_opaque, const float* g_rgb, const float* g_depth,
                                        const float* g_opacity, float* g_rgb_s, float* g_sigma, void* stream) {
  MNERF_REQUIRE(n_rays >= 0 && n_samples >= 1, MNERF_E_RANGE, "mnerf_composite_backward: n_rays=%d n_samples=%d", n_rays,
                n_samples);
  if (n_rays == 0) return MNERF_OK;
  MNERF_REQUIRE(rgb_s && sigma && depth_s && g_rgb && g_rgb_s && g_sigma, MNERF_E_NULL,
                "mnerf_composite_backward: NULL buffer");
  MNERF_REQUIRE(wo_render_interval || ray_len, MNERF_E_NULL,
                "mnerf_composite_backward: ray_len required when wo_render_interval == 0");
  int blocks = (n_rays + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(composite_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_rays, n_samples, rgb_s,
                     sigma, depth_s, ray_len, wo_render_interval, setbg_opaque, g_rgb, g_depth, g_opacity, g_rgb_s, g_sigma);
  return mnerf_check_launch("mnerf_composite_backward");
}

// ------------------------------------------------------------------ K1+K2 backward
// Only the cosine entries of a conditioning row depend on the feature maps (colours and masks come from the images
// and the geometry).  For one (sample, pair, scale, group) with interpolated features a, b (C/G channels):
//   cos = <a,b> / (na nb),  na = max(|a|, eps), nb = max(|b|, eps)
//   d cos / d a = b / (na nb) - [|a| > eps] cos a / |a|^2        (the clamp branch has no |a| term), same for b
// and a = sum_t w_t tap_t, so the four taps of each map receive w_t * d a (scatter-add: many samples share texels).
// Mapping: a slot of 16 lanes owns one sample; lane `sub` holds channels sub, 16 + sub, .., 112 + sub, so that every load and
// every float atomic of the slot covers 16 CONSECUTIVE channels = one 64-byte line (with 8 consecutive channels per lane the
// 16 lanes of an atomic instruction touched 8 lines, two dwords each: 10.9 ms for 1 024 rays x 64 samples, the float-atomic
// line rate of the L2; this form: see DESIGN.md).  A channel group (128 / G >= 16 channels) is then a set of register indices,
// the same in every lane, and its sums are 16-lane all-reductions.  The forward interpolation is recomputed (nothing but the
// rows' gradient is read).
template <int CTRL>
__device__ __forceinline__ float bwd_dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float slot16_sum(float v) {  // all-reduce over the 16 lanes of a slot (one DPP row)
  v = bwd_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = bwd_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = bwd_dpp_add<0x141>(v);  // row_half_mirror
  v = bwd_dpp_add<0x140>(v);  // row_mirror
  return v;
}

__global__ __launch_bounds__(256) void cost_volume_backward_kernel(mnerf_scene sc, mnerf_rays R, int cond_stride,
                                                                   const float* __restrict__ g_cond,
                                                                   float* __restrict__ g_feat0,
                                                                   float* __restrict__ g_feat1) {
  constexpr int CPL = 8, LPS = FEAT_C / CPL;
  const int sub = threadIdx.x % LPS;
  const int S = R.n_samples, V = sc.n_views;
  const int P = V * (V - 1) / 2;
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);
  const float inv_pairs = 1.0f / (float)P;
  const long long total = (long long)R.n_rays * S;
  const long long slots = ((long long)gridDim.x * blockDim.x) / LPS;
  const long long rounds = (total + slots - 1) / slots;  // every lane takes part in the DPP reductions of every round
  const long long slot0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / LPS;
  for (long long rnd = 0; rnd < rounds; ++rnd) {
    const long long s_raw = slot0 + rnd * slots;
    const bool live = s_raw < total;
    const long long s_idx = live ? s_raw : total - 1;
    const int ray = (int)(s_idx / S);
    const int j = (int)(s_idx - (long long)ray * S);
    const RayGeom g = make_ray(R, ray);
    const float d = sample_depth(R, ray, j);
    float px, py, pz;
    ray_point(g, d, px, py, pz);
    const float* grow = g_cond + (size_t)s_idx * cond_stride;
    int p = 0;
    for (int a = 0; a < V - 1; ++a) {
      float ua, va, za;
      project(sc.views[a], px, py, pz, wm1, hm1, ua, va, za);
      for (int b = a + 1; b < V; ++b, ++p) {
        float ub, vb, zb;
        project(sc.views[b], px, py, pz, wm1, hm1, ub, vb, zb);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s >= sc.n_scales) break;
          const int fh = sc.fh[s], fw = sc.fw[s];
          const size_t map_elems = (size_t)fh * fw * FEAT_C;
          const float* m0 = sc.feat[s] + (size_t)(2 * p) * map_elems + sub;
          const float* m1 = m0 + map_elems;
          float* gm0 = (s ? g_feat1 : g_feat0) + (size_t)(2 * p) * map_elems + sub;
          float* gm1 = gm0 + map_elems;
          const Bilin ba = bilin_setup(ua, va, fh, fw), bb = bilin_setup(ub, vb, fh, fw);
          float fa[CPL], fb[CPL];
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const int ch = c * LPS;  // (+ sub, folded into the base pointers)
            fa[c] = m0[(size_t)ba.o00 * FEAT_C + ch] * ba.w00 + m0[(size_t)ba.o01 * FEAT_C + ch] * ba.w01 +
                    m0[(size_t)ba.o10 * FEAT_C + ch] * ba.w10 + m0[(size_t)ba.o11 * FEAT_C + ch] * ba.w11;
            fb[c] = m1[(size_t)bb.o00 * FEAT_C + ch] * bb.w00 + m1[(size_t)bb.o01 * FEAT_C + ch] * bb.w01 +
                    m1[(size_t)bb.o10 * FEAT_C + ch] * bb.w10 + m1[(size_t)bb.o11 * FEAT_C + ch] * bb.w11;
          }
          const int G = sc.n_group[s];
          const int cpg = CPL / G;  // register indices per channel group (G in {1, 2, 4, 8})
          const float* gsrc = grow + (s ? sc.n_group[0] : 0);
#pragma unroll
          for (int gi = 0; gi < CPL; ++gi) {
            if (gi >= G) break;
            float dot = 0.f, na2 = 0.f, nb2 = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
              if (c / cpg == gi) {
                dot += fa[c] * fb[c];
                na2 += fa[c] * fa[c];
                nb2 += fb[c] * fb[c];
              }
            dot = slot16_sum(dot);
            na2 = slot16_sum(na2);
            nb2 = slot16_sum(nb2);
            const float ra = sqrtf(na2), rb = sqrtf(nb2);
            const float na = fmaxf(ra, 1e-8f), nb = fmaxf(rb, 1e-8f);
            const float inv = 1.0f / (na * nb);
            const float cosv = dot * inv;
            const float gcos = gsrc[gi] * inv_pairs;
            const float ka = ra > 1e-8f ? cosv / na2 : 0.0f, kb = rb > 1e-8f ? cosv / nb2 : 0.0f;
            if (live) {
#pragma unroll
              for (int c = 0; c < CPL; ++c)
                if (c / cpg == gi) {
                  const int ch = c * LPS;
                  const float da = gcos * (fb[c] * inv - ka * fa[c]);
                  const float db = gcos * (fa[c] * inv - kb * fb[c]);
                  atomicAdd(gm0 + (size_t)ba.o00 * FEAT_C + ch, da * ba.w00);
                  atomicAdd(gm0 + (size_t)ba.o01 * FEAT_C + ch, da * ba.w01);
                  atomicAdd(gm0 + (size_t)ba.o10 * FEAT_C + ch, da * ba.w10);
                  atomicAdd(gm0 + (size_t)ba.o11 * FEAT_C + ch, da * ba.w11);
                  atomicAdd(gm1 + (size_t)bb.o00 * FEAT_C + ch, db * bb.w00);
                  atomicAdd(gm1 + (size_t)bb.o01 * FEAT_C + ch, db * bb.w01);
                  atomicAdd(gm1 + (size_t)bb.o10 * FEAT_C + ch, db * bb.w10);
                  atomicAdd(gm1 + (size_t)bb.o11 * FEAT_C + ch, db * bb.w11);
                }
            }
          }
        }
      }
    }
  }
}

// The same gradients with the scatter-adds of CONSECUTIVE SAMPLES OF A RAY merged (round 4): a 16-lane slot owns one (ray, pair,
// scale) and walks the ray's samples; the gradient of the four texels under the current sample accumulates in registers for as
// long as the walk stays in that texel cell (2-4 samples at the 1/8 scale, 1-2 at 1/4) and goes out as one atomic per texel and
// channel when it leaves the cell, instead of one per sample; the taps of a cell are loaded once per visit, not once per sample.
// Same channel mapping as above (an atomic instruction of a slot covers one 64-byte line), same arithmetic per sample; what
// differs is the order in which contributions reach a texel (they did not have a fixed order before either: float atomics).
#ifndef CVB_WALK_WAVES
#define CVB_WALK_WAVES 2
#endif
__global__ __launch_bounds__(256, CVB_WALK_WAVES) void cost_volume_backward_walk_kernel(mnerf_scene sc, mnerf_rays R, int cond_stride,
                                                                        const float* __restrict__ g_cond,
                                                                        float* __restrict__ g_feat0,
                                                                        float* __restrict__ g_feat1, int n_seg) {
  // n_seg (round 6): a ray's samples in n_seg consecutive segments, one slot each.  A training batch is 1 024 rays: 6 144 (ray,
  // pair, scale) items = 1.5 workgroups per CU, each a serial walk over all samples; with segments of >= 8 samples the launch
  // fills the chip (a cell that straddles a segment boundary goes out twice - the atomics were unordered before, too).
  constexpr int CPL = 8, LPS = FEAT_C / CPL;
  const int sub = threadIdx.x % LPS;
  const int S = R.n_samples, V = sc.n_views, NS = sc.n_scales;
  const int P = V * (V - 1) / 2;
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);
  const float inv_pairs = 1.0f / (float)P;
  const long long total = (long long)R.n_rays * P * NS * n_seg;
  const long long slots = ((long long)gridDim.x * blockDim.x) / LPS;
  const long long rounds = (total + slots - 1) / slots;  // every lane takes part in the DPP reductions of every round
  const long long slot0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / LPS;
  for (long long rnd = 0; rnd < rounds; ++rnd) {
    const long long it_raw = slot0 + rnd * slots;
    const bool live = it_raw < total;
    const long long it = live ? it_raw : total - 1;
    const long long it_rps = it / n_seg;
    const int seg = (int)(it - it_rps * n_seg);
    const int j_begin = (int)(((long long)S * seg) / n_seg), j_end = (int)(((long long)S * (seg + 1)) / n_seg);
    const int ray = (int)(it_rps / (P * NS));
    const int rem = (int)(it_rps - (long long)ray * (P * NS));
    const int p = rem / NS, s = rem - p * NS;
    int va = 0, vb = p;  // pair p = (va, vb), va < vb, lexicographic order
    while (vb >= V - 1 - va) {
      vb -= V - 1 - va;
      ++va;
    }
    vb += va + 1;
    const int fh = sc.fh[s], fw = sc.fw[s];
    const size_t map_elems = (size_t)fh * fw * FEAT_C;
    const float* m0 = sc.feat[s] + (size_t)(2 * p) * map_elems + sub;
    const float* m1 = m0 + map_elems;
    float* gm0 = (s ? g_feat1 : g_feat0) + (size_t)(2 * p) * map_elems + sub;
    float* gm1 = gm0 + map_elems;
    const int G = sc.n_group[s];
    const int cpg = CPL / G;  // register indices per channel group (G in {1, 2, 4, 8})
    const int g_off = s ? sc.n_group[0] : 0;
    const RayGeom g = make_ray(R, ray);

    // the current texel cell of each side: its four texel indices, their taps and the gradient gathered for them so far
    int ca[4] = {-1, -1, -1, -1}, cb[4] = {-1, -1, -1, -1};
    float ta[4][CPL], tb[4][CPL], ga[4][CPL], gb[4][CPL];
    auto flush = [&](float* gm, const int (&cell)[4], float (&acc)[4][CPL]) {
      if (cell[0] < 0 || !live) return;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // (border cells repeat a texel: both of its entries go out, as two atomics on one address)
#pragma unroll
        for (int c = 0; c < CPL; ++c) atomicAdd(gm + (size_t)cell[t] * FEAT_C + c * LPS, acc[t][c]);
      }
    };
    auto enter = [&](const float* m, const Bilin& b, int (&cell)[4], float (&taps)[4][CPL], float (&acc)[4][CPL]) {
      cell[0] = b.o00, cell[1] = b.o01, cell[2] = b.o10, cell[3] = b.o11;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          taps[t][c] = m[(size_t)cell[t] * FEAT_C + c * LPS];
          acc[t][c] = 0.0f;
        }
    };
    for (int j = j_begin; j < j_end; ++j) {
      const float d = sample_depth(R, ray, j);
      float px, py, pz;
      ray_point(g, d, px, py, pz);
      float ua, va_, za, ub, vb_, zb;
      project(sc.views[va], px, py, pz, wm1, hm1, ua, va_, za);
      project(sc.views[vb], px, py, pz, wm1, hm1, ub, vb_, zb);
      const Bilin ba = bilin_setup(ua, va_, fh, fw), bb = bilin_setup(ub, vb_, fh, fw);
      if (ba.o00 != ca[0] || ba.o11 != ca[3]) {
        flush(gm0, ca, ga);
        enter(m0, ba, ca, ta, ga);
      }
      if (bb.o00 != cb[0] || bb.o11 != cb[3]) {
        flush(gm1, cb, gb);
        enter(m1, bb, cb, tb, gb);
      }
      float fa[CPL], fb[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        fa[c] = ta[0][c] * ba.w00 + ta[1][c] * ba.w01 + ta[2][c] * ba.w10 + ta[3][c] * ba.w11;
        fb[c] = tb[0][c] * bb.w00 + tb[1][c] * bb.w01 + tb[2][c] * bb.w10 + tb[3][c] * bb.w11;
      }
      const float* gsrc = g_cond + ((size_t)ray * S + j) * cond_stride + g_off;
#pragma unroll
      for (int gi = 0; gi < CPL; ++gi) {
        if (gi >= G) break;
        float dot = 0.f, na2 = 0.f, nb2 = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          if (c / cpg == gi) {
            dot += fa[c] * fb[c];
            na2 += fa[c] * fa[c];
            nb2 += fb[c] * fb[c];
          }
        dot = slot16_sum(dot);
        na2 = slot16_sum(na2);
        nb2 = slot16_sum(nb2);
        const float ra = sqrtf(na2), rb = sqrtf(nb2);
        const float na = fmaxf(ra, 1e-8f), nb = fmaxf(rb, 1e-8f);
        const float inv = 1.0f / (na * nb);
        const float cosv = dot * inv;
        const float gcos = gsrc[gi] * inv_pairs;
        const float ka = ra > 1e-8f ? cosv / na2 : 0.0f, kb = rb > 1e-8f ? cosv / nb2 : 0.0f;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          if (c / cpg == gi) {
            const float da = gcos * (fb[c] * inv - ka * fa[c]);
            const float db = gcos * (fa[c] * inv - kb * fb[c]);
            ga[0][c] += da * ba.w00, ga[1][c] += da * ba.w01, ga[2][c] += da * ba.w10, ga[3][c] += da * ba.w11;
            gb[0][c] += db * bb.w00, gb[1][c] += db * bb.w01, gb[2][c] += db * bb.w10, gb[3][c] += db * bb.w11;
          }
      }
    }
    flush(gm0, ca, ga);
    flush(gm1, cb, gb);
  }
}

extern "C" int mnerf_cost_volume_backward(const mnerf_scene* scene, const mnerf_rays* rays, int32_t cond_stride,
                                          const float* g_cond, float* g_feat0, float* g_feat1, void* stream) {
  int rc = mnerf_scene_check(scene, rays, "mnerf_cost_volume_backward");
  if (rc) return rc;
  MNERF_REQUIRE(g_cond && g_feat0 && (scene->n_scales < 2 || g_feat1), MNERF_E_NULL,
                "mnerf_cost_volume_backward: NULL buffer");
  MNERF_REQUIRE(!rays->pose_table, MNERF_E_UNSUPPORTED, "mnerf_cost_volume_backward: pose tables are inference-only");
  const int sumG = scene->n_group[0] + (scene->n_scales > 1 ? scene->n_group[1] : 0);
  MNERF_REQUIRE(cond_stride >= sumG + 4 * scene->n_views + 1, MNERF_E_RANGE,
                "mnerf_cost_volume_backward: cond_stride=%d < cond_dim+1=%d", cond_stride, sumG + 4 * scene->n_views + 1);
  if (rays->n_rays == 0) return MNERF_OK;
  // MNERF_CV_BWD_WALK (environment, read once; default 1): the form that merges the scatter-adds along a ray
  static const int walk = [] {
    const char* e = getenv("MNERF_CV_BWD_WALK");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  if (walk) {
    long long items = (long long)rays->n_rays * (scene->n_views * (scene->n_views - 1) / 2) * scene->n_scales;
    // segments per ray (MNERF_CV_BWD_SEGS overrides): enough items for ~6 workgroups per CU, segments of at least 8 samples
    static const int seg_env = [] {
      const char* e = getenv("MNERF_CV_BWD_SEGS");
      return e ? atoi(e) : 0;
    }();
    int n_seg = seg_env > 0 ? seg_env : (int)((16 * 1536 + items - 1) / items);  // (1 024 rays x 3 pairs x 2 scales: 4 segments; measured 1.07 / 1.06 / 0.93 / 0.97 ms for 1 / 2 / 4 / 8)
    if (n_seg > rays->n_samples / 8) n_seg = rays->n_samples / 8;
    if (n_seg < 1) n_seg = 1;
    items *= n_seg;
    long long blocks = (items + 15) / 16;  // 16 slots per 256-thread workgroup, one (ray, pair, scale, segment) each
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cost_volume_backward_walk_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *scene, *rays,
                       cond_stride, g_cond, g_feat0, g_feat1, n_seg);
    return mnerf_check_launch("mnerf_cost_volume_backward");
  }
  const long long total = (long long)rays->n_rays * rays->n_samples;
  long long blocks = (total + 15) / 16;  // 16 sample slots per 256-thread workgroup
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cost_volume_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *scene, *rays,
                     cond_stride, g_cond, g_feat0, g_feat1);
  return mnerf_check_launch("mnerf_cost_volume_backward");
}
