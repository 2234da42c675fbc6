// K1+K2 — epipolar feature sampling + group-wise cosine cost volume + colours + masks.
//
// Replaces MatchNeRF.query_cond_info (/root/reference/models/matchnerf.py:209-293):
// for every 3D sample, project into each source view (misc/camera.py:351-379), bilinear
// lookup (border, align_corners=True; gmflow/utils.py:133-134, matchnerf.py:245) of the
// pair-specific GMFlow features at two scales and of the RGB image, strict in-frustum mask
// (matchnerf.py:248-250), then per view pair the cosine similarity of G channel groups
// (matchnerf.py:256-275), averaged over pairs.  The reference materialises six
// [1,256,R,S] sampled-feature tensors (268 MB each at the DTU config); here nothing but the
// conditioning vector (cond_stride floats per sample) ever leaves registers.
//
// Data layout (owned by the build, include/mnerf.h): features are pair-major, channel-last
// [P][2][h][w][128] fp32, so one bilinear tap of one map is a contiguous 512 B run.
// Mapping: a "slot" of 8 lanes owns one sample; lane `sub` owns channels [16 sub, 16 sub+16)
// = 4 x 16-byte loads per tap, i.e. each tap is read by the slot as 4 fully used 128-byte
// lines.  8 slots per wavefront take 8 consecutive samples of a ray, whose projections are
// a fraction of a texel apart, so most taps of one wave instruction hit the same lines in
// the CU's vector L1; the maps themselves (78.6 MB fp32 at 512x640x3 views) stay resident in
// L2 / Infinity Cache.  Group reductions (dot, |a|^2, |b|^2 over 128/G channels) are in-lane
// for G=8 and 1-3 xor-shuffle steps inside the slot for G=4,2,1.
#include <stdlib.h>

#include "common.hpp"

#define FEAT_C MNERF_FEAT_CH

struct Bilin {
  int o00, o01, o10, o11;  // texel indices (y*w+x)
  float w00, w01, w10, w11;
};

// grid_sample(border, align_corners=True) coordinate handling for a map of size (h,w);
// u,v are the reference's [0,1]-normalised pixel coordinates (grid = 2u-1).
__device__ __forceinline__ Bilin bilin_setup(float u, float v, int h, int w) {
  float gx = u * 2.0f - 1.0f, gy = v * 2.0f - 1.0f;
  float x = ((gx + 1.0f) * 0.5f) * (float)(w - 1);
  float y = ((gy + 1.0f) * 0.5f) * (float)(h - 1);
  x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
  y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
  float x0f = floorf(x), y0f = floorf(y);
  float fx = x - x0f, fy = y - y0f;
  int x0 = (int)x0f, y0 = (int)y0f;
  int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
  Bilin b;
  b.o00 = y0 * w + x0;
  b.o01 = y0 * w + x1;
  b.o10 = y1 * w + x0;
  b.o11 = y1 * w + x1;
  b.w00 = (1.0f - fx) * (1.0f - fy);
  b.w01 = fx * (1.0f - fy);
  b.w10 = (1.0f - fx) * fy;
  b.w11 = fx * fy;
  return b;
}

// 16 interpolated channels of one map for this lane
__device__ __forceinline__ void sample16(const float* __restrict__ map, const Bilin& b, int sub,
                                         float (&out)[16]) {
  const float4* p00 = reinterpret_cast<const float4*>(map + (size_t)b.o00 * FEAT_C) + sub * 4;
  const float4* p01 = reinterpret_cast<const float4*>(map + (size_t)b.o01 * FEAT_C) + sub * 4;
  const float4* p10 = reinterpret_cast<const float4*>(map + (size_t)b.o10 * FEAT_C) + sub * 4;
  const float4* p11 = reinterpret_cast<const float4*>(map + (size_t)b.o11 * FEAT_C) + sub * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 a = p00[q], c = p01[q], d = p10[q], e = p11[q];
    out[q * 4 + 0] = a.x * b.w00 + c.x * b.w01 + d.x * b.w10 + e.x * b.w11;
    out[q * 4 + 1] = a.y * b.w00 + c.y * b.w01 + d.y * b.w10 + e.y * b.w11;
    out[q * 4 + 2] = a.z * b.w00 + c.z * b.w01 + d.z * b.w10 + e.z * b.w11;
    out[q * 4 + 3] = a.w * b.w00 + c.w * b.w01 + d.w * b.w10 + e.w * b.w11;
  }
}

template <int LANES_PER_GROUP>
__device__ __forceinline__ float slot_reduce(float v) {
#pragma unroll
  for (int m = 1; m < LANES_PER_GROUP; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ float group_reduce(float v, int lanes_per_group) {
  switch (lanes_per_group) {
    case 1: return v;
    case 2: return slot_reduce<2>(v);
    case 4: return slot_reduce<4>(v);
    case 8: return slot_reduce<8>(v);
    default: return slot_reduce<16>(v);
  }
}

#ifndef CV_WAVES_PER_SIMD
#define CV_WAVES_PER_SIMD 2
#endif
__global__ __launch_bounds__(256, CV_WAVES_PER_SIMD) void cost_volume_kernel(mnerf_scene sc, mnerf_rays R,
                                                          int cond_stride,
                                                          float* __restrict__ cond) {
  const int sub = threadIdx.x & 7;
  const int slot_in_wg = threadIdx.x >> 3;  // 32 sample slots per workgroup
  const int S = R.n_samples;
  const int V = sc.n_views;
  const int P = V * (V - 1) / 2;
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);
  const int sumG = sc.n_group[0] + (sc.n_scales > 1 ? sc.n_group[1] : 0);
  const float inv_pairs = 1.0f / (float)P;

  // Work mapping.  Samples are cut into contiguous chunks, one per workgroup, and chunk ids are
  // assigned XCD-major: the dispatcher places workgroup b on XCD b % 8 (observed, speed only),
  // so XCD x walks chunks [x*cpx, (x+1)*cpx) = one compact band of the image.  Its L2 then
  // holds just that band's epipolar texels instead of every resident workgroup sweeping the
  // whole frame (a grid-stride mapping fetched 7 GB/frame past L2 for 95 MB of maps).
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;  // bijective remap also when nwg % 8 != 0
  const int chunk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin;
  // A workgroup iteration covers 32 ADJACENT RAYS at ONE depth index (slot = ray), then steps
  // along the depth: neighbouring pixels project a fraction of a texel apart, so the 8 slots of
  // a wave mostly ask for the same 128-byte lines in one load instruction and the lines are
  // re-used again by the next depth steps (temporal L1 reuse).  One chunk = a run of 32-ray
  // blocks; all lanes of a wave iterate the same number of times (shuffles need full slots).
  const long long blocks_total = ((long long)R.n_rays + 31) / 32;   // 32-ray blocks
  const long long bpc = (blocks_total + nwg - 1) / nwg;             // blocks per chunk
  const long long b_begin = (long long)chunk * bpc;
  long long b_end = b_begin + bpc;
  if (b_end > blocks_total) b_end = blocks_total;

  for (long long it = b_begin * S; it < b_end * S; ++it) {
    const long long rb = it / S;
    const int j_it = (int)(it - rb * S);
    long long ray_ll = rb * 32 + slot_in_wg;
    const bool live = ray_ll < R.n_rays;
    if (!live) ray_ll = R.n_rays - 1;  // keep the lanes busy on a valid ray
    long long s_idx = ray_ll * S + j_it;
    const int ray = (int)(s_idx / S);
    const int j = (int)(s_idx - (long long)ray * S);
    const RayGeom g = make_ray(R, ray);
    const float d = sample_depth(R, ray, j);
    float px, py, pz;
    ray_point(g, d, px, py, pz);
    float* out = cond + (size_t)s_idx * cond_stride;

    // ---- pair-wise cosine cost volume
    float cos_acc[2] = {0.f, 0.f};
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
          const float* m0 = sc.feat[s] + (size_t)(2 * p) * map_elems;
          const float* m1 = m0 + map_elems;
          float fa[16], fb[16];
          sample16(m0, bilin_setup(ua, va, fh, fw), sub, fa);
          sample16(m1, bilin_setup(ub, vb, fh, fw), sub, fb);
          const int G = sc.n_group[s];
          const int lpg = 8 / G;  // lanes per channel group (G in {1,2,4,8})
          float dot = 0.f, na = 0.f, nb = 0.f;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            dot += fa[c] * fb[c];
            na += fa[c] * fa[c];
            nb += fb[c] * fb[c];
          }
          dot = group_reduce(dot, lpg);
          na = group_reduce(na, lpg);
          nb = group_reduce(nb, lpg);
          // CosineSimilarity: x1/max(|x1|,eps) . x2/max(|x2|,eps), eps = 1e-8
          const float da = fmaxf(sqrtf(na), 1e-8f), db = fmaxf(sqrtf(nb), 1e-8f);
          cos_acc[s] += dot / (da * db);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s >= sc.n_scales) break;
        const int G = sc.n_group[s];
        const int lpg = 8 / G;
        if ((sub % lpg) == 0) out[(s ? sc.n_group[0] : 0) + sub / lpg] = cos_acc[s] * inv_pairs;
      }
    }

    // ---- colours + visibility masks: lane `sub` takes views sub, sub+8, ...
    for (int v = sub; v < V; v += 8) {
      float u, w_, z;
      project(sc.views[v], px, py, pz, wm1, hm1, u, w_, z);
      const Bilin b = bilin_setup(u, w_, R.height, R.width);
      const float4* img = reinterpret_cast<const float4*>(sc.images) + (size_t)v * R.height * R.width;
      const float4 t00 = img[b.o00], t01 = img[b.o01], t10 = img[b.o10], t11 = img[b.o11];
      const float gx = u * 2.0f - 1.0f, gy = w_ * 2.0f - 1.0f;
      const float m = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;
      if (live) {
        out[sumG + 3 * v + 0] = t00.x * b.w00 + t01.x * b.w01 + t10.x * b.w10 + t11.x * b.w11;
        out[sumG + 3 * v + 1] = t00.y * b.w00 + t01.y * b.w01 + t10.y * b.w10 + t11.y * b.w11;
        out[sumG + 3 * v + 2] = t00.z * b.w00 + t01.z * b.w01 + t10.z * b.w10 + t11.z * b.w11;
        out[sumG + 3 * V + v] = m;
      }
    }
    if (live && sub == 0) {
      const int dc = sumG + 4 * V;
      out[dc] = 1.0f;  // constant input of the packed FiLM bias column
      for (int c = dc + 1; c < cond_stride; ++c) out[c] = 0.0f;
    }
  }
}

// ============================================================================ v2: segment walk
// Same arithmetic as cost_volume_kernel, different traversal.  PMC showed that kernel bound by
// the bytes the texture path delivers to registers (~18 TB/s of taps, 53 % of L1 peak), not by
// L2/HBM — and consecutive samples of a ray move only ~1/4 texel at 1/8 resolution (~1/2 at 1/4),
// so most of those bytes are the same texels again.  Here a slot (8 lanes) WALKS a segment of
// CVW_SEG consecutive samples of one ray; for each view pair and scale it keeps the current 2x2
// texel quad of both maps in registers (2 x 4 taps x 16 channels per lane) and reloads a quad
// only when its integer texel changes.  Projections are evaluated once per (sample, view) and
// parked in LDS; the per-sample cosine sums over pairs accumulate in LDS.
// Two instantiations: 16 channels per lane (8 lanes/sample, 2 waves/SIMD) and 8 channels per lane
// (16 lanes/sample, half the quad registers, 3 waves/SIMD — measured 7 % faster, the default).
#ifndef CVW_SEG
#define CVW_SEG 16
#endif
// CPL = channels per lane (16 -> 8 lanes per sample, 8 -> 16 lanes per sample)

template <int CPL>
struct QuadCache {
  float t00[CPL], t01[CPL], t10[CPL], t11[CPL];
  int o00, o01, o10, o11;
};

template <int CPL>
__device__ __forceinline__ void quad_load(QuadCache<CPL>& q, const float* __restrict__ map, const Bilin& b, int sub) {
  const float4* p00 = reinterpret_cast<const float4*>(map + (size_t)b.o00 * FEAT_C) + sub * (CPL / 4);
  const float4* p01 = reinterpret_cast<const float4*>(map + (size_t)b.o01 * FEAT_C) + sub * (CPL / 4);
  const float4* p10 = reinterpret_cast<const float4*>(map + (size_t)b.o10 * FEAT_C) + sub * (CPL / 4);
  const float4* p11 = reinterpret_cast<const float4*>(map + (size_t)b.o11 * FEAT_C) + sub * (CPL / 4);
#pragma unroll
  for (int k = 0; k < CPL / 4; ++k) {
    const float4 a = p00[k], c = p01[k], d = p10[k], e = p11[k];
    q.t00[4 * k] = a.x; q.t00[4 * k + 1] = a.y; q.t00[4 * k + 2] = a.z; q.t00[4 * k + 3] = a.w;
    q.t01[4 * k] = c.x; q.t01[4 * k + 1] = c.y; q.t01[4 * k + 2] = c.z; q.t01[4 * k + 3] = c.w;
    q.t10[4 * k] = d.x; q.t10[4 * k + 1] = d.y; q.t10[4 * k + 2] = d.z; q.t10[4 * k + 3] = d.w;
    q.t11[4 * k] = e.x; q.t11[4 * k + 1] = e.y; q.t11[4 * k + 2] = e.z; q.t11[4 * k + 3] = e.w;
  }
  q.o00 = b.o00;
  q.o01 = b.o01;
  q.o10 = b.o10;
  q.o11 = b.o11;
}

template <int CPL>
__device__ __forceinline__ bool quad_stale(const QuadCache<CPL>& q, const Bilin& b) {
  return (q.o00 != b.o00) | (q.o01 != b.o01) | (q.o10 != b.o10) | (q.o11 != b.o11);
}

template <int CPL>
__device__ __forceinline__ void quad_interp(const QuadCache<CPL>& q, const Bilin& b, float (&out)[CPL]) {
#pragma unroll
  for (int c = 0; c < CPL; ++c)  // same expression as sample16()
    out[c] = q.t00[c] * b.w00 + q.t01[c] * b.w01 + q.t10[c] * b.w10 + q.t11[c] * b.w11;
}

template <int CPL>
__global__ __launch_bounds__(256, (CPL == 16 ? 2 : 3)) void cost_volume_walk_kernel(mnerf_scene sc, mnerf_rays R,
                                                                  int cond_stride,
                                                                  float* __restrict__ cond) {
  constexpr int LPS = FEAT_C / CPL;     // lanes per sample slot (8 or 16)
  constexpr int NSLOT = 256 / LPS;      // ray slots per workgroup (32 or 16)
  constexpr int SPL = CVW_SEG / LPS > 0 ? CVW_SEG / LPS : 1;  // pass-1 samples per lane
  extern __shared__ __attribute__((aligned(16))) float cvw_smem[];
  const int V = sc.n_views;
  const int sub = threadIdx.x % LPS;
  const int slot = threadIdx.x / LPS;                               // NSLOT adjacent rays
  float* uv_lds = cvw_smem + (size_t)slot * CVW_SEG * V * 2;       // [seg sample][view][u,v]
  float* cs_lds = cvw_smem + NSLOT * CVW_SEG * V * 2 + slot * CVW_SEG * 16;  // [seg sample][<=16 cos sums]
  const int S = R.n_samples;
  const int P = V * (V - 1) / 2;
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);
  const int G0 = sc.n_group[0], G1 = sc.n_scales > 1 ? sc.n_group[1] : 0;
  const int sumG = G0 + G1;
  const float inv_pairs = 1.0f / (float)P;
  const int n_seg = (S + CVW_SEG - 1) / CVW_SEG;

  // XCD-major contiguous runs of 32-ray blocks (see cost_volume_kernel)
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int chunk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin;
  const long long blocks_total = ((long long)R.n_rays + NSLOT - 1) / NSLOT;
  const long long bpc = (blocks_total + nwg - 1) / nwg;
  const long long b_begin = (long long)chunk * bpc;
  long long b_end = b_begin + bpc;
  if (b_end > blocks_total) b_end = blocks_total;

  for (long long it = b_begin * n_seg; it < b_end * n_seg; ++it) {
    const long long rb = it / n_seg;
    const int j0 = (int)(it - rb * n_seg) * CVW_SEG;
    long long ray_ll = rb * NSLOT + slot;
    const bool ray_live = ray_ll < R.n_rays;
    if (!ray_live) ray_ll = R.n_rays - 1;
    const int ray = (int)ray_ll;
    const RayGeom g = make_ray(R, ray);

    // ---- pass 1: projections, colours, masks.  Lane `sub` takes segment samples sub, sub+LPS, ..
#pragma unroll
    for (int half = 0; half < SPL; ++half) {
      const int js = sub + LPS * half;
      if (js >= CVW_SEG) break;
      const int j = min(j0 + js, S - 1);
      const bool live = ray_live && (j0 + js < S);
      const float d = sample_depth(R, ray, j);
      float px, py, pz;
      ray_point(g, d, px, py, pz);
      float* out = cond + ((size_t)ray * S + j) * cond_stride;
      for (int v = 0; v < V; ++v) {
        float u, w_, z;
        project(sc.views[v], px, py, pz, wm1, hm1, u, w_, z);
        uv_lds[(js * V + v) * 2 + 0] = u;
        uv_lds[(js * V + v) * 2 + 1] = w_;
        const Bilin b = bilin_setup(u, w_, R.height, R.width);
        const float4* img = reinterpret_cast<const float4*>(sc.images) + (size_t)v * R.height * R.width;
        const float4 t00 = img[b.o00], t01 = img[b.o01], t10 = img[b.o10], t11 = img[b.o11];
        const float gx = u * 2.0f - 1.0f, gy = w_ * 2.0f - 1.0f;
        const float m = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;
        if (live) {
          out[sumG + 3 * v + 0] = t00.x * b.w00 + t01.x * b.w01 + t10.x * b.w10 + t11.x * b.w11;
          out[sumG + 3 * v + 1] = t00.y * b.w00 + t01.y * b.w01 + t10.y * b.w10 + t11.y * b.w11;
          out[sumG + 3 * v + 2] = t00.z * b.w00 + t01.z * b.w01 + t10.z * b.w10 + t11.z * b.w11;
          out[sumG + 3 * V + v] = m;
        }
      }
      if (live) {
        const int dc = sumG + 4 * V;
        out[dc] = 1.0f;
        for (int c = dc + 1; c < cond_stride; ++c) out[c] = 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < (CVW_SEG * 16) / LPS; ++i) cs_lds[i * LPS + sub] = 0.0f;  // this slot's cosine sums
    // slot-local LDS hand-off: the 8 lanes of a slot belong to one wave => wave-level ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- pass 2: walk the segment once per (pair, scale) with the two texel quads in registers
    int p = 0;
    for (int a = 0; a < V - 1; ++a) {
      for (int b = a + 1; b < V; ++b, ++p) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s >= sc.n_scales) break;
          const int fh = sc.fh[s], fw = sc.fw[s];
          const size_t map_elems = (size_t)fh * fw * FEAT_C;
          const float* m0 = sc.feat[s] + (size_t)(2 * p) * map_elems;
          const float* m1 = m0 + map_elems;
          const int G = sc.n_group[s];
          const int lpg = LPS / G;  // lanes per channel group
          const int goff = s ? G0 : 0;
          QuadCache<CPL> qa, qb;
          qa.o00 = qa.o01 = qa.o10 = qa.o11 = -1;
          qb.o00 = qb.o01 = qb.o10 = qb.o11 = -1;
          for (int js = 0; js < CVW_SEG; ++js) {
            const float ua = uv_lds[(js * V + a) * 2], va = uv_lds[(js * V + a) * 2 + 1];
            const float ub = uv_lds[(js * V + b) * 2], vb = uv_lds[(js * V + b) * 2 + 1];
            const Bilin ba = bilin_setup(ua, va, fh, fw);
            const Bilin bb = bilin_setup(ub, vb, fh, fw);
            if (quad_stale(qa, ba)) quad_load(qa, m0, ba, sub);
            if (quad_stale(qb, bb)) quad_load(qb, m1, bb, sub);
            float fa[CPL], fb[CPL];
            quad_interp(qa, ba, fa);
            quad_interp(qb, bb, fb);
            float dot = 0.f, na = 0.f, nb = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
              dot += fa[c] * fb[c];
              na += fa[c] * fa[c];
              nb += fb[c] * fb[c];
            }
            dot = group_reduce(dot, lpg);
            na = group_reduce(na, lpg);
            nb = group_reduce(nb, lpg);
            const float da = fmaxf(sqrtf(na), 1e-8f), db = fmaxf(sqrtf(nb), 1e-8f);
            if ((sub % lpg) == 0) cs_lds[js * 16 + goff + sub / lpg] += dot / (da * db);  // owner lane, pair order
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- write the averaged cosines: lane `sub` writes samples sub, sub+LPS, ..
#pragma unroll
    for (int half = 0; half < SPL; ++half) {
      const int js = sub + LPS * half;
      if (js < CVW_SEG && ray_live && (j0 + js < S)) {
        float* out = cond + ((size_t)ray * S + j0 + js) * cond_stride;
        for (int c = 0; c < sumG; ++c) out[c] = cs_lds[js * 16 + c] * inv_pairs;
      }
    }
    __builtin_amdgcn_wave_barrier();  // cs_lds / uv_lds are rewritten by the next unit
  }
}

static int check_scene(const mnerf_scene* sc, const mnerf_rays* rays, const char* who) {
  MNERF_REQUIRE(sc && rays, MNERF_E_NULL, "%s: NULL argument struct", who);
  MNERF_REQUIRE(sc->n_views >= 2 && sc->n_views <= MNERF_MAX_VIEWS, MNERF_E_RANGE,
                "%s: n_views=%d outside [2,%d]", who, sc->n_views, MNERF_MAX_VIEWS);
  MNERF_REQUIRE(sc->n_scales == 1 || sc->n_scales == 2, MNERF_E_RANGE, "%s: n_scales=%d", who,
                sc->n_scales);
  for (int s = 0; s < sc->n_scales; ++s) {
    const int G = sc->n_group[s];
    MNERF_REQUIRE(G == 1 || G == 2 || G == 4 || G == 8, MNERF_E_UNSUPPORTED,
                  "%s: cos_n_group[%d]=%d not in {1,2,4,8}", who, s, G);
    MNERF_REQUIRE(sc->feat[s], MNERF_E_NULL, "%s: feat[%d] is NULL", who, s);
    MNERF_REQUIRE(mnerf_aligned16(sc->feat[s]), MNERF_E_ALIGN, "%s: feat[%d] not 16B aligned", who, s);
    MNERF_REQUIRE(sc->fh[s] >= 1 && sc->fw[s] >= 1, MNERF_E_RANGE, "%s: feature map %d is %dx%d",
                  who, s, sc->fh[s], sc->fw[s]);
  }
  MNERF_REQUIRE(sc->images, MNERF_E_NULL, "%s: images is NULL", who);
  MNERF_REQUIRE(mnerf_aligned16(sc->images), MNERF_E_ALIGN, "%s: images not 16B aligned", who);
  MNERF_REQUIRE(rays->n_rays >= 0 && rays->n_samples >= 1, MNERF_E_RANGE, "%s: n_rays=%d S=%d",
                who, rays->n_rays, rays->n_samples);
  MNERF_REQUIRE(rays->height >= 2 && rays->width >= 2, MNERF_E_RANGE, "%s: image %dx%d", who,
                rays->height, rays->width);
  MNERF_REQUIRE(rays->legacy_coord == 0 || rays->n_samples >= 2, MNERF_E_RANGE,
                "%s: legacy depth sampling needs S >= 2", who);
  return MNERF_OK;
}

extern "C" int mnerf_cost_volume(const mnerf_scene* scene, const mnerf_rays* rays,
                                 int32_t cond_stride, float* cond, void* stream) {
  int rc = check_scene(scene, rays, "mnerf_cost_volume");
  if (rc) return rc;
  MNERF_REQUIRE(cond, MNERF_E_NULL, "mnerf_cost_volume: cond is NULL");
  int sumG = scene->n_group[0] + (scene->n_scales > 1 ? scene->n_group[1] : 0);
  MNERF_REQUIRE(cond_stride >= sumG + 4 * scene->n_views + 1, MNERF_E_RANGE,
                "mnerf_cost_volume: cond_stride=%d < cond_dim+1=%d", cond_stride,
                sumG + 4 * scene->n_views + 1);
  if (rays->n_rays == 0) return MNERF_OK;
  const long long total = (long long)rays->n_rays * rays->n_samples;
  long long blocks = (total + 31) / 32;  // 32 sample slots per 256-thread workgroup
  if (blocks > 2048) blocks = 2048;      // 8 workgroups per CU, contiguous chunk each
  int variant = 2;  // 1 / 2 = segment walk (register quad cache), 8 / 16 lanes per sample; 0 = one sample per slot iteration
  if (const char* e = getenv("MNERF_CV_VARIANT")) variant = atoi(e);
  if (sumG > 16) variant = 0;
  if (variant == 1 || variant == 2) {
    const int nslot = variant == 1 ? 32 : 16;
    const size_t lds = (size_t)(nslot * CVW_SEG * scene->n_views * 2 + nslot * CVW_SEG * 16) * sizeof(float);
    static size_t lds_set[3] = {0, 0, 0};
    if (lds > lds_set[variant]) {
      if (variant == 1)
        (void)hipFuncSetAttribute((const void*)cost_volume_walk_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      else
        (void)hipFuncSetAttribute((const void*)cost_volume_walk_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      lds_set[variant] = lds;
    }
    long long wgs = ((long long)rays->n_rays + nslot - 1) / nslot;
    int cap = variant == 1 ? 2048 : 4096;
    if (const char* e = getenv("MNERF_CV_GRID")) cap = atoi(e);
    if (wgs > cap) wgs = cap;
    if (variant == 1)
      hipLaunchKernelGGL(cost_volume_walk_kernel<16>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream,
                         *scene, *rays, cond_stride, cond);
    else
      hipLaunchKernelGGL(cost_volume_walk_kernel<8>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream,
                         *scene, *rays, cond_stride, cond);
  } else {
    hipLaunchKernelGGL(cost_volume_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       *scene, *rays, cond_stride, cond);
  }
  return mnerf_check_launch("mnerf_cost_volume");
}
