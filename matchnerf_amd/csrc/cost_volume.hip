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

#include "cv_walk.hpp"

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
#ifdef CV_PROBE_DUP
#define CV_DBG_PARAM , unsigned* __restrict__ dbg
#else
#define CV_DBG_PARAM
#endif
__global__ __launch_bounds__(256, CV_WAVES_PER_SIMD) void cost_volume_kernel(mnerf_scene sc, mnerf_rays R,
                                                          int cond_stride,
                                                          float* __restrict__ cond CV_DBG_PARAM) {
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
#ifdef CV_PROBE_DUP
      {  // race probe: the same four taps loaded a second time (opaque addresses); any difference is recorded
        int o00 = b.o00, o01 = b.o01, o10 = b.o10, o11 = b.o11;
        asm volatile("" : "+v"(o00), "+v"(o01), "+v"(o10), "+v"(o11));
        const float4 r00 = img[o00], r01 = img[o01], r10 = img[o10], r11 = img[o11];
        const bool d0 = t00.x != r00.x || t00.y != r00.y || t00.z != r00.z, d1 = t01.x != r01.x || t01.y != r01.y || t01.z != r01.z,
                   d2 = t10.x != r10.x || t10.y != r10.y || t10.z != r10.z, d3 = t11.x != r11.x || t11.y != r11.y || t11.z != r11.z;
        if (dbg && (d0 || d1 || d2 || d3)) {
          const unsigned k = atomicAdd(dbg, 1u);
          if (k < 64) {
            unsigned* o = dbg + 16 + k * 40;
            o[0] = threadIdx.x & 63, o[1] = (unsigned)ray, o[2] = (unsigned)j, o[3] = (unsigned)v;
            o[4] = (d0 ? 1u : 0u) | (d1 ? 2u : 0u) | (d2 ? 4u : 0u) | (d3 ? 8u : 0u);
            o[5] = (unsigned)o00, o[6] = (unsigned)o01, o[7] = (unsigned)o10, o[8] = (unsigned)o11;
            const float f[24] = {t00.x, t00.y, t00.z, t01.x, t01.y, t01.z, t10.x, t10.y, t10.z, t11.x, t11.y, t11.z,
                                 r00.x, r00.y, r00.z, r01.x, r01.y, r01.z, r10.x, r10.y, r10.z, r11.x, r11.y, r11.z};
            for (int q = 0; q < 24; ++q) o[9 + q] = __float_as_uint(f[q]);
            o[33] = __float_as_uint(b.w00), o[34] = __float_as_uint(b.w01), o[35] = __float_as_uint(b.w10), o[36] = __float_as_uint(b.w11);
            o[37] = blockIdx.x, o[38] = (unsigned)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
          }
        }
      }
#endif
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

// ============================================================================ segment walk (stand-alone kernel)
// The walk itself lives in cv_walk.hpp (shared with the fused ray-chunk kernel); this kernel maps slots to rays.
template <int CPL>
__global__ __launch_bounds__(256, (CPL == 16 ? 2 : CVW_WAVES)) void cost_volume_lean_kernel(mnerf_scene sc, mnerf_rays R,
                                                                                   int cond_stride,
                                                                                   float* __restrict__ cond) {
  constexpr int LPS = FEAT_C / CPL;     // lanes per sample slot (8 or 16)
  constexpr int NSLOT = 256 / LPS;      // ray slots per workgroup (32 or 16)
  extern __shared__ __attribute__((aligned(16))) float cvw_smem[];
  const int V = sc.n_views;
  const int sub = threadIdx.x % LPS;
  const int slot = threadIdx.x / LPS;                               // NSLOT adjacent rays
  const int sumG = sc.n_group[0] + (sc.n_scales > 1 ? sc.n_group[1] : 0);
  const int cs_stride = (sumG + 3) & ~3;                            // cosine sums per segment sample in LDS
  // LDS per slot: projections [js][view](u,v) | walk records [js][view a|b][idx|weights] (float4) | cosine sums [js][cs]
  float* uv_lds = cvw_smem + (size_t)slot * CVW_SEG * V * 2;
  float4* wrec_lds = reinterpret_cast<float4*>(cvw_smem + (size_t)NSLOT * CVW_SEG * V * 2) + (size_t)slot * CVW_SEG * 4;
  float* cs_lds = cvw_smem + (size_t)NSLOT * CVW_SEG * (V * 2 + 16) + slot * CVW_SEG * cs_stride;
  const int S = R.n_samples;
  const int n_seg = (S + CVW_SEG - 1) / CVW_SEG;

  // XCD-major contiguous runs of ray blocks (see cost_volume_kernel)
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
    const int jrow = j0 < S ? j0 : S - 1;
    cv_walk_unit<CPL, CVW_SEG, true>(sc, R, ray, ray_live, j0, cond + ((size_t)ray * S + jrow) * cond_stride, cond_stride,
                               uv_lds, wrec_lds, cs_lds, sub);
  }
}

int mnerf_scene_check(const mnerf_scene* sc, const mnerf_rays* rays, const char* who) {
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
  int rc = mnerf_scene_check(scene, rays, "mnerf_cost_volume");
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
  int variant = mnerf_tune().cv_variant;  // 3 / 4 = segment walk with 16 / 8 lanes per sample; 0 = one sample per slot iteration
  if (variant != 3 && variant != 4) variant = 0;
  if (sumG > CVW_CS_MAX) variant = 0;
  if (variant == 3 || variant == 4) {  // lean walk: 16 / 8 lanes per sample
    const int nslot = variant == 4 ? 32 : 16;
    const size_t lds = (size_t)nslot * CVW_SEG * (scene->n_views * 2 + 16 + ((sumG + 3) & ~3)) * sizeof(float);
    MNERF_REQUIRE(lds <= 160 * 1024, MNERF_E_UNSUPPORTED, "mnerf_cost_volume: %d views need %zu B of LDS", scene->n_views, lds);
    // the LDS attribute is per device and only ever raised: largest request seen per (variant, device)
    static std::atomic<int> lean_lds_set[2][64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<int>& seen = lean_lds_set[variant - 3][dev & 63];
    if ((int)lds > seen.load(std::memory_order_relaxed)) {
      if (variant == 4)
        (void)hipFuncSetAttribute((const void*)cost_volume_lean_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      else
        (void)hipFuncSetAttribute((const void*)cost_volume_lean_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      seen.store((int)lds, std::memory_order_relaxed);
    }
    long long wgs = ((long long)rays->n_rays + nslot - 1) / nslot;
    int cap = variant == 4 ? 2048 : 4096;
    if (mnerf_tune().cv_grid > 0) cap = mnerf_tune().cv_grid;
    if (wgs > cap) wgs = cap;
    if (variant == 4)
      hipLaunchKernelGGL(cost_volume_lean_kernel<16>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream,
                         *scene, *rays, cond_stride, cond);
    else
      hipLaunchKernelGGL(cost_volume_lean_kernel<8>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream,
                         *scene, *rays, cond_stride, cond);
  } else {
#ifdef CV_PROBE_DUP
    unsigned* dbg = nullptr;
    if (const char* e = getenv("MNERF_CVDBG_PTR")) dbg = (unsigned*)strtoull(e, nullptr, 0);
    hipLaunchKernelGGL(cost_volume_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       *scene, *rays, cond_stride, cond, dbg);
#else
    hipLaunchKernelGGL(cost_volume_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       *scene, *rays, cond_stride, cond);
#endif
  }
  return mnerf_check_launch("mnerf_cost_volume");
}
