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
  // chunk c = blocks [c n / nwg, (c + 1) n / nwg): every chunk gets floor or ceil of the mean, so that all eight XCDs carry the same
  // load at every launch size (ceil(n / nwg) blocks per chunk left the last XCDs idle whenever n was not a multiple of nwg)
  const long long b_begin = (long long)chunk * blocks_total / nwg;
  const long long b_end = (long long)(chunk + 1) * blocks_total / nwg;

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
        out[sumG + 3 * v + 0] = bilin4(t00.x, t01.x, t10.x, t11.x, b);
        out[sumG + 3 * v + 1] = bilin4(t00.y, t01.y, t10.y, t11.y, b);
        out[sumG + 3 * v + 2] = bilin4(t00.z, t01.z, t10.z, t11.z, b);
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

// padding of a slot's LDS areas (floats): the stride becomes = 8 (projections) / 4 (cosine sums) mod 32
// MEASURED (round 4, same box, two runs each): padded 9.78-9.91 ms per frame, unpadded 9.57-9.59 - the conflicts the counters
// show are not on the kernel's critical path, and the padding costs more than it buys.  Off by default (-DCVW_PAD=1 to try).
#if defined(CVW_PAD) && CVW_PAD
#define CVW_PAD_UV(n) ((40 - ((n) & 31)) & 31)
#define CVW_PAD_CS(n) ((36 - ((n) & 31)) & 31)
#define CVW_PAD_REC 1
#else
#define CVW_PAD_UV(n) 0
#define CVW_PAD_CS(n) 0
#define CVW_PAD_REC 0
#endif
__host__ __device__ inline size_t cvw_lean_lds_floats(int nslot, int seg, int views_kept, int cs_pad) {
  const int uv = seg * views_kept * 2, cs = seg * cs_pad;
  return (size_t)nslot * ((uv + CVW_PAD_UV(uv)) + (seg * 4 + CVW_PAD_REC) * 4 + (cs + CVW_PAD_CS(cs)));
}

// ============================================================================ segment walk (stand-alone kernel)
// The walk itself lives in cv_walk.hpp (shared with the fused ray-chunk kernel); this kernel maps slots to rays.
template <int CPL, bool UVPAIR = false, bool POSES = false>  // POSES: pose table (mnerf_rays.pose_table), one pose per block of 16 rays
__global__ __launch_bounds__(256, (CPL == 16 ? 2 : CVW_WAVES)) void cost_volume_lean_kernel(mnerf_scene sc, mnerf_rays R,
                                                                                   int cond_stride,
                                                                                   float* __restrict__ cond,
                                                                                   int pair_begin, int pair_end) {
  constexpr int LPS = FEAT_C / CPL;     // lanes per sample slot (8 or 16)
  constexpr int NSLOT = 256 / LPS;      // ray slots per workgroup (32 or 16)
  extern __shared__ __attribute__((aligned(16))) float cvw_smem[];
  const int V = UVPAIR ? 2 : sc.n_views;  // views per sample kept in the slot's projection scratch
  const int sub = threadIdx.x % LPS;
  const int slot = threadIdx.x / LPS;                               // NSLOT adjacent rays
  const int sumG = sc.n_group[0] + (sc.n_scales > 1 ? sc.n_group[1] : 0);
  const int cs_stride = (sumG + 3) & ~3;                            // cosine sums per segment sample in LDS
  // LDS per slot: projections [js][view](u,v) | walk records [js][view a|b][idx|weights] (float4) | cosine sums [js][cs]
  // (experiment, CVW_PAD=1: every slot's three areas padded so that the SAME offset in different slots falls into different
  // LDS banks - the slots of a wave read their walk records / projections / cosine sums at the same offsets in one instruction,
  // and slot strides of 96, 256 and 192 floats are all = 0 mod 32: the 4.6 conflict cycles per LDS instruction of round 3's
  // PMC.  Measured slower, see CVW_PAD_* above.)
  const int uv_str = CVW_SEG * V * 2 + CVW_PAD_UV(CVW_SEG * V * 2), cs_str = CVW_SEG * cs_stride + CVW_PAD_CS(CVW_SEG * cs_stride);
  float* uv_lds = cvw_smem + (size_t)slot * uv_str;
  float4* wrec_lds = reinterpret_cast<float4*>(cvw_smem + (size_t)NSLOT * uv_str) + (size_t)slot * (CVW_SEG * 4 + CVW_PAD_REC);
  float* cs_lds = cvw_smem + (size_t)NSLOT * (uv_str + (CVW_SEG * 4 + CVW_PAD_REC) * 4) + slot * cs_str;
  const int S = R.n_samples;
  const int n_seg = (S + CVW_SEG - 1) / CVW_SEG;

  // XCD-major contiguous runs of ray blocks (see cost_volume_kernel)
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int chunk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin;
  const long long blocks_total = ((long long)R.n_rays + NSLOT - 1) / NSLOT;
  const long long b_begin = (long long)chunk * blocks_total / nwg;  // balanced contiguous runs (see cost_volume_kernel)
  const long long b_end = (long long)(chunk + 1) * blocks_total / nwg;

  for (long long it = b_begin * n_seg; it < b_end * n_seg; ++it) {
    const long long rb = it / n_seg;
    const int j0 = (int)(it - rb * n_seg) * CVW_SEG;
    long long ray_ll = rb * NSLOT + slot;
    const bool ray_live = ray_ll < R.n_rays;
    if (!ray_live) ray_ll = R.n_rays - 1;
    const int ray = (int)ray_ll;
    const int jrow = j0 < S ? j0 : S - 1;
    if constexpr (POSES) {
      mnerf_rays Rt = R;  // the block's NSLOT rays share a pose (rays_per_pose is a multiple of 64)
      const long long first = rb * NSLOT < R.n_rays ? rb * NSLOT : R.n_rays - 1;
      rays_for_pose(Rt, R, pose_of_ray(R, (int)first));
      cv_walk_unit<CPL, CVW_SEG, true, UVPAIR>(sc, Rt, ray, ray_live, j0, cond + ((size_t)ray * S + jrow) * cond_stride, cond_stride,
                                         uv_lds, wrec_lds, cs_lds, sub, pair_begin, pair_end);
    } else {
      cv_walk_unit<CPL, CVW_SEG, true, UVPAIR>(sc, R, ray, ray_live, j0, cond + ((size_t)ray * S + jrow) * cond_stride, cond_stride,
                                         uv_lds, wrec_lds, cs_lds, sub, pair_begin, pair_end);
    }
  }
}

// ============================================================================ texel tiles (MNERF_CV_VARIANT=5; NOT the default)
// The walk above fetches every tap through the texture path: ~9 load instructions per wave-step, most of them a quarter full
// (the four slots of a wave cross texel boundaries at different steps), the texture-address unit ~85 % busy.  The 16 adjacent
// rays x 8 samples of a workgroup step touch only 12-30 distinct texels of a map (tools/exp/cv_torus.py) and read each of them
// ~20 times, so this kernel has the WORKGROUP stage those texels in LDS once per (pair, scale) and the walk read its taps with
// ds_read_b128:
//   B  lane (js, side) of a slot evaluates the walk record of its sample and view (as before) and CLAIMS an LDS place for each of
//      the record's four texels: place = (x mod 16, y mod 3) of a small torus per map, claimed with one LDS compare-and-swap on a
//      tag word (EMPTY -> texel index).  A tile whose footprint is at most 16 x 3 texels claims without collisions; a texel that
//      finds its place taken by a different texel keeps its global index (encoded negative) and is loaded as before.
//   C  all 256 threads copy the claimed texels into their places (32 lanes x 16 B per texel, 8 texels per pass) and clear the
//      other tag set for the next (pair, scale);
//   D  the slots walk: tile_walk (straight-line, taps of step js+1 requested before the arithmetic of step js) when every texel
//      of the wave's walk is staged, lean_walk<.., TILE> otherwise.  Same arithmetic, same order, same bits as the walk kernel.
// Two workgroup barriers per (pair, scale): after B (claims complete; every wave has also left the previous walk, so the
// texel area may be overwritten) and after C.  Wave-local scratch (projections, walk records, cosine sums) as in the walk kernel.
//
// MEASURED (MI355X, tools/exp/cvt_check.sh, cvt_stats.py): bit-identical rows, and SLOWER than the walk kernel - 15.8 vs 9.8 ms
// per frame at 3 views (0.29 % of the references lose their place, 7 % of the wave-walks take the cached walk), 286 vs 182 ms at
// 10 views (11 % lose their place: wider baselines, 36-72 distinct texels per tile and map).  The staged walk itself runs at
// 1 380 cycles per wave-step with two waves per SIMD = 690 per SIMD and step, against 750 for the walk kernel (3 000 at four
// waves): 126 VALU instructions per step (88 of them the interpolation and the three dot products) are ~504 issue cycles, so
// BOTH kernels sit at 60-73 % of the vector-ALU bound of this formulation and the texture path was never the only wall; what the
// tiles add - claims 11 %, copy 15 %, barriers 6 %, 247 VGPRs = half the waves - is not paid back.  The packed-fp32 form of the
// arithmetic would halve the 88 (see cv_walk.hpp for why it is not used).  Kept as an opt-in because it is the measured answer to
// "stage the texels cooperatively", not because it is useful.
#define CVT_SEG 8
#define CVT_TW 16           // torus: 16 texels wide (the 16 rays of a tile are neighbours along x), 3 rows
#define CVT_TH 3
#define CVT_PLACES_MAX (CVT_TW * CVT_TH)
__device__ __forceinline__ int cvt_ymod(int y) { return y - (int)(__umulhi((unsigned)y, 0xAAAAAAABu) >> 1) * 3; }  // y mod 3
__host__ __device__ inline size_t cvt_lds_bytes(int n_views, int sum_groups) {
  return (size_t)2 * CVT_PLACES_MAX * FEAT_C * 4 + (size_t)2 * 2 * CVT_PLACES_MAX * 4 +
         (size_t)16 * cv_slot_lds_floats(CVT_SEG, n_views, sum_groups) * 4;
}

#ifdef CVT_STATS
#define CVT_DBG_PARAM , unsigned long long* __restrict__ dbg
#define CVT_T(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[i] += t_ - t_last; t_last = t_; }
#else
#define CVT_DBG_PARAM
#define CVT_T(i)
#endif
__global__ __launch_bounds__(256, 2) void cost_volume_tile_kernel(mnerf_scene sc, mnerf_rays R, int cond_stride,
                                                                  float* __restrict__ cond CVT_DBG_PARAM) {
#ifdef CVT_STATS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
  unsigned n_fallback = 0, n_claims = 0, n_slow = 0;
#endif
  constexpr int CPL = 8, LPS = 16, NSLOT = 16, SEG = CVT_SEG;
  static_assert(LPS == 2 * SEG, "a slot's lanes are the (sample, side) pairs of one walk");
  extern __shared__ __attribute__((aligned(16))) float cvt_smem[];
  const int V = sc.n_views;
  const int tid = threadIdx.x;
  const int sub = tid % LPS, slot = tid / LPS;
  const int G0 = sc.n_group[0], G1 = sc.n_scales > 1 ? sc.n_group[1] : 0;
  const int sumG = G0 + G1;
  const int cs_stride = (sumG + 3) & ~3;
  const float inv_pairs = 1.0f / (float)(V * (V - 1) / 2);
  const unsigned lane_bytes = (unsigned)sub * CPL * 4;
  // LDS: staged texels [2 maps][places][512 B] | tags [2 sets][2 maps x places] | per slot: uv | walk records | cosine sums
  char* tex = reinterpret_cast<char*>(cvt_smem);
  int* tags = reinterpret_cast<int*>(cvt_smem + 2 * CVT_PLACES_MAX * FEAT_C);
  float* scratch = cvt_smem + 2 * CVT_PLACES_MAX * FEAT_C + 2 * 2 * CVT_PLACES_MAX;
  float* uv_lds = scratch + (size_t)slot * SEG * V * 2;
  float4* wrec_lds = reinterpret_cast<float4*>(scratch + (size_t)NSLOT * SEG * V * 2) + (size_t)slot * SEG * 4;
  float* cs_lds = scratch + (size_t)NSLOT * SEG * (V * 2 + 16) + slot * SEG * cs_stride;
  const int S = R.n_samples;
  const int n_seg = (S + SEG - 1) / SEG;

  // XCD-major contiguous runs of ray blocks (see cost_volume_kernel)
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int chunk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin;
  const long long blocks_total = ((long long)R.n_rays + NSLOT - 1) / NSLOT;
  const long long b_begin = (long long)chunk * blocks_total / nwg;  // balanced contiguous runs (see cost_volume_kernel)
  const long long b_end = (long long)(chunk + 1) * blocks_total / nwg;

  for (int i = tid; i < 2 * 2 * CVT_PLACES_MAX; i += 256) tags[i] = -1;
  __syncthreads();
  int tag_set = 0;  // the set the next (pair, scale) claims in; the other one is cleared meanwhile

  const int js_mine = sub & (SEG - 1), side_mine = sub >> 3;  // phase B: this lane's (sample, side)
  for (long long it = b_begin * n_seg; it < b_end * n_seg; ++it) {
    const long long rb = it / n_seg;
    const int j0 = (int)(it - rb * n_seg) * SEG;
    long long ray_ll = rb * NSLOT + slot;
    const bool ray_live = ray_ll < R.n_rays;
    if (!ray_live) ray_ll = R.n_rays - 1;
    const int ray = (int)ray_ll;
    const int jrow = j0 < S ? j0 : S - 1;
    float* row0 = cond + ((size_t)ray * S + jrow) * cond_stride;

    CVT_T(7)
    cv_pass1<CPL, SEG, true>(sc, R, ray, ray_live, j0, row0, cond_stride, uv_lds, sub);
    for (int i = sub; i < SEG * cs_stride; i += LPS) cs_lds[i] = 0.0f;
    cvw_handoff();
    CVT_T(0)

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
          const int lpg = LPS / G;
          const int goff = s ? G0 : 0;
          float* cs_group = cs_lds + goff + sub / lpg;
          constexpr int places = CVT_PLACES_MAX;
          int* tg = tags + tag_set * (2 * CVT_PLACES_MAX);

          // ---- B: walk record of (js_mine, side_mine), its four texels claimed in the torus of that map
          bool wave_all_staged;
          cvw_handoff();  // this wave's previous walk has read its records
          {
            const int vw = side_mine ? b : a;
            int x0, y0;
            const TapRec t = tap_setup_xy(uv_lds[(js_mine * V + vw) * 2], uv_lds[(js_mine * V + vw) * 2 + 1], fh, fw, x0, y0);
            float4 ri, rw;
            tap_expand(t, fw, ri, rw);
            int ix[4] = {__float_as_int(ri.x), __float_as_int(ri.y), __float_as_int(ri.z), __float_as_int(ri.w)};
            const int fl = t.flags;  // bit0/1: x offset of the even/odd column set, bit2/3: row offset of the even/odd row set
            const int xm[2] = {(x0 + (fl & 1)) & (CVT_TW - 1), (x0 + ((fl >> 1) & 1)) & (CVT_TW - 1)};
            const int ym[2] = {cvt_ymod(y0 + ((fl >> 2) & 1)), cvt_ymod(y0 + ((fl >> 3) & 1))};
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // {EE, EO, OE, OO} = [row set][column set], tap_expand()'s order
              const int place = side_mine * places + ym[k >> 1] * CVT_TW + xm[k & 1];
              const int old = atomicCAS(tg + place, -1, ix[k]);
              ix[k] = (old == -1 || old == ix[k]) ? place * (FEAT_C * 4) : ~ix[k];
#ifdef CVT_STATS
              n_claims++;
              n_fallback += ix[k] < 0;
#endif
            }
            wave_all_staged = __builtin_amdgcn_ballot_w64((ix[0] | ix[1] | ix[2] | ix[3]) < 0) == 0;
            wrec_lds[(js_mine * 2 + side_mine) * 2] =
                make_float4(__int_as_float(ix[0]), __int_as_float(ix[1]), __int_as_float(ix[2]), __int_as_float(ix[3]));
            wrec_lds[(js_mine * 2 + side_mine) * 2 + 1] = rw;
          }
          CVT_T(1)
          __syncthreads();
          CVT_T(2)

          // ---- C: copy the claimed texels into their places; clear the other tag set
          {
            const int l32 = tid & 31;
            const unsigned dst_lane = (unsigned)((l32 & 1) * 256 + (l32 >> 1) * 16);  // [half][owner lane] (tap_load_tile)
            constexpr int n_ent = 2 * places;  // 96: whole groups of 32
            for (int e0 = tid >> 5; e0 < n_ent; e0 += 32) {
              int tx[4];
              v4f val[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) tx[q] = tg[e0 + 8 * q];
#pragma unroll
              for (int q = 0; q < 4; ++q) {  // unclaimed places read texel 0 (one cached line) and store nothing
                const int e = e0 + 8 * q;
                val[q] = *reinterpret_cast<const v4f*>((e >= places ? m1 : m0) + (size_t)max(tx[q], 0) * FEAT_C + l32 * 4);
              }
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (tx[q] >= 0) *reinterpret_cast<v4f*>(tex + (unsigned)(e0 + 8 * q) * (FEAT_C * 4) + dst_lane) = val[q];
            }
            int* other = tags + (tag_set ^ 1) * (2 * CVT_PLACES_MAX);
            if (tid < 2 * CVT_PLACES_MAX) other[tid] = -1;
            tag_set ^= 1;
          }
          CVT_T(3)
          __syncthreads();
          CVT_T(4)

          // ---- D: walk
          if (wave_all_staged) {
            switch (lpg) {
              case 2: tile_walk<CPL, 2, SEG>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes, tex); break;
              case 4: tile_walk<CPL, 4, SEG>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes, tex); break;
              default: tile_walk<CPL, 8, SEG>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes, tex); break;
            }
          } else {  // some texel of this wave's walk lost its LDS place: the walk with the register tap cache
#ifdef CVT_STATS
            n_slow++;
#endif
            switch (lpg) {
              case 2: lean_walk<CPL, 2, SEG, true>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes, tex); break;
              case 4: lean_walk<CPL, 4, SEG, true>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes, tex); break;
              default: lean_walk<CPL, 8, SEG, true>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes, tex); break;
            }
          }
          CVT_T(5)
        }
      }
    }
    cvw_handoff();
    cv_write_cosines<CPL, SEG, true>(R, ray_live, j0, row0, cond_stride, cs_lds, cs_stride, sumG, inv_pairs, sub);
    __builtin_amdgcn_wave_barrier();
    CVT_T(6)
  }
#ifdef CVT_STATS
  if (dbg) {
    if ((tid & 63) == 0)
      for (int i = 0; i < 8; ++i) atomicAdd(dbg + i, st[i]);
    atomicAdd(dbg + 8, (unsigned long long)n_claims);
    atomicAdd(dbg + 9, (unsigned long long)n_fallback);
    if (tid == 0) atomicAdd(dbg + 10, 1ull);
    if ((tid & 63) == 0) atomicAdd(dbg + 11, (unsigned long long)n_slow);
  }
#endif
}

int mnerf_scene_check(const mnerf_scene* sc, const mnerf_rays* rays, const char* who) {
  MNERF_REQUIRE(sc && rays, MNERF_E_NULL, "%s: NULL argument struct", who);
  MNERF_REQUIRE(rays->pose_table || rays->rays_per_pose == 0, MNERF_E_RANGE, "%s: rays_per_pose=%d without a pose table", who,
                rays->rays_per_pose);
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

// Which kernel a scene gets: variant 3 / 4 = segment walk with 16 / 8 lanes per sample, 5 = texel tiles, 0 = one sample per slot
// iteration; uvpair = the 16-lane walk that keeps only the current pair's projections (many views).
static void cv_pick_kernel(const mnerf_scene* scene, int sumG, int* variant_out, bool* uvpair_out) {
  int variant = mnerf_tune().cv_variant;
  if (variant != 3 && variant != 4 && variant != 5) variant = 0;
  if (sumG > CVW_CS_MAX) variant = 0;
  if (variant == 5) {  // texel tiles: 8-sample walks need at most 8 lanes per channel group; LDS for two workgroups per CU
    bool ok = cvt_lds_bytes(scene->n_views, sumG) <= 80 * 1024;
    for (int s = 0; s < scene->n_scales; ++s) ok = ok && scene->n_group[s] >= 2;
    if (!ok) variant = 3;
  }
  // A workgroup's LDS is NSLOT x SEG x (2 V + 16 + cs) floats: 38 KiB at 3 views (four workgroups per CU, what 128 VGPRs
  // allow), 52 KiB at 10 views (three).  From the view count at which the fourth workgroup no longer fits, the 16-lane form
  // keeps only the current pair's projections (UVPAIR, cv_walk.hpp).
  bool uvpair = variant == 3 && cvw_lean_lds_floats(16, CVW_SEG, scene->n_views, (sumG + 3) & ~3) * sizeof(float) > 40 * 1024;
  if (variant == 3 && mnerf_tune().cv_uvpair >= 0) uvpair = mnerf_tune().cv_uvpair != 0;
  *variant_out = variant, *uvpair_out = uvpair;
}

// the pose-table instance of the cost volume is the 16-lane walk with all views' projections resident (<= 5 views)
bool mnerf_cost_volume_takes_pose_table(const mnerf_scene* scene) {
  int sumG = 0, variant;
  bool uvpair;
  for (int s = 0; s < scene->n_scales; ++s) sumG += scene->n_group[s];
  cv_pick_kernel(scene, sumG, &variant, &uvpair);
  return variant == 3 && !uvpair;
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
  if (mnerf_tune().cv_mm && mnerf_cost_volume_mm_applies(scene, rays))  // matrix form (cost_volume_mm.hip)
    return mnerf_cost_volume_mm_launch(scene, rays, cond_stride, cond, stream);
  const bool poses = rays->pose_table != nullptr;
  if (poses) {
    MNERF_REQUIRE(rays->rays_per_pose > 0 && rays->rays_per_pose % 64 == 0, MNERF_E_RANGE,
                  "mnerf_cost_volume: pose table needs rays_per_pose = a positive multiple of 64, got %d", rays->rays_per_pose);
    MNERF_REQUIRE(!rays->ray_idx && !rays->strat_u, MNERF_E_UNSUPPORTED, "mnerf_cost_volume: a pose table excludes ray_idx / strat_u");
  }
  const long long total = (long long)rays->n_rays * rays->n_samples;
  long long blocks = (total + 31) / 32;  // 32 sample slots per 256-thread workgroup
  if (blocks > 2048) blocks = 2048;      // 8 workgroups per CU, contiguous chunk each
  int variant;
  bool uvpair;
  cv_pick_kernel(scene, sumG, &variant, &uvpair);
  MNERF_REQUIRE(!poses || (variant == 3 && !uvpair), MNERF_E_UNSUPPORTED,
                "mnerf_cost_volume: a pose table needs the 16-lane segment walk with all views' projections resident "
                "(cv_variant 3, <= 5 source views; mnerf_render_takes_pose_table tells)");
  if (variant == 5) {
    const size_t lds = cvt_lds_bytes(scene->n_views, sumG);
    static std::atomic<int> tile_lds_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<int>& seen = tile_lds_set[dev & 63];
    if ((int)lds > seen.load(std::memory_order_relaxed)) {
      (void)hipFuncSetAttribute((const void*)cost_volume_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      seen.store((int)lds, std::memory_order_relaxed);
    }
    long long wgs = ((long long)rays->n_rays + 15) / 16;
    int cap = 512;  // two resident workgroups per CU, one contiguous run of ray blocks each
    if (mnerf_tune().cv_grid > 0) cap = mnerf_tune().cv_grid;
    if (wgs > cap) wgs = cap;
#ifdef CVT_STATS
    unsigned long long* dbg = nullptr;
    if (const char* e = getenv("MNERF_CVDBG_PTR")) dbg = (unsigned long long*)strtoull(e, nullptr, 0);
    hipLaunchKernelGGL(cost_volume_tile_kernel, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream, *scene, *rays,
                       cond_stride, cond, dbg);
#else
    hipLaunchKernelGGL(cost_volume_tile_kernel, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream, *scene, *rays,
                       cond_stride, cond);
#endif
    return mnerf_check_launch("mnerf_cost_volume");
  }
  if (variant == 3 || variant == 4) {  // lean walk: 16 / 8 lanes per sample
    const int nslot = variant == 4 ? 32 : 16;
    const int cs_pad = (sumG + 3) & ~3;
    const size_t lds = cvw_lean_lds_floats(nslot, CVW_SEG, uvpair ? 2 : scene->n_views, cs_pad) * sizeof(float);
    MNERF_REQUIRE(lds <= 160 * 1024, MNERF_E_UNSUPPORTED, "mnerf_cost_volume: %d views need %zu B of LDS", scene->n_views, lds);
    // the LDS attribute is per device and only ever raised: largest request seen per (variant, device)
    static std::atomic<int> lean_lds_set[3][64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int which = variant == 4 ? 1 : (uvpair ? 2 : 0);
    std::atomic<int>& seen = lean_lds_set[which][dev & 63];
    if ((int)lds > seen.load(std::memory_order_relaxed)) {
      if (which == 1)
        (void)hipFuncSetAttribute((const void*)cost_volume_lean_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      else if (which == 2)
        (void)hipFuncSetAttribute((const void*)cost_volume_lean_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      else
        (void)hipFuncSetAttribute((const void*)cost_volume_lean_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      seen.store((int)lds, std::memory_order_relaxed);
    }
    long long wgs = ((long long)rays->n_rays + nslot - 1) / nslot;
    // 16-lane walk: ONE ray block per workgroup at every launch size.  A capped grid (4 096 until round 4) gave the workgroups
    // 1 or 2 blocks each whenever the launch was not a multiple of the cap, and the hardware's round-robin placement put the
    // 2-block workgroups on the same quarter of the CUs: 81 920 rays took 15.7 ms where 65 536 take 9.5 (now 9.4; one launch
    // of the whole 327 680-ray frame 9.2).  MNERF_CV_GRID caps it again.
    long long cap = variant == 4 ? 2048 : (1ll << 30);
    if (mnerf_tune().cv_grid > 0) cap = mnerf_tune().cv_grid;
    if (wgs > cap) wgs = cap;
    const int n_pairs = scene->n_views * (scene->n_views - 1) / 2;
    if (which == 1)
      hipLaunchKernelGGL(cost_volume_lean_kernel<16>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream,
                         *scene, *rays, cond_stride, cond, 0, n_pairs);
    else if (which == 2) {
      // Many views: one launch per BLOCK of view pairs over all rays, so that the maps a launch gathers from fit the 256 MiB
      // Infinity Cache (cv_walk.hpp "PAIR BLOCKS"; 8 pairs x 2 sides x 13.1 MB at 512x640 = 210 MB).  Same stream: the blocks
      // run in order, each continues the cosine sums the previous one left in the rows.  MNERF_CV_PAIR_BLOCK: pairs per
      // launch (default 8; 0 = all pairs in one launch, the round-3 form).
      const int pb = mnerf_tune().cv_pair_block;
      const int blk = pb > 0 ? pb : (pb < 0 ? 8 : n_pairs);  // (-1 = default: 8 pairs per launch of the walk)
      for (int p0 = 0; p0 < n_pairs; p0 += blk)
        hipLaunchKernelGGL((cost_volume_lean_kernel<8, true>), dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream,
                           *scene, *rays, cond_stride, cond, p0, p0 + blk < n_pairs ? p0 + blk : n_pairs);
    } else if (poses) {
      static std::atomic<int> pose_lds_set[64];
      std::atomic<int>& pseen = pose_lds_set[dev & 63];
      if ((int)lds > pseen.load(std::memory_order_relaxed)) {
        (void)hipFuncSetAttribute((const void*)cost_volume_lean_kernel<8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        pseen.store((int)lds, std::memory_order_relaxed);
      }
      hipLaunchKernelGGL((cost_volume_lean_kernel<8, false, true>), dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream,
                         *scene, *rays, cond_stride, cond, 0, n_pairs);
    } else
      hipLaunchKernelGGL(cost_volume_lean_kernel<8>, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream,
                         *scene, *rays, cond_stride, cond, 0, n_pairs);
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
