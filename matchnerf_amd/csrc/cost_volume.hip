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

// ============================================================================ segment walk
// Same arithmetic as cost_volume_kernel, different traversal.  PMC showed that kernel bound by
// the bytes the texture path delivers to registers (~18 TB/s of taps, 53 % of L1 peak), not by
// L2/HBM — and consecutive samples of a ray move only ~1/4 texel at 1/8 resolution (~1/2 at 1/4),
// so most of those bytes are the same texels again.  Here a slot (16 or 8 lanes) WALKS a segment
// of CVW_SEG consecutive samples of one ray; for each view pair and scale it keeps the current 2x2
// texel quad of both maps in registers and reloads a quad only when its top-left texel changes.
// Per-sample cosine sums over pairs accumulate in LDS.
// The first walk kernel (19.7 ms/frame) turned out VALU-bound on its own bookkeeping: 96 % VALU
// busy, ~360 VALU instructions per (sample, pair, scale) step of a wave against ~90 of
// interpolation + dot products.  This version (12.9 ms/frame) keeps the traversal with a lean step:
//  * nothing that is the same for the 16 lanes of a slot is evaluated per lane per step: the bilinear set-up
//    and the parity-set bookkeeping of a (sample, view) are evaluated once per (pair, scale) walk by the
//    lane whose index equals the sample's position in the segment and handed over as a 32-byte LDS record;
//    the cosine of a (sample, group) is evaluated by one lane, LPG samples at a time;
//  * interpolation and the three dot products run on channel PAIRS (v_pk_mul/v_pk_fma_f32: the
//    dwordx4 loads already put consecutive channels in consecutive registers);
//  * the group reductions are compile-time DPP butterflies (quad_perm / row_half_mirror /
//    row_mirror) instead of ds_bpermute shuffles with a run-time width;
//  * reload addresses are 32-bit byte offsets from the scalar map base;
//  * the per-pair cosines accumulate with LDS float atomics (one fixed lane per (sample, group), program
//    order per lane, so the sum over pairs keeps the reference's pair order).
// Where it stands (PMC): 113 VALU instructions per step (from ~360), VALU 60 % busy, texture-address unit
// 85 % busy: 8.8 tap-load instructions per wave-step, most of them with a quarter of the lanes active because
// a wave's four slots rarely cross a texel boundary in the same step.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// pass-1 record of one (sample, view, scale).  The four taps of a quad are kept in four register
// sets named by the PARITY of the texel's row and column (E/O), not by their position in the quad:
// when the walk crosses one texel boundary only the two sets of the leaving row/column change, the
// other two are reused as they are — no register moves, half the tap traffic of reloading a quad
// (the kernel was bound by the texture-address unit: TA busy 92 % with whole-quad reloads).
// The record carries what the inner loop needs to find each set's texel: offsets from the top-left
// texel (+1 / +w if that neighbour is inside the map) and the parities.
struct TapRec {
  int o00;     // top-left texel index y0*w + x0
  float fx, fy;
  int flags;   // bit0/1: x offset of the even/odd column set; bit2/3: row offset (x w) of the even/odd
               // row set; bit4: x0 odd; bit5: y0 odd
};

__device__ __forceinline__ TapRec tap_setup(float u, float v, int h, int w) {  // bilin_setup()'s arithmetic
  float gx = u * 2.0f - 1.0f, gy = v * 2.0f - 1.0f;
  float x = ((gx + 1.0f) * 0.5f) * (float)(w - 1);
  float y = ((gy + 1.0f) * 0.5f) * (float)(h - 1);
  x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
  y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int dx = x0 + 1 <= w - 1 ? 1 : 0, dy = y0 + 1 <= h - 1 ? 1 : 0;
  const int px = x0 & 1, py = y0 & 1;
  TapRec r;
  r.o00 = y0 * w + x0;
  r.fx = x - x0f;
  r.fy = y - y0f;
  r.flags = (px & dx) | ((~px & dx) << 1) | ((py & dy) << 2) | ((~py & dy) << 3) | (px << 4) | (py << 5);
  return r;
}

#ifndef CVW_SEG
#define CVW_SEG 16
#endif
#define CVW_CS_MAX 16  // cosine sums per sample the walk kernel supports (sum of groups)
#ifndef CVW_FAST_COS
#define CVW_FAST_COS 0
#endif
#ifndef CVW_WAVES
#define CVW_WAVES 4  // 128 VGPRs; LDS (40 KB/workgroup at 3 views) allows 4 workgroups per CU
#endif

template <int CPL>  // CPL = channels per lane (8 -> 16 lanes per sample, 16 -> 8 lanes per sample)
struct PairQuad {
  v2f t[2][2][CPL / 2];  // [row parity][column parity][channel pair]
  int idx[2][2];         // texel held by each set (-1: none)
};

template <int CPL>
__device__ __forceinline__ void tap_load(v2f (&t)[CPL / 2], const float* __restrict__ map, int texel,
                                         unsigned lane_bytes) {
  const v4f* p = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(map) +
                                              ((unsigned)texel * (unsigned)(FEAT_C * 4) + lane_bytes));
#pragma unroll
  for (int k = 0; k < CPL / 4; ++k) {
    const v4f a = p[k];
    t[2 * k] = a.lo;
    t[2 * k + 1] = a.hi;
  }
}

// Expanded walk record of one (sample, view) at the scale being walked: texel held by each parity set
// {EE, EO, OE, OO} and that set's bilinear weight.  Everything in it is the same for the 16 lanes of a slot, so
// it is evaluated ONCE per (pair, scale) walk by the lane whose index equals the sample's position in the
// segment (52 VALU ops per step in every lane became ~6) and handed over through 32 bytes of LDS.
__device__ __forceinline__ void tap_expand(const TapRec& t, int w, float4& idx, float4& wts) {
  const int fl = t.flags;
  const int ex0 = fl & 1, ex1 = (fl >> 1) & 1;
  const int ey0 = (fl & 4) ? w : 0, ey1 = (fl & 8) ? w : 0;
  const int b0 = t.o00 + ey0, b1 = t.o00 + ey1;
  idx = make_float4(__int_as_float(b0 + ex0), __int_as_float(b0 + ex1), __int_as_float(b1 + ex0), __int_as_float(b1 + ex1));
  const float fx = t.fx, fy = t.fy, gx = 1.0f - fx, gy = 1.0f - fy;
  const bool px = fl & 16, py = fl & 32;
  const float wxE = px ? fx : gx, wxO = px ? gx : fx;  // the even column is x0 (weight 1-fx) iff x0 is even
  const float wyE = py ? fy : gy, wyO = py ? gy : fy;
  // same products as bilin_setup(): (1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx fy
  wts = make_float4(wxE * wyE, wxO * wyE, wxE * wyO, wxO * wyO);
}

// bring the four parity sets up to date for this walk record
template <int CPL>
__device__ __forceinline__ void quad_update(PairQuad<CPL>& q, const float* __restrict__ map, const float4 ix,
                                            unsigned lane_bytes) {
  const int i00 = __float_as_int(ix.x), i01 = __float_as_int(ix.y), i10 = __float_as_int(ix.z), i11 = __float_as_int(ix.w);
  if (i00 != q.idx[0][0]) { tap_load<CPL>(q.t[0][0], map, i00, lane_bytes); q.idx[0][0] = i00; }
  if (i01 != q.idx[0][1]) { tap_load<CPL>(q.t[0][1], map, i01, lane_bytes); q.idx[0][1] = i01; }
  if (i10 != q.idx[1][0]) { tap_load<CPL>(q.t[1][0], map, i10, lane_bytes); q.idx[1][0] = i10; }
  if (i11 != q.idx[1][1]) { tap_load<CPL>(q.t[1][1], map, i11, lane_bytes); q.idx[1][1] = i11; }
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
  return v + __int_as_float(t);
}

// all-reduce over LPG adjacent lanes (LPG | 16, aligned): same pairing tree as the xor butterfly
template <int LPG>
__device__ __forceinline__ float dpp_group_sum(float v) {
  if (LPG >= 2) v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  if (LPG >= 4) v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  if (LPG >= 8) v = dpp_add<0x141>(v);  // row_half_mirror
  if (LPG >= 16) v = dpp_add<0x140>(v); // row_mirror
  return v;
}

// one (pair, scale): walk the CVW_SEG samples of this slot's segment
template <int CPL, int LPG>
__device__ __forceinline__ void lean_walk(const float* __restrict__ m0, const float* __restrict__ m1,
                                          const float4* __restrict__ wrec /* [js][view a|b][idx|weights] */,
                                          float* __restrict__ cs_group, int cs_stride, int sub, unsigned lane_bytes) {
  static_assert(CVW_SEG % LPG == 0, "segment length must be a multiple of the lanes per channel group");
  float k_dot = 0.0f, k_na = 1.0f, k_nb = 1.0f;  // the (sample, group) triple this lane will turn into a cosine
  PairQuad<CPL> qa, qb;
#pragma unroll
  for (int i = 0; i < 4; ++i) qa.idx[i >> 1][i & 1] = qb.idx[i >> 1][i & 1] = -1;
  for (int js = 0; js < CVW_SEG; ++js) {
    const float4 wa = wrec[js * 4 + 1], wb = wrec[js * 4 + 3];
    quad_update<CPL>(qa, m0, wrec[js * 4 + 0], lane_bytes);
    quad_update<CPL>(qb, m1, wrec[js * 4 + 2], lane_bytes);
    const v2f A00 = {wa.x, wa.x}, A01 = {wa.y, wa.y}, A10 = {wa.z, wa.z}, A11 = {wa.w, wa.w};
    const v2f B00 = {wb.x, wb.x}, B01 = {wb.y, wb.y}, B10 = {wb.z, wb.z}, B11 = {wb.w, wb.w};
    v2f dot2 = {0.f, 0.f}, na2 = {0.f, 0.f}, nb2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CPL / 2; ++k) {
      v2f fa = qa.t[0][0][k] * A00;
      fa = __builtin_elementwise_fma(qa.t[0][1][k], A01, fa);
      fa = __builtin_elementwise_fma(qa.t[1][0][k], A10, fa);
      fa = __builtin_elementwise_fma(qa.t[1][1][k], A11, fa);
      v2f fb = qb.t[0][0][k] * B00;
      fb = __builtin_elementwise_fma(qb.t[0][1][k], B01, fb);
      fb = __builtin_elementwise_fma(qb.t[1][0][k], B10, fb);
      fb = __builtin_elementwise_fma(qb.t[1][1][k], B11, fb);
      dot2 = __builtin_elementwise_fma(fa, fb, dot2);
      na2 = __builtin_elementwise_fma(fa, fa, na2);
      nb2 = __builtin_elementwise_fma(fb, fb, nb2);
    }
    const float dot = dpp_group_sum<LPG>(dot2.x + dot2.y);
    const float na = dpp_group_sum<LPG>(na2.x + na2.y);
    const float nb = dpp_group_sum<LPG>(nb2.x + nb2.y);
    // After the all-reduce the LPG lanes of a channel group hold the same (dot, |a|^2, |b|^2), so evaluating the
    // cosine (two IEEE square roots and a divide, ~45 VALU) in all of them is LPG-fold redundant.  Instead lane u
    // of the group keeps the triple of sample js = jb + u, and the cosines of LPG samples are evaluated together
    // once per LPG steps (CVW_SEG is a multiple of every LPG): every lane then owns one (sample, group) sum.
    const int u = js & (LPG - 1);
    if (LPG == 1 || (sub & (LPG - 1)) == u) {
      k_dot = dot;
      k_na = na;
      k_nb = nb;
    }
    if (u == LPG - 1) {
#if CVW_FAST_COS
      // 1-ulp hardware sqrt / rcp instead of the correctly rounded sequences: |error| <= ~3 ulp of a cosine
      const float da = fmaxf(__builtin_amdgcn_sqrtf(k_na), 1e-8f), db = fmaxf(__builtin_amdgcn_sqrtf(k_nb), 1e-8f);
      const float c = k_dot * __builtin_amdgcn_rcpf(da * db);
#else
      const float da = fmaxf(sqrtf(k_na), 1e-8f), db = fmaxf(sqrtf(k_nb), 1e-8f);
      const float c = k_dot / (da * db);
#endif
      const int js_mine = js - (LPG - 1) + (sub & (LPG - 1));
      __hip_atomic_fetch_add(cs_group + js_mine * cs_stride, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
}

template <int CPL>
__global__ __launch_bounds__(256, (CPL == 16 ? 2 : CVW_WAVES)) void cost_volume_lean_kernel(mnerf_scene sc, mnerf_rays R,
                                                                                   int cond_stride,
                                                                                   float* __restrict__ cond) {
  constexpr int LPS = FEAT_C / CPL;     // lanes per sample slot (8 or 16)
  constexpr int NSLOT = 256 / LPS;      // ray slots per workgroup (32 or 16)
  constexpr int SPL = CVW_SEG / LPS > 0 ? CVW_SEG / LPS : 1;  // pass-1 samples per lane
  extern __shared__ __attribute__((aligned(16))) float cvw_smem[];
  const int V = sc.n_views;
  const int sub = threadIdx.x % LPS;
  const int slot = threadIdx.x / LPS;                               // NSLOT adjacent rays
  const int G0 = sc.n_group[0], G1 = sc.n_scales > 1 ? sc.n_group[1] : 0;
  const int sumG = G0 + G1;
  const int cs_stride = (sumG + 3) & ~3;                            // cosine sums per segment sample in LDS
  // LDS per slot: projections [js][view](u,v) | walk records [js][view a|b][idx|weights] (float4) | cosine sums [js][cs]
  float* uv_lds = cvw_smem + (size_t)slot * CVW_SEG * V * 2;
  float4* wrec_lds = reinterpret_cast<float4*>(cvw_smem + (size_t)NSLOT * CVW_SEG * V * 2) + (size_t)slot * CVW_SEG * 4;
  float* cs_lds = cvw_smem + (size_t)NSLOT * CVW_SEG * (V * 2 + 16) + slot * CVW_SEG * cs_stride;
  const int S = R.n_samples;
  const int P = V * (V - 1) / 2;
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);
  const float inv_pairs = 1.0f / (float)P;
  const int n_seg = (S + CVW_SEG - 1) / CVW_SEG;
  const unsigned lane_bytes = (unsigned)sub * CPL * 4;

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
    const RayGeom g = make_ray(R, ray);

    // ---- pass 1: projections, tap records, colours, masks.  Lane `sub` takes samples sub, sub+LPS, ..
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
    for (int i = sub; i < CVW_SEG * cs_stride; i += LPS) cs_lds[i] = 0.0f;  // this slot's cosine sums
    // slot-local LDS hand-off: the lanes of a slot belong to one wave => wave-level ordering
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
          float* cs_group = cs_lds + goff + sub / lpg;  // this lane's channel group; it owns sample jb + sub % lpg
          // walk records of this (pair, scale): lane `sub` expands samples sub, sub+LPS, .. of views a and b
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the previous walk's reads of wrec_lds are done
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int half = 0; half < SPL; ++half) {
            const int js = sub + LPS * half;
            if (js >= CVW_SEG) break;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
              const int vw = side ? b : a;
              const TapRec t = tap_setup(uv_lds[(js * V + vw) * 2], uv_lds[(js * V + vw) * 2 + 1], fh, fw);
              float4 ri, rw;
              tap_expand(t, fw, ri, rw);
              wrec_lds[(js * 2 + side) * 2] = ri;
              wrec_lds[(js * 2 + side) * 2 + 1] = rw;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          switch (lpg) {
            case 1: lean_walk<CPL, 1>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes); break;
            case 2: lean_walk<CPL, 2>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes); break;
            case 4: lean_walk<CPL, 4>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes); break;
            case 8: lean_walk<CPL, 8>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes); break;
            default: lean_walk<CPL, 16>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes); break;
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
        for (int c = 0; c < sumG; ++c) out[c] = cs_lds[js * cs_stride + c] * inv_pairs;
      }
    }
    __builtin_amdgcn_wave_barrier();  // cs_lds / uv_lds / wrec_lds are rewritten by the next unit
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
    hipLaunchKernelGGL(cost_volume_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       *scene, *rays, cond_stride, cond);
  }
  return mnerf_check_launch("mnerf_cost_volume");
}
