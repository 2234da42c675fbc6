// Shared device code of the cost volume (K1+K2): bilinear set-up and the register-quad segment walk.
// Included by cost_volume.hip (stand-alone kernel, 16-sample walks, rows to global memory) and by decoder.hip
// (fused ray-chunk kernel: the workgroup produces the conditioning rows of its own tile in LDS, 8-sample walks).
#pragma once
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


// one bilinear value: the products summed in tap order, as an explicit multiply + FMA chain (what the compiler made of
// a w00 + b w01 + c w10 + d w11 in the walk kernels; spelled out so that every cost-volume kernel gives the same bits whatever
// the surrounding code looks like)
__device__ __forceinline__ float bilin4(float a, float b, float c, float d, const Bilin& w) {
  return __builtin_fmaf(d, w.w11, __builtin_fmaf(c, w.w10, __builtin_fmaf(b, w.w01, a * w.w00)));
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
// Interpolation / dot-product arithmetic on channel pairs, WITHOUT packed-fp32 vector instructions (one v_fma_f32 / v_mul_f32
// per channel; the two source files that include this header are also compiled with -fno-slp-vectorize, build.py).
// Measured on MI355X (tools/exp/race_probe.py): v_pk_fma_f32 / v_pk_mul_f32 of a wave lose their
// result in lanes 48-63 now and then while ANOTHER wave of the same SIMD issues v_mfma_f32_32x32x16_{f16,bf16} — the
// one-launch ray chunk (walk and MFMA trunk in co-resident workgroups) and this kernel next to the decoder on a second
// stream both produced a few wrong conditioning rows per launch, always in the last 16 lanes of a wave, one walk step at a
// time; never next to the exact-f32 decoder, never alone, never in this form.  The packed form (CVW_PK=1, probe builds) is
// not faster either: the kernel is bound by the texture-address unit (9.9 vs 10.1 ms per frame).
#ifndef CVW_PK
#define CVW_PK 0
#endif
#if !CVW_PK
__device__ __forceinline__ float cvw_opaque(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b) { return v2f{cvw_opaque(a.x * b.x), cvw_opaque(a.y * b.y)}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) {
  return v2f{cvw_opaque(__builtin_fmaf(a.x, b.x, c.x)), cvw_opaque(__builtin_fmaf(a.y, b.y, c.y))};
}
#else
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b) { return a * b; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#endif

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

__device__ __forceinline__ TapRec tap_setup_xy(float u, float v, int h, int w, int& x0_out, int& y0_out) {  // bilin_setup()'s arithmetic
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
  x0_out = x0;
  y0_out = y0;
  return r;
}
__device__ __forceinline__ TapRec tap_setup(float u, float v, int h, int w) {
  int x0, y0;
  return tap_setup_xy(u, v, h, w, x0, y0);
}

#ifndef CVW_RELOAD_ANY
#define CVW_RELOAD_ANY 0
#endif
#ifndef CVW_PROBE
#define CVW_PROBE 0  // race probes (tools/exp/race_probe.py): 3 shuffles instead of DPP, 4 unconditional tap reloads,
#endif               // 5 / 7 hard waits at the slot-local LDS hand-offs, 6 no LDS atomic
// slot-local LDS hand-off between lanes of one wave
__device__ __forceinline__ void cvw_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#if CVW_PROBE == 5 || CVW_PROBE == 7
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#endif
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifndef CVW_SEG
#define CVW_SEG 16  // samples per walk of the stand-alone kernel (the fused ray-chunk kernel walks 8)
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
#ifdef CVW_DEBUG_WAIT
  // probe build: every tap load is complete before anything else is issued (would hide a result consumed early)
#pragma unroll
  for (int k = 0; k < CPL / 2; ++k) asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[k]) : : "memory");
#endif
}

// Tile form (cost_volume_tile_kernel): the texels a 16-ray x SEG-sample tile touches in the two maps of a (pair, scale) are
// staged in LDS once by the whole workgroup; a walk record then names a tap set's texel by an ENCODED index:
//   enc >= 0 : byte offset of the staged texel in the workgroup's texel area.  A staged texel is stored as two 256-byte
//              halves, [16 lanes][first 16 B of the lane's 32-byte channel run] then [16 lanes][second 16 B], so that the 16
//              lanes of a slot read 256 contiguous bytes per ds_read_b128 (the global layout would be a 32-byte lane stride);
//   enc <  0 : ~enc is the texel index in the global map (its LDS place was taken by another texel) - loaded as before.
template <int CPL>
__device__ __forceinline__ void tap_load_tile(v2f (&t)[CPL / 2], const float* __restrict__ map, const char* tex, int enc,
                                              unsigned lane_bytes, unsigned lane_lds) {
  static_assert(CPL == 8, "the staged texel layout is written for 16 lanes x 8 channels");
  if (enc >= 0) {
    const v4f a = *reinterpret_cast<const v4f*>(tex + (unsigned)enc + lane_lds);
    const v4f b = *reinterpret_cast<const v4f*>(tex + (unsigned)enc + 256u + lane_lds);
    t[0] = a.lo;
    t[1] = a.hi;
    t[2] = b.lo;
    t[3] = b.hi;
  } else {
    tap_load<CPL>(t, map, ~enc, lane_bytes);
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
#if CVW_PROBE == 4
  tap_load<CPL>(q.t[0][0], map, i00, lane_bytes);
  tap_load<CPL>(q.t[0][1], map, i01, lane_bytes);
  tap_load<CPL>(q.t[1][0], map, i10, lane_bytes);
  tap_load<CPL>(q.t[1][1], map, i11, lane_bytes);
  return;
#endif
#if CVW_RELOAD_ANY
  // A tap set is reloaded by the WHOLE wave as soon as ANY of its slots has crossed a texel boundary: the texture-address
  // unit charges a load instruction 16 cycles whatever its lane mask (tools/exp/ta_mask.hip), and the four slots of a wave
  // cross at different steps, so per-slot reloads are four quarter-full instructions where one full one does (the slots
  // that had not moved re-read their texel from L1).
  if (__builtin_amdgcn_ballot_w64(i00 != q.idx[0][0])) { tap_load<CPL>(q.t[0][0], map, i00, lane_bytes); q.idx[0][0] = i00; }
  if (__builtin_amdgcn_ballot_w64(i01 != q.idx[0][1])) { tap_load<CPL>(q.t[0][1], map, i01, lane_bytes); q.idx[0][1] = i01; }
  if (__builtin_amdgcn_ballot_w64(i10 != q.idx[1][0])) { tap_load<CPL>(q.t[1][0], map, i10, lane_bytes); q.idx[1][0] = i10; }
  if (__builtin_amdgcn_ballot_w64(i11 != q.idx[1][1])) { tap_load<CPL>(q.t[1][1], map, i11, lane_bytes); q.idx[1][1] = i11; }
  return;
#endif
  if (i00 != q.idx[0][0]) { tap_load<CPL>(q.t[0][0], map, i00, lane_bytes); q.idx[0][0] = i00; }
  if (i01 != q.idx[0][1]) { tap_load<CPL>(q.t[0][1], map, i01, lane_bytes); q.idx[0][1] = i01; }
  if (i10 != q.idx[1][0]) { tap_load<CPL>(q.t[1][0], map, i10, lane_bytes); q.idx[1][0] = i10; }
  if (i11 != q.idx[1][1]) { tap_load<CPL>(q.t[1][1], map, i11, lane_bytes); q.idx[1][1] = i11; }
}

template <int CPL>
__device__ __forceinline__ void quad_update_tile(PairQuad<CPL>& q, const float* __restrict__ map, const char* tex, const float4 ix,
                                                 unsigned lane_bytes, unsigned lane_lds) {
  const int i00 = __float_as_int(ix.x), i01 = __float_as_int(ix.y), i10 = __float_as_int(ix.z), i11 = __float_as_int(ix.w);
  if (i00 != q.idx[0][0]) { tap_load_tile<CPL>(q.t[0][0], map, tex, i00, lane_bytes, lane_lds); q.idx[0][0] = i00; }
  if (i01 != q.idx[0][1]) { tap_load_tile<CPL>(q.t[0][1], map, tex, i01, lane_bytes, lane_lds); q.idx[0][1] = i01; }
  if (i10 != q.idx[1][0]) { tap_load_tile<CPL>(q.t[1][0], map, tex, i10, lane_bytes, lane_lds); q.idx[1][0] = i10; }
  if (i11 != q.idx[1][1]) { tap_load_tile<CPL>(q.t[1][1], map, tex, i11, lane_bytes, lane_lds); q.idx[1][1] = i11; }
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
  return v + __int_as_float(t);
}

// all-reduce over LPG adjacent lanes (LPG | 16, aligned): same pairing tree as the xor butterfly
template <int LPG>
__device__ __forceinline__ float dpp_group_sum(float v) {
#if CVW_PROBE == 3
#pragma unroll
  for (int m = 1; m < LPG; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
#endif
  if (LPG >= 2) v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  if (LPG >= 4) v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  if (LPG >= 8) v = dpp_add<0x141>(v);  // row_half_mirror
  if (LPG >= 16) v = dpp_add<0x140>(v); // row_mirror
  return v;
}

// one (pair, scale): walk the SEG samples of this slot's segment
template <int CPL, int LPG, int SEG, bool TILE = false>
__device__ __forceinline__ void lean_walk(const float* __restrict__ m0, const float* __restrict__ m1,
                                          const float4* __restrict__ wrec /* [js][view a|b][idx|weights] */,
                                          float* __restrict__ cs_group, int cs_stride, int sub, unsigned lane_bytes,
                                          const char* tex = nullptr /* TILE: the workgroup's staged texels */) {
  static_assert(SEG % LPG == 0, "segment length must be a multiple of the lanes per channel group");
  float k_dot = 0.0f, k_na = 1.0f, k_nb = 1.0f;  // the (sample, group) triple this lane will turn into a cosine
  PairQuad<CPL> qa, qb;
#pragma unroll
  for (int i = 0; i < 4; ++i) qa.idx[i >> 1][i & 1] = qb.idx[i >> 1][i & 1] = TILE ? (int)0x80000000 : -1;  // names no texel
  const unsigned lane_lds = (unsigned)sub * 16u;
  for (int js = 0; js < SEG; ++js) {
    const float4 wa = wrec[js * 4 + 1], wb = wrec[js * 4 + 3];
    if constexpr (TILE) {
      quad_update_tile<CPL>(qa, m0, tex, wrec[js * 4 + 0], lane_bytes, lane_lds);
      quad_update_tile<CPL>(qb, m1, tex, wrec[js * 4 + 2], lane_bytes, lane_lds);
    } else {
      quad_update<CPL>(qa, m0, wrec[js * 4 + 0], lane_bytes);
      quad_update<CPL>(qb, m1, wrec[js * 4 + 2], lane_bytes);
    }
    const v2f A00 = {wa.x, wa.x}, A01 = {wa.y, wa.y}, A10 = {wa.z, wa.z}, A11 = {wa.w, wa.w};
    const v2f B00 = {wb.x, wb.x}, B01 = {wb.y, wb.y}, B10 = {wb.z, wb.z}, B11 = {wb.w, wb.w};
    v2f dot2 = {0.f, 0.f}, na2 = {0.f, 0.f}, nb2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CPL / 2; ++k) {
      v2f fa = pk_mul(qa.t[0][0][k], A00);
      fa = pk_fma(qa.t[0][1][k], A01, fa);
      fa = pk_fma(qa.t[1][0][k], A10, fa);
      fa = pk_fma(qa.t[1][1][k], A11, fa);
      v2f fb = pk_mul(qb.t[0][0][k], B00);
      fb = pk_fma(qb.t[0][1][k], B01, fb);
      fb = pk_fma(qb.t[1][0][k], B10, fb);
      fb = pk_fma(qb.t[1][1][k], B11, fb);
      dot2 = pk_fma(fa, fb, dot2);
      na2 = pk_fma(fa, fa, na2);
      nb2 = pk_fma(fb, fb, nb2);
    }
    const float dot = dpp_group_sum<LPG>(dot2.x + dot2.y);
    const float na = dpp_group_sum<LPG>(na2.x + na2.y);
    const float nb = dpp_group_sum<LPG>(nb2.x + nb2.y);
    // After the all-reduce the LPG lanes of a channel group hold the same (dot, |a|^2, |b|^2), so evaluating the
    // cosine (two IEEE square roots and a divide, ~45 VALU) in all of them is LPG-fold redundant.  Instead lane u
    // of the group keeps the triple of sample js = jb + u, and the cosines of LPG samples are evaluated together
    // once per LPG steps (SEG is a multiple of every LPG): every lane then owns one (sample, group) sum.
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
#if CVW_PROBE == 6
      cs_group[js_mine * cs_stride] += c;
#else
      __hip_atomic_fetch_add(cs_group + js_mine * cs_stride, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
    }
  }
}


// Walk of the tile form WITHOUT the register tap cache, for a wave whose walk records all name staged texels (enc >= 0; the
// kernel checks that per wave and walk, and takes lean_walk<.., TILE> otherwise): with the texels in LDS a step simply reads its
// eight taps (16 ds_read_b128 per lane) - no compares, no divergent reloads, straight-line code - and the reads of step js+1
// are issued before the arithmetic of step js, so the LDS round trip is hidden even at two waves per SIMD.
// Arithmetic and its order are lean_walk's.
template <int CPL>
struct StepTaps {
  v2f a[4][CPL / 2], b[4][CPL / 2];  // [EE, EO, OE, OO][channel pair] of map a / map b
  float4 wa, wb;
};

template <int CPL>
__device__ __forceinline__ void step_fetch(StepTaps<CPL>& t, const float4* __restrict__ wrec, int js, const char* tex,
                                           unsigned lane_lds) {
  const float4 ia = wrec[js * 4 + 0], ib = wrec[js * 4 + 2];
  t.wa = wrec[js * 4 + 1];
  t.wb = wrec[js * 4 + 3];
  const unsigned ea[4] = {__float_as_uint(ia.x), __float_as_uint(ia.y), __float_as_uint(ia.z), __float_as_uint(ia.w)};
  const unsigned eb[4] = {__float_as_uint(ib.x), __float_as_uint(ib.y), __float_as_uint(ib.z), __float_as_uint(ib.w)};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const v4f x = *reinterpret_cast<const v4f*>(tex + ea[k] + lane_lds);
    const v4f y = *reinterpret_cast<const v4f*>(tex + ea[k] + 256u + lane_lds);
    t.a[k][0] = x.lo, t.a[k][1] = x.hi, t.a[k][2] = y.lo, t.a[k][3] = y.hi;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const v4f x = *reinterpret_cast<const v4f*>(tex + eb[k] + lane_lds);
    const v4f y = *reinterpret_cast<const v4f*>(tex + eb[k] + 256u + lane_lds);
    t.b[k][0] = x.lo, t.b[k][1] = x.hi, t.b[k][2] = y.lo, t.b[k][3] = y.hi;
  }
}

template <int CPL, int LPG, int SEG>
__device__ __forceinline__ void tile_walk(const float* __restrict__ m0, const float* __restrict__ m1,
                                          const float4* __restrict__ wrec, float* __restrict__ cs_group, int cs_stride, int sub,
                                          unsigned lane_bytes, const char* tex) {
  static_assert(SEG % LPG == 0, "segment length must be a multiple of the lanes per channel group");
  float k_dot = 0.0f, k_na = 1.0f, k_nb = 1.0f;
  const unsigned lane_lds = (unsigned)sub * 16u;
  StepTaps<CPL> buf[2];
  step_fetch<CPL>(buf[0], wrec, 0, tex, lane_lds);
#pragma unroll
  for (int js = 0; js < SEG; ++js) {
    StepTaps<CPL>& t = buf[js & 1];
    if (js + 1 < SEG) step_fetch<CPL>(buf[(js + 1) & 1], wrec, js + 1, tex, lane_lds);
    const float4 wa = t.wa, wb = t.wb;
    const v2f A00 = {wa.x, wa.x}, A01 = {wa.y, wa.y}, A10 = {wa.z, wa.z}, A11 = {wa.w, wa.w};
    const v2f B00 = {wb.x, wb.x}, B01 = {wb.y, wb.y}, B10 = {wb.z, wb.z}, B11 = {wb.w, wb.w};
    v2f dot2 = {0.f, 0.f}, na2 = {0.f, 0.f}, nb2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CPL / 2; ++k) {
      v2f fa = pk_mul(t.a[0][k], A00);
      fa = pk_fma(t.a[1][k], A01, fa);
      fa = pk_fma(t.a[2][k], A10, fa);
      fa = pk_fma(t.a[3][k], A11, fa);
      v2f fb = pk_mul(t.b[0][k], B00);
      fb = pk_fma(t.b[1][k], B01, fb);
      fb = pk_fma(t.b[2][k], B10, fb);
      fb = pk_fma(t.b[3][k], B11, fb);
      dot2 = pk_fma(fa, fb, dot2);
      na2 = pk_fma(fa, fa, na2);
      nb2 = pk_fma(fb, fb, nb2);
    }
    const float dot = dpp_group_sum<LPG>(dot2.x + dot2.y);
    const float na = dpp_group_sum<LPG>(na2.x + na2.y);
    const float nb = dpp_group_sum<LPG>(nb2.x + nb2.y);
    const int u = js & (LPG - 1);
    if (LPG == 1 || (sub & (LPG - 1)) == u) {
      k_dot = dot;
      k_na = na;
      k_nb = nb;
    }
    if (u == LPG - 1) {  // see lean_walk: every lane turns one (sample, group) triple into a cosine
      const float da = fmaxf(sqrtf(k_na), 1e-8f), db = fmaxf(sqrtf(k_nb), 1e-8f);
      const float c = k_dot / (da * db);
      const int js_mine = js - (LPG - 1) + (sub & (LPG - 1));
      __hip_atomic_fetch_add(cs_group + js_mine * cs_stride, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
}

// LDS floats one slot needs for a segment of SEG samples: projections [js][view](u,v) | walk records
// [js][view a|b][idx|weights] (float4) | cosine sums [js][cs_stride]
__host__ __device__ inline int cv_slot_lds_floats(int seg, int n_views, int sum_groups) {
  return seg * (n_views * 2 + 16 + ((sum_groups + 3) & ~3));
}

// One walk UNIT: a slot of LPS = 128/CPL lanes (inside one wave) produces the conditioning rows of SEG consecutive
// samples j0 .. j0+SEG-1 of ray `ray`: pass 1 (projections, colours, masks, constant column), then one register-quad
// walk per (view pair, scale), then the averaged cosines.  `row0` points at the row of sample j0 (global memory in
// the stand-alone kernel, LDS in the fused ray-chunk kernel), rows are `cond_stride` floats apart; samples at or
// beyond S are computed on the last real sample and not written.  uv / wrec / cs: this slot's LDS scratch.
// Only wave-level synchronisation inside (the lanes of a slot belong to one wave).
// row stores of the stand-alone form (the NT flag marked the rows for non-temporal stores in a round-2 experiment: measured slower,
// tools/exp/patches/decoder_experiments_until_r5.patch)
template <bool NT>
__device__ __forceinline__ void cv_store(float* p, float v) {
  *p = v;
}

// pass 1 of a walk unit: projections (to uv_lds), colours, masks and the constant column of SEG samples of one ray
template <int CPL, int SEG, bool NT>
__device__ __forceinline__ void cv_pass1(const mnerf_scene& sc, const mnerf_rays& R, int ray, bool ray_live, int j0,
                                         float* __restrict__ row0, int cond_stride, float* __restrict__ uv_lds, int sub) {
  constexpr int LPS = FEAT_C / CPL;
  constexpr int SPL = SEG / LPS > 0 ? SEG / LPS : 1;  // pass-1 samples per lane
  const int V = sc.n_views;
  const int S = R.n_samples;
  const int sumG = sc.n_group[0] + (sc.n_scales > 1 ? sc.n_group[1] : 0);
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);
  const RayGeom g = make_ray(R, ray);
  // ---- pass 1: projections, tap records, colours, masks.  Lane `sub` takes samples sub, sub+LPS, ..
#pragma unroll
  for (int half = 0; half < SPL; ++half) {
    const int js = sub + LPS * half;
    if (js >= SEG) break;
    const int j = min(j0 + js, S - 1);
    const bool live = ray_live && (j0 + js < S);
    const float d = sample_depth(R, ray, j);
    float px, py, pz;
    ray_point(g, d, px, py, pz);
    float* out = row0 + (size_t)js * cond_stride;
    for (int v = 0; v < V; ++v) {
      float u, w_, z;
      project(sc.views[v], px, py, pz, wm1, hm1, u, w_, z);
      if (uv_lds) {
        uv_lds[(js * V + v) * 2 + 0] = u;
        uv_lds[(js * V + v) * 2 + 1] = w_;
      }
      const Bilin b = bilin_setup(u, w_, R.height, R.width);
      const float4* img = reinterpret_cast<const float4*>(sc.images) + (size_t)v * R.height * R.width;
      float4 t00 = img[b.o00], t01 = img[b.o01], t10 = img[b.o10], t11 = img[b.o11];
#if CVW_PROBE == 9
      asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7" : "+v"(t00.x), "+v"(t00.y), "+v"(t00.z), "+v"(t01.x), "+v"(t01.y), "+v"(t01.z),
                   "+v"(t10.x), "+v"(t10.y), "+v"(t10.z), "+v"(t11.x), "+v"(t11.y), "+v"(t11.z) : : "memory");
#endif
      const float gx = u * 2.0f - 1.0f, gy = w_ * 2.0f - 1.0f;
      const float m = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;
      if (live) {
        cv_store<NT>(out + sumG + 3 * v + 0, bilin4(t00.x, t01.x, t10.x, t11.x, b));
        cv_store<NT>(out + sumG + 3 * v + 1, bilin4(t00.y, t01.y, t10.y, t11.y, b));
        cv_store<NT>(out + sumG + 3 * v + 2, bilin4(t00.z, t01.z, t10.z, t11.z, b));
        cv_store<NT>(out + sumG + 3 * V + v, m);
      }
    }
    if (live) {
      const int dc = sumG + 4 * V;
      cv_store<NT>(out + dc, 1.0f);  // constant input of the packed FiLM bias column
      for (int c = dc + 1; c < cond_stride; ++c) cv_store<NT>(out + c, 0.0f);
    }
  }
}

// the averaged cosines of a unit: lane `sub` writes samples sub, sub+LPS, ..
template <int CPL, int SEG, bool NT>
__device__ __forceinline__ void cv_write_cosines(const mnerf_rays& R, bool ray_live, int j0, float* __restrict__ row0,
                                                 int cond_stride, const float* __restrict__ cs_lds, int cs_stride, int sumG,
                                                 float inv_pairs, int sub) {
  constexpr int LPS = FEAT_C / CPL;
  constexpr int SPL = SEG / LPS > 0 ? SEG / LPS : 1;
  const int S = R.n_samples;
#pragma unroll
  for (int half = 0; half < SPL; ++half) {
    const int js = sub + LPS * half;
    if (js < SEG && ray_live && (j0 + js < S)) {
      float* out = row0 + (size_t)js * cond_stride;
#pragma clang loop vectorize(disable) interleave(disable)  // (a vectorised multiply is a packed-fp32 instruction)
      for (int c = 0; c < sumG; ++c) cv_store<NT>(out + c, cs_lds[js * cs_stride + c] * inv_pairs);
    }
  }
}

// UVPAIR: the projections of a segment are not kept for all V views ([js][V] (u,v) in uv_lds) but re-evaluated per view pair
// for its two views ([js][2]): 2 P instead of V projections per sample (+1 % of a unit's instructions at 10 views) for
// 2 V - 4 fewer LDS floats per sample — at 10 views that is what lets a fourth workgroup share the CU (cost_volume.hip).
// PAIR BLOCKS (many views).  At 10 views the 45 pairs' maps are 1.18 GB — 4.6 x the Infinity Cache — and a unit that walks all
// pairs keeps all of them live: round 3 measured 60 GB fetched per 65,536-ray launch for those 1.18 GB (52 % L2 hits).  The
// stand-alone kernel can therefore be launched once per BLOCK of pairs [pair_begin, pair_end) over ALL rays of the launch
// (cost_volume.hip: 8 pairs = 210 MB of maps): the first block also does pass 1, every block continues the per-sample cosine
// sums where the previous one stopped — they travel through the rows' cosine columns as RAW sums, the last block scales them
// by 1 / pairs — so every sum is still accumulated pair by pair in the reference's order: same bits as one launch.
template <int CPL, int SEG, bool NT = false, bool UVPAIR = false>
__device__ __forceinline__ void cv_walk_unit(const mnerf_scene& sc, const mnerf_rays& R, int ray, bool ray_live, int j0,
                                             float* __restrict__ row0, int cond_stride, float* __restrict__ uv_lds,
                                             float4* __restrict__ wrec_lds, float* __restrict__ cs_lds, int sub,
                                             int pair_begin = 0, int pair_end = 0x7fffffff) {
  constexpr int LPS = FEAT_C / CPL;
  constexpr int SPL = SEG / LPS > 0 ? SEG / LPS : 1;  // pass-1 samples per lane
  const int V = sc.n_views;
  const int UVS = UVPAIR ? 2 : V;  // views per sample in uv_lds
  const int G0 = sc.n_group[0], G1 = sc.n_scales > 1 ? sc.n_group[1] : 0;
  const int sumG = G0 + G1;
  const int cs_stride = (sumG + 3) & ~3;
  const float inv_pairs = 1.0f / (float)(V * (V - 1) / 2);
  const unsigned lane_bytes = (unsigned)sub * CPL * 4;

  const bool first_block = pair_begin <= 0, last_block = pair_end >= V * (V - 1) / 2;
  if (first_block) {
    cv_pass1<CPL, SEG, NT>(sc, R, ray, ray_live, j0, row0, cond_stride, UVPAIR ? nullptr : uv_lds, sub);
    for (int i = sub; i < SEG * cs_stride; i += LPS) cs_lds[i] = 0.0f;  // this slot's cosine sums
  } else {
    static_assert(UVPAIR || SEG > 0, "");
    // a later pair block: the raw sums of the blocks before it (lane `sub` wrote samples sub, sub + LPS, .. of this segment)
#pragma unroll
    for (int half = 0; half < SPL; ++half) {
      const int js = sub + LPS * half;
      if (js >= SEG) break;
      const bool have = ray_live && (j0 + js < R.n_samples);
      const float* in = row0 + (size_t)js * cond_stride;
      for (int c = 0; c < sumG; ++c) cs_lds[js * cs_stride + c] = have ? in[c] : 0.0f;
    }
  }
  // slot-local LDS hand-off: the lanes of a slot belong to one wave => wave-level ordering
  cvw_handoff();

  // ---- pass 2: walk the segment once per (pair, scale) with the two texel quads in registers
  int p = 0;
  for (int a = 0; a < V - 1; ++a) {
    for (int b = a + 1; b < V; ++b, ++p) {
      if (p < pair_begin || p >= pair_end) continue;  // not in this launch's pair block
      if constexpr (UVPAIR) {  // this pair's two projections of every sample of the segment (cv_pass1's arithmetic)
        const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);
        const RayGeom g = make_ray(R, ray);
        cvw_handoff();  // the previous pair's records have been expanded
#pragma unroll
        for (int half = 0; half < SPL; ++half) {
          const int js = sub + LPS * half;
          if (js >= SEG) break;
          const float d = sample_depth(R, ray, min(j0 + js, R.n_samples - 1));
          float px, py, pz;
          ray_point(g, d, px, py, pz);
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            float u, w_, z;
            project(sc.views[side ? b : a], px, py, pz, wm1, hm1, u, w_, z);
            uv_lds[(js * 2 + side) * 2 + 0] = u;
            uv_lds[(js * 2 + side) * 2 + 1] = w_;
          }
        }
      }
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
        cvw_handoff();  // the previous walk's reads of wrec_lds are done
#pragma unroll
        for (int half = 0; half < SPL; ++half) {
          const int js = sub + LPS * half;
          if (js >= SEG) break;
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const int vw = UVPAIR ? side : (side ? b : a);
            const TapRec t = tap_setup(uv_lds[(js * UVS + vw) * 2], uv_lds[(js * UVS + vw) * 2 + 1], fh, fw);
            float4 ri, rw;
            tap_expand(t, fw, ri, rw);
            wrec_lds[(js * 2 + side) * 2] = ri;
            wrec_lds[(js * 2 + side) * 2 + 1] = rw;
          }
        }
        cvw_handoff();
        switch (lpg) {
          case 1: lean_walk<CPL, 1, SEG>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes); break;
          case 2: lean_walk<CPL, 2, SEG>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes); break;
          case 4: lean_walk<CPL, 4, SEG>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes); break;
          case 8: lean_walk<CPL, 8, SEG>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes); break;
          default:
            if constexpr (SEG % 16 == 0) lean_walk<CPL, 16, SEG>(m0, m1, wrec_lds, cs_group, cs_stride, sub, lane_bytes);
            break;  // (a 16-lane group needs 16-sample walks: callers with shorter walks require G >= 2)
        }
      }
    }
  }
  cvw_handoff();
  cv_write_cosines<CPL, SEG, NT>(R, ray_live, j0, row0, cond_stride, cs_lds, cs_stride, sumG, last_block ? inv_pairs : 1.0f, sub);
  __builtin_amdgcn_wave_barrier();  // cs_lds / uv_lds / wrec_lds are rewritten by the next unit
}
