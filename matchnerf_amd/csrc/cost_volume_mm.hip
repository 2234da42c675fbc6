// K1+K2, MATRIX FORM (round 6) — the cost volume's 128-channel interpolation on the matrix pipe.
//
// Replaces the same reference code as cost_volume.hip (MatchNeRF.query_cond_info, /root/reference/models/matchnerf.py:209-293;
// sample_features_by_grid, models/gmflow/utils.py:131-134) for launches over CONTIGUOUS pixels of one target view.
//
// Why another formulation.  The segment walk (cv_walk.hpp) gathers four taps per (ray, sample, map) through the texture path and
// interpolates 128 channels on the vector ALU: round 5's counters showed it co-bound — texture-address units 0.87 busy on
// quarter-full tap instructions, vector ALU 0.82 busy on ~2 400 instructions per 64 rays x (sample, pair, scale).  Neither side
// can be tuned away inside that formulation.  Here the interpolation is a matrix product:
//
//   a wave owns an 8 x 4 PIXEL TILE of the target view (32 rays; lanes n and n + 32 are the two K-halves of ray n) at one depth
//   index.  The 32 rays' bilinear footprints in a source map fall into a handful of aligned 4 x 4 TEXEL CHUNKS (one or two at
//   the DTU shape).  For a chunk,
//        F^T [32 channels x 32 rays] += T^T [32 channels x 16 texels] . W^T [16 texels x 32 rays]
//   is one v_mfma_f32_32x32x16_f16 per product term: T = the chunk's texels (A operand, read ready-made from the OPERAND IMAGE
//   below with two fully coalesced 16-byte loads per lane), W = the rays' bilinear weights scattered over the chunk's 16 texels
//   (B operand; lane (n, half) builds the 8 weights of ray n for the chunk's row pair `half` in registers — no cross-lane
//   traffic: the ray IS the lane).  Split-fp16 arithmetic as in the decoder: T = hi + lo (fp16 each, one power-of-two gain per
//   map), W = hi + lo, three products hi.hi + lo.hi + hi.lo with fp32 accumulation (22-bit operands; the dropped lo.lo term is
//   < 2^-22 of a product).  The accumulators then hold the interpolated features of both maps of the pair for all 128 channels
//   (2 x 4 x 16 registers); the three dot products per channel run on the vector ALU straight from the accumulators, the two
//   K-halves of a ray are folded with v_permlane32_swap (two quantities per swap: lower half-wave ends up with the even
//   channel groups, upper with the odd ones), one lane per (ray, group) evaluates the cosine.
//
//   Per 64 rays x (sample, pair, scale): ~40 tap-free load instructions (was ~140 quarter-full ones), ~800 vector instructions
//   (was ~2 400), ~75 matrix instructions (was 0).  Colours, masks and the projections are the walk kernel's arithmetic.
//
// OPERAND IMAGE (mnerf_cost_volume_operands; caller-owned buffer of mnerf_cost_volume_operand_bytes bytes):
//   header  float gain[2][CVM_MAX_MAPS] | float inv_gain[2][CVM_MAX_MAPS] | u32 absmax_bits[2][CVM_MAX_MAPS]
//   scale s [map = 2 pair + side][row pair rp][x block xb][channel tile ct (4)][hi | lo][32 channels][8 x fp16]
//           the 8 values of (rp, xb, channel) are texels (row 2 rp + r, column 4 xb + c) at index 4 r + c, scaled by the map's
//           gain (largest magnitude in [2^14, 2^15)); rows / columns outside the map and one extra row pair are zero-filled.
//   = 4 KiB per (row pair, x block), the same bytes as the fp32 map.  A chunk = row pairs (p, p + 1) x one x block: lanes 0-31
//   read 512 contiguous bytes per (channel tile, hi | lo) of row pair p, lanes 32-63 of row pair p + 1.
#include <stdlib.h>

#include "cv_walk.hpp"

typedef _Float16 cvm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cvm_h2 __attribute__((ext_vector_type(2)));
typedef float cvm_f2 __attribute__((ext_vector_type(2)));
typedef float cvm_f16 __attribute__((ext_vector_type(16)));
typedef unsigned cvm_u4 __attribute__((ext_vector_type(4)));

#define CVM_MAX_MAPS (MNERF_MAX_VIEWS * (MNERF_MAX_VIEWS - 1))  // 2 sides x 120 pairs
#define CVM_HDR_BYTES 8192
#define CVM_CELL_BYTES 4096  // one (row pair, x block): 8 texels x 128 channels x (hi + lo) fp16
#define CVM_TARGET_EXP 15    // largest |texel| of a map is scaled into [2^14, 2^15)
#define CVM_UV_MAX_VIEWS MNERF_MAX_VIEWS

struct CvmLayout {
  int nrp[2], nxb[2];         // row pairs (incl. the zero one past the end), x blocks
  size_t map_bytes[2], off[2], total;
};

__host__ __device__ inline CvmLayout cvm_layout(const mnerf_scene& sc) {
  CvmLayout L;
  const size_t maps = (size_t)sc.n_views * (sc.n_views - 1);
  size_t o = CVM_HDR_BYTES;
  for (int s = 0; s < 2; ++s) {
    if (s < sc.n_scales) {
      L.nrp[s] = (sc.fh[s] + 1) / 2 + 1;  // + the zero pair a chunk's second half reads past the last row
      L.nxb[s] = (sc.fw[s] + 3) / 4;
      L.map_bytes[s] = (size_t)L.nrp[s] * L.nxb[s] * CVM_CELL_BYTES;
    } else {
      L.nrp[s] = L.nxb[s] = 0;
      L.map_bytes[s] = 0;
    }
    L.off[s] = o;
    o += L.map_bytes[s] * maps;
  }
  L.total = o;
  return L;
}

// ============================================================================ pre-pass 1: largest magnitude per map
__global__ __launch_bounds__(256) void cvm_absmax_kernel(mnerf_scene sc, unsigned* __restrict__ absmax_bits) {
  const int maps = sc.n_views * (sc.n_views - 1);
  const int s = (int)blockIdx.y >= maps ? 1 : 0, m = (int)blockIdx.y - s * maps;
  const size_t n4 = (size_t)sc.fh[s] * sc.fw[s] * (FEAT_C / 4);
  const v4f* src = reinterpret_cast<const v4f*>(sc.feat[s]) + (size_t)m * n4;
  float mx = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const v4f v = src[i];
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  // non-negative floats order as unsigned integers; NaN / inf texels give the map gain 1 (cvm_gain_exp)
  if ((threadIdx.x & 63) == 0) atomicMax(absmax_bits + s * CVM_MAX_MAPS + m, __float_as_uint(mx));
}

// exponent e with 2^e m in [2^14, 2^15); 0 for m = 0 / inf / nan; clamped so that 2^e and 2^-e stay normal fp32 numbers
__device__ __forceinline__ int cvm_gain_exp(float m) {
  if (!(m > 0.0f) || !(m < 3.0e38f)) return 0;
  int e = __builtin_amdgcn_frexp_expf(m);  // m = f 2^e, f in [0.5, 1)
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return CVM_TARGET_EXP - e;
}

// ============================================================================ pre-pass 2: split + re-layout
// one thread per (map, row pair, x block, channel): 8 texels of one channel -> hi8 | lo8
__global__ __launch_bounds__(256) void cvm_split_kernel(mnerf_scene sc, char* __restrict__ opnd) {
  const CvmLayout L = cvm_layout(sc);
  const int maps = sc.n_views * (sc.n_views - 1);
  const int s = (int)blockIdx.y >= maps ? 1 : 0, m = (int)blockIdx.y - s * maps;
  const int fh = sc.fh[s], fw = sc.fw[s];
  const int cell = (int)blockIdx.x * 2 + (int)(threadIdx.x >> 7), ch = threadIdx.x & 127;
  if (cell >= L.nrp[s] * L.nxb[s]) return;
  const int rp = cell / L.nxb[s], xb = cell - rp * L.nxb[s];
  float* gain = reinterpret_cast<float*>(opnd);
  const unsigned* absmax_bits = reinterpret_cast<const unsigned*>(opnd + 2 * 2 * CVM_MAX_MAPS * 4);
  const int e = cvm_gain_exp(__uint_as_float(absmax_bits[s * CVM_MAX_MAPS + m]));
  const float g = ldexpf(1.0f, e);
  if (cell == 0 && ch == 0) {
    gain[s * CVM_MAX_MAPS + m] = g;
    gain[2 * CVM_MAX_MAPS + s * CVM_MAX_MAPS + m] = ldexpf(1.0f, -e);
  }
  const float* src = sc.feat[s] + (size_t)m * fh * fw * FEAT_C + ch;
  cvm_h8 hi, lo;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int y = 2 * rp + r, x = 4 * xb + c;
      float v = 0.0f;
      if (y < fh && x < fw) v = src[((size_t)y * fw + x) * FEAT_C];
      v = v * g;                                        // exact (power of two) unless the product leaves the fp32 range
      v = fminf(fmaxf(v, -65504.0f), 65504.0f);         // (only non-finite texels reach the clamp: NaN -> -65504)
      const _Float16 h = (_Float16)v;                   // RNE
      hi[4 * r + c] = h;
      lo[4 * r + c] = (_Float16)(v - (float)h);
    }
  char* dst = opnd + L.off[s] + (size_t)m * L.map_bytes[s] + (size_t)cell * CVM_CELL_BYTES + (size_t)(ch >> 5) * 1024 + (size_t)(ch & 31) * 16;
  *reinterpret_cast<cvm_h8*>(dst) = hi;
  *reinterpret_cast<cvm_h8*>(dst + 512) = lo;
}

// ============================================================================ the kernel
// wave-uniform minimum / maximum of a per-lane int: row all-reduce with DPP, the four rows through scalar registers
template <bool MAX>
__device__ __forceinline__ int cvm_wave_minmax(int v) {
#define CVM_MM_STEP(CTRL)                                                        \
  {                                                                              \
    const int t = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);      \
    v = MAX ? max(v, t) : min(v, t);                                             \
  }
  CVM_MM_STEP(0xB1)   // quad_perm [1,0,3,2]
  CVM_MM_STEP(0x4E)   // quad_perm [2,3,0,1]
  CVM_MM_STEP(0x141)  // row_half_mirror
  CVM_MM_STEP(0x140)  // row_mirror
#undef CVM_MM_STEP
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
            d = __builtin_amdgcn_readlane(v, 48);
  return MAX ? max(max(a, b), max(c, d)) : min(min(a, b), min(c, d));
}

// v_permlane32_swap: upper half of a <-> lower half of b
__device__ __forceinline__ void cvm_swap32(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1" : "+v"(a), "+v"(b));
}
// (x, y) per-lane partial sums of two quantities whose other half lives in lane ^ 32: lower half-wave <- total of x,
// upper half-wave <- total of y
__device__ __forceinline__ float cvm_fold_pair(float x, float y) {
  cvm_swap32(x, y);
  return x + y;
}

// bilinear footprint of one ray in one map: top-left texel and the four weights as two packed fp16 pairs (row y0 / row y0 + 1:
// (weight of x0, weight of x0 + 1)), hi and lo terms.  bilin_setup()'s arithmetic (cv_walk.hpp).
struct CvmTap {
  int x0, y0;
  unsigned top_hi, bot_hi, top_lo, bot_lo;
};

__device__ __forceinline__ unsigned cvm_pack_h2(float a, float b) {
  const cvm_f2 ab = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(ab, cvm_h2));  // v_cvt_pk_f16_f32 (RNE)
}
__device__ __forceinline__ float cvm_h_lo(unsigned p) { return (float)__builtin_bit_cast(cvm_h2, p).x; }
__device__ __forceinline__ float cvm_h_hi(unsigned p) { return (float)__builtin_bit_cast(cvm_h2, p).y; }

__device__ __forceinline__ CvmTap cvm_tap(float u, float v, int h, int w) {
  const float gx = u * 2.0f - 1.0f, gy = v * 2.0f - 1.0f;
  float x = ((gx + 1.0f) * 0.5f) * (float)(w - 1);
  float y = ((gy + 1.0f) * 0.5f) * (float)(h - 1);
  x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
  y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
  const float x0f = floorf(x), y0f = floorf(y);
  const float fx = x - x0f, fy = y - y0f;
  const float w00 = (1.0f - fx) * (1.0f - fy), w01 = fx * (1.0f - fy), w10 = (1.0f - fx) * fy, w11 = fx * fy;
  CvmTap t;
  t.x0 = (int)x0f;
  t.y0 = (int)y0f;
  t.top_hi = cvm_pack_h2(w00, w01);
  t.bot_hi = cvm_pack_h2(w10, w11);
  t.top_lo = cvm_pack_h2(w00 - cvm_h_lo(t.top_hi), w01 - cvm_h_hi(t.top_hi));
  t.bot_lo = cvm_pack_h2(w10 - cvm_h_lo(t.bot_hi), w11 - cvm_h_hi(t.bot_hi));
  return t;
}

// wave-uniform chunk range of one map: row pairs [p_lo, p_hi], x blocks [xb_lo, xb_hi]
struct CvmBox {
  int p_lo, p_hi, xb_lo, xb_hi;
};
// (a footprint's second row / column beyond the map's last one carries weight exactly 0 - the coordinate was clamped onto the
// last texel - and is left out, as bilin_setup()'s min(x0 + 1, w - 1) does: chunks never start past the map)
__device__ __forceinline__ CvmBox cvm_box(const CvmTap& t, int h, int w) {
  CvmBox b;
  b.p_lo = cvm_wave_minmax<false>(t.y0) >> 1;
  b.p_hi = min(cvm_wave_minmax<true>(t.y0) + 1, h - 1) >> 1;
  b.xb_lo = cvm_wave_minmax<false>(t.x0) >> 2;
  b.xb_hi = min(cvm_wave_minmax<true>(t.x0) + 1, w - 1) >> 2;
  return b;
}

// one row of a chunk for this lane: the packed weight pair E = (w(x0), w(x0 + 1)) of that row placed at columns dx, dx + 1 of
// the chunk's four: two dwords (columns 0,1 | columns 2,3).  dx in [-1, 3] (checked by the caller: E = 0 otherwise).
__device__ __forceinline__ void cvm_strip(unsigned E, int dx, unsigned& d0, unsigned& d1) {
  const unsigned e = dx < 0 ? (E >> 16) : E;
  const unsigned long long s = (unsigned long long)e << ((dx < 0 ? 0 : 16 * dx) & 63);
  d0 = (unsigned)s;
  d1 = (unsigned)(s >> 32);
}

// B operand (hi and lo) of one chunk for this lane: rows r0 + 2 half, r0 + 2 half + 1; columns c0 .. c0 + 3
__device__ __forceinline__ void cvm_weights(const CvmTap& t, int r0, int c0, int half, cvm_h8& bh, cvm_h8& bl) {
  const int dy = t.y0 - (r0 + 2 * half);  // 0: (top, bottom); -1: (bottom, -); 1: (-, top)
  const int dx = t.x0 - c0;
  const bool xok = (unsigned)(dx + 1) <= 4u;
  const bool s0_top = xok && dy == 0, s0_bot = xok && dy == -1, s1_bot = s0_top, s1_top = xok && dy == 1;
  const unsigned e0h = s0_top ? t.top_hi : (s0_bot ? t.bot_hi : 0u), e1h = s1_bot ? t.bot_hi : (s1_top ? t.top_hi : 0u);
  const unsigned e0l = s0_top ? t.top_lo : (s0_bot ? t.bot_lo : 0u), e1l = s1_bot ? t.bot_lo : (s1_top ? t.top_lo : 0u);
  cvm_u4 H, Lo;
  unsigned a, b;
  cvm_strip(e0h, dx, a, b), H.x = a, H.y = b;
  cvm_strip(e1h, dx, a, b), H.z = a, H.w = b;
  cvm_strip(e0l, dx, a, b), Lo.x = a, Lo.y = b;
  cvm_strip(e1l, dx, a, b), Lo.z = a, Lo.w = b;
  bh = __builtin_bit_cast(cvm_h8, H);
  bl = __builtin_bit_cast(cvm_h8, Lo);
}

__device__ __forceinline__ cvm_f16 cvm_mfma(cvm_h8 a, cvm_h8 b, cvm_f16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// F^T (all 128 channels x the wave's 32 rays) of one map: every occupied chunk of the rays' footprints.
// `map` = the map's operand image, wave-uniform; acc[ct] = channels 32 ct .. 32 ct + 31.
__device__ __forceinline__ void cvm_interp(cvm_f16 (&acc)[4], const char* __restrict__ map, int nxb, const CvmTap& t, const CvmBox& box,
                                           int n, int half) {
  bool first = true;
  for (int p = box.p_lo; p <= box.p_hi; p += 2) {
    for (int xb = box.xb_lo; xb <= box.xb_hi; ++xb) {
      const int r0 = 2 * p, c0 = 4 * xb;
      // a ray touches the chunk iff its 2 x 2 footprint intersects rows [r0, r0 + 4) x columns [c0, c0 + 4)
      const bool mine = (unsigned)(t.y0 - r0 + 1) <= 4u && (unsigned)(t.x0 - c0 + 1) <= 4u;
      if (__builtin_amdgcn_ballot_w64(mine) == 0) continue;
      const unsigned voff = (unsigned)((p + half) * nxb + xb) * (unsigned)CVM_CELL_BYTES + (unsigned)n * 16u;
      const char* src = map + voff;
      cvm_u4 ah[4], al[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        ah[ct] = *reinterpret_cast<const cvm_u4*>(src + ct * 1024);
        al[ct] = *reinterpret_cast<const cvm_u4*>(src + ct * 1024 + 512);
      }
      cvm_h8 bh, bl;
      cvm_weights(t, r0, c0, half, bh, bl);
      if (first) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          cvm_f16 z;
#pragma unroll
          for (int i = 0; i < 16; ++i) z[i] = 0.0f;
          const cvm_h8 h = __builtin_bit_cast(cvm_h8, ah[ct]), l = __builtin_bit_cast(cvm_h8, al[ct]);
          z = cvm_mfma(l, bh, z);
          z = cvm_mfma(h, bl, z);
          acc[ct] = cvm_mfma(h, bh, z);
        }
        first = false;
      } else {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const cvm_h8 h = __builtin_bit_cast(cvm_h8, ah[ct]), l = __builtin_bit_cast(cvm_h8, al[ct]);
          acc[ct] = cvm_mfma(l, bh, acc[ct]);
          acc[ct] = cvm_mfma(h, bl, acc[ct]);
          acc[ct] = cvm_mfma(h, bh, acc[ct]);
        }
      }
    }
  }
}

// one (pair, scale) for the wave's 32 rays at one depth index: adds the cosines of this lane's groups (group 2 i + half in
// slot i) to cacc.  G = channel groups of the scale (1, 2, 4, 8).
__device__ __forceinline__ void cvm_unit(float (&cacc)[4], const char* __restrict__ map_a, const char* __restrict__ map_b, int fh, int fw,
                                         int nxb, const CvmTap& ta, const CvmTap& tb, float inv_ga, float inv_gb, int G, int n, int half) {
  cvm_f16 fa[4], fb[4];
  cvm_interp(fa, map_a, nxb, ta, cvm_box(ta, fh, fw), n, half);
  cvm_interp(fb, map_b, nxb, tb, cvm_box(tb, fh, fw), n, half);
  // three dot products per 16-channel granule (this lane's 8 channels of it: registers 8 (q & 1) .. + 7 of tile q >> 1)
  float dot[8], na[8], nb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int ct = q >> 1, r0 = (q & 1) * 8;
    float d = fa[ct][r0] * fb[ct][r0], a = fa[ct][r0] * fa[ct][r0], b = fb[ct][r0] * fb[ct][r0];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      d = __builtin_fmaf(fa[ct][r0 + r], fb[ct][r0 + r], d);
      a = __builtin_fmaf(fa[ct][r0 + r], fa[ct][r0 + r], a);
      b = __builtin_fmaf(fb[ct][r0 + r], fb[ct][r0 + r], b);
    }
    dot[q] = d, na[q] = a, nb[q] = b;
  }
  // granules -> groups (wave-uniform): 8 / G granules each
  if (G <= 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dot[q] = dot[2 * q] + dot[2 * q + 1], na[q] = na[2 * q] + na[2 * q + 1], nb[q] = nb[2 * q] + nb[2 * q + 1];
  }
  if (G <= 2) {
#pragma unroll
    for (int q = 0; q < 2; ++q) dot[q] = dot[2 * q] + dot[2 * q + 1], na[q] = na[2 * q] + na[2 * q + 1], nb[q] = nb[2 * q] + nb[2 * q + 1];
  }
  if (G <= 1) dot[0] = dot[0] + dot[1], na[0] = na[0] + na[1], nb[0] = nb[0] + nb[1], dot[1] = na[1] = nb[1] = 0.0f;
  const float sdot = inv_ga * inv_gb, sa = inv_ga * inv_ga, sb = inv_gb * inv_gb;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (2 * i < G) {  // wave-uniform
      const float d = cvm_fold_pair(dot[2 * i], dot[2 * i + 1]) * sdot;
      const float a = cvm_fold_pair(na[2 * i], na[2 * i + 1]) * sa;
      const float b = cvm_fold_pair(nb[2 * i], nb[2 * i + 1]) * sb;
      // CosineSimilarity: x1 / max(|x1|, eps) . x2 / max(|x2|, eps), eps = 1e-8  (max(sqrt(a), eps) = sqrt(max(a, eps^2)))
      const float c = d * (__builtin_amdgcn_rsqf(fmaxf(a, 1e-16f)) * __builtin_amdgcn_rsqf(fmaxf(b, 1e-16f)));
      cacc[i] += c;
    }
  }
}

struct CvmGrid {
  int tile_y0, ntx, n_tiles, nsg, spw;  // first tile row, tiles per row, tiles, sample groups per tile, samples per wave and item
};

__global__ __launch_bounds__(256, 2) void cost_volume_mm_kernel(mnerf_scene sc, mnerf_rays R, int cond_stride, float* __restrict__ cond,
                                                                const char* __restrict__ opnd, CvmGrid grid) {
  __shared__ float uv_lds[4][CVM_UV_MAX_VIEWS][32][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  const int S = R.n_samples, V = sc.n_views;
  const int W = R.width, H = R.height;
  const int maps = V * (V - 1);
  const CvmLayout L = cvm_layout(sc);
  const float* inv_gain = reinterpret_cast<const float*>(opnd) + 2 * CVM_MAX_MAPS;

  // XCD-major contiguous runs of items (see cost_volume_kernel): item = (tile, sample group), sample groups of a tile adjacent
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin;
  const int tile = item / grid.nsg, sg = item - tile * grid.nsg;
  const int tyi = tile / grid.ntx, txi = tile - tyi * grid.ntx;

  // this lane's ray: pixel (8 txi + n % 8, 4 (tile_y0 + tyi) + n / 8); lanes outside the image or the launch's pixel range work
  // on the nearest pixel inside and store nothing
  const int px = txi * 8 + (n & 7), py = (grid.tile_y0 + tyi) * 4 + (n >> 3);
  int pix = min(py, H - 1) * W + min(px, W - 1);
  const bool ray_live = px < W && py < H && pix >= R.ray_begin && pix < R.ray_begin + R.n_rays;
  pix = max(R.ray_begin, min(pix, R.ray_begin + R.n_rays - 1));
  const int ray = pix - R.ray_begin;
  const RayGeom g = make_ray(R, ray);
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  const int G0 = sc.n_group[0], G1 = sc.n_scales > 1 ? sc.n_group[1] : 0;
  const int sumG = G0 + G1;
  const float inv_pairs = 1.0f / (float)(V * (V - 1) / 2);
  float (*uv)[32][2] = uv_lds[wave];

  for (int k = 0; k < grid.spw; ++k) {
    const int j = (sg * grid.spw + k) * 4 + wave;
    if (j >= S) break;  // wave-uniform
    const float d = sample_depth(R, ray, j);
    float wx, wy, wz;
    ray_point(g, d, wx, wy, wz);
    float* out = cond + ((size_t)ray * S + j) * cond_stride;

    // ---- pass 1 (cv_pass1's arithmetic): projections, colours, masks; half-wave `half` takes views half, half + 2, ..
    cvw_handoff();  // the previous sample's reads of uv are done
    for (int v = half; v < V; v += 2) {
      float u, w_, z;
      project(sc.views[v], wx, wy, wz, wm1, hm1, u, w_, z);
      uv[v][n][0] = u;
      uv[v][n][1] = w_;
      const Bilin b = bilin_setup(u, w_, H, W);
      const float4* img = reinterpret_cast<const float4*>(sc.images) + (size_t)v * H * W;
      const float4 t00 = img[b.o00], t01 = img[b.o01], t10 = img[b.o10], t11 = img[b.o11];
      const float gx = u * 2.0f - 1.0f, gy = w_ * 2.0f - 1.0f;
      const float m = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;
      if (ray_live) {
        out[sumG + 3 * v + 0] = t00.x * b.w00 + t01.x * b.w01 + t10.x * b.w10 + t11.x * b.w11;
        out[sumG + 3 * v + 1] = t00.y * b.w00 + t01.y * b.w01 + t10.y * b.w10 + t11.y * b.w11;
        out[sumG + 3 * v + 2] = t00.z * b.w00 + t01.z * b.w01 + t10.z * b.w10 + t11.z * b.w11;
        out[sumG + 3 * V + v] = m;
      }
    }
    if (ray_live && half == 0) {
      const int dc = sumG + 4 * V;
      out[dc] = 1.0f;  // constant input of the packed FiLM bias column
      for (int c = dc + 1; c < cond_stride; ++c) out[c] = 0.0f;
    }
    cvw_handoff();

    // ---- pass 2: one unit per (pair, scale); slot i of a lane = group 2 i + half
    float c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (int a = 0; a < V - 1; ++a) {
      const float ua = uv[a][n][0], va = uv[a][n][1];
      for (int b = a + 1; b < V; ++b, ++p) {
        const float ub = uv[b][n][0], vb = uv[b][n][1];
        {
          const char* ma = opnd + L.off[0] + (size_t)(2 * p) * L.map_bytes[0];
          cvm_unit(c0, ma, ma + L.map_bytes[0], sc.fh[0], sc.fw[0], L.nxb[0], cvm_tap(ua, va, sc.fh[0], sc.fw[0]), cvm_tap(ub, vb, sc.fh[0], sc.fw[0]),
                   inv_gain[2 * p], inv_gain[2 * p + 1], G0, n, half);
        }
        if (sc.n_scales > 1) {
          const char* ma = opnd + L.off[1] + (size_t)(2 * p) * L.map_bytes[1];
          cvm_unit(c1, ma, ma + L.map_bytes[1], sc.fh[1], sc.fw[1], L.nxb[1], cvm_tap(ua, va, sc.fh[1], sc.fw[1]), cvm_tap(ub, vb, sc.fh[1], sc.fw[1]),
                   inv_gain[CVM_MAX_MAPS + 2 * p], inv_gain[CVM_MAX_MAPS + 2 * p + 1], G1, n, half);
        }
      }
    }
    if (ray_live) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gi = 2 * i + half;
        if (gi < G0) out[gi] = c0[i] * inv_pairs;
        if (gi < G1) out[G0 + gi] = c1[i] * inv_pairs;
      }
    }
  }
  (void)maps;
}

// ============================================================================ host
extern "C" int64_t mnerf_cost_volume_operand_bytes(const mnerf_scene* scene) {
  if (!scene || scene->n_views < 2 || scene->n_views > MNERF_MAX_VIEWS || scene->n_scales < 1 || scene->n_scales > 2) return -1;
  return (int64_t)cvm_layout(*scene).total;
}

extern "C" int mnerf_cost_volume_operands(const mnerf_scene* scene, void* opnd, void* stream) {
  MNERF_REQUIRE(scene && opnd, MNERF_E_NULL, "mnerf_cost_volume_operands: NULL argument");
  MNERF_REQUIRE(scene->n_views >= 2 && scene->n_views <= MNERF_MAX_VIEWS, MNERF_E_RANGE, "mnerf_cost_volume_operands: n_views=%d",
                scene->n_views);
  MNERF_REQUIRE(scene->n_scales == 1 || scene->n_scales == 2, MNERF_E_RANGE, "mnerf_cost_volume_operands: n_scales=%d", scene->n_scales);
  MNERF_REQUIRE(mnerf_aligned16(opnd), MNERF_E_ALIGN, "mnerf_cost_volume_operands: buffer not 16B aligned");
  for (int s = 0; s < scene->n_scales; ++s) {
    MNERF_REQUIRE(scene->feat[s], MNERF_E_NULL, "mnerf_cost_volume_operands: feat[%d] is NULL", s);
    MNERF_REQUIRE(mnerf_aligned16(scene->feat[s]), MNERF_E_ALIGN, "mnerf_cost_volume_operands: feat[%d] not 16B aligned", s);
    MNERF_REQUIRE(scene->fh[s] >= 1 && scene->fw[s] >= 1, MNERF_E_RANGE, "mnerf_cost_volume_operands: feature map %d is %dx%d", s,
                  scene->fh[s], scene->fw[s]);
  }
  const CvmLayout L = cvm_layout(*scene);
  const int maps = scene->n_views * (scene->n_views - 1);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(opnd, 0, CVM_HDR_BYTES, st);
  if (e != hipSuccess) {
    mnerf_set_error("mnerf_cost_volume_operands: memset failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  unsigned* absmax_bits = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(opnd) + 2 * 2 * CVM_MAX_MAPS * 4);
  hipLaunchKernelGGL(cvm_absmax_kernel, dim3(32, (unsigned)(maps * scene->n_scales)), dim3(256), 0, st, *scene, absmax_bits);
  int cells = 0;
  for (int s = 0; s < scene->n_scales; ++s) cells = cells > L.nrp[s] * L.nxb[s] ? cells : L.nrp[s] * L.nxb[s];
  hipLaunchKernelGGL(cvm_split_kernel, dim3((unsigned)((cells + 1) / 2), (unsigned)(maps * scene->n_scales)), dim3(256), 0, st, *scene,
                     reinterpret_cast<char*>(opnd));
  return mnerf_check_launch("mnerf_cost_volume_operands");
}

// the matrix form takes: an operand image, contiguous pixels of one pose (no ray_idx, no pose table); any view count / groups
bool mnerf_cost_volume_mm_applies(const mnerf_scene* scene, const mnerf_rays* rays) {
  return scene->feat_op != nullptr && rays->ray_idx == nullptr && rays->pose_table == nullptr;
}

int mnerf_cost_volume_mm_launch(const mnerf_scene* scene, const mnerf_rays* rays, int cond_stride, float* cond, void* stream) {
  MNERF_REQUIRE(mnerf_aligned16(scene->feat_op), MNERF_E_ALIGN, "mnerf_cost_volume: feat_op not 16B aligned");
  const int W = rays->width;
  const int row_first = rays->ray_begin / W, row_last = (rays->ray_begin + rays->n_rays - 1) / W;
  CvmGrid g;
  g.tile_y0 = row_first / 4;
  g.ntx = (W + 7) / 8;
  g.n_tiles = g.ntx * (row_last / 4 - g.tile_y0 + 1);
  int spw = mnerf_tune().cv_mm_spw;
  if (spw <= 0) spw = 4;
  g.spw = spw;
  g.nsg = (rays->n_samples + 4 * spw - 1) / (4 * spw);
  const long long items = (long long)g.n_tiles * g.nsg;
  MNERF_REQUIRE(items < (1ll << 31), MNERF_E_RANGE, "mnerf_cost_volume: %lld work items", items);
  hipLaunchKernelGGL(cost_volume_mm_kernel, dim3((unsigned)items), dim3(256), 0, (hipStream_t)stream, *scene, *rays, cond_stride,
                     cond, reinterpret_cast<const char*>(scene->feat_op), g);
  return mnerf_check_launch("mnerf_cost_volume");
}
