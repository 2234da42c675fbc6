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
//   scale s [map = 2 pair + side][row pair rp][column x][hi | lo][128 channels] dwords: one dword = the two fp16 values of texels
//           (row 2 rp, column x) and (row 2 rp + 1, column x) of that channel, scaled by the map's gain (largest magnitude in
//           [2^14, 2^15)); a row pair is fw + 3 columns wide (three zero columns) and one zero row pair follows the last.
//   = 1 KiB per (row pair, column), the same bytes as the fp32 map.  A chunk = row pairs (p, p + 1) x FOUR CONSECUTIVE COLUMNS
//   starting at ANY column: lanes 0-31 read 128 contiguous bytes per (column, hi | lo, channel tile) of row pair p, lanes 32-63 of
//   row pair p + 1; the four dwords of a lane (one per column) are the A operand as they arrive (K index = 2 column + row).
//   Anchoring a tile's chunks at its leftmost texel column - not at an aligned block of four - is what makes ONE chunk per
//   (depth index, map) the rule at the DTU shape (tools/exp/cvmm_chunks.py: 1.00 chunks against 1.53 with aligned blocks).
#include <stdlib.h>

#include "cv_walk.hpp"

typedef _Float16 cvm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cvm_h2 __attribute__((ext_vector_type(2)));
typedef float cvm_f2 __attribute__((ext_vector_type(2)));
typedef float cvm_f16 __attribute__((ext_vector_type(16)));
typedef unsigned cvm_u4 __attribute__((ext_vector_type(4)));

#define CVM_MAX_MAPS (MNERF_MAX_VIEWS * (MNERF_MAX_VIEWS - 1))  // 2 sides x 120 pairs
#define CVM_HDR_BYTES 8192
#define CVM_COL_BYTES 1024   // one (row pair, column): 2 texels x 128 channels x (hi + lo) fp16
#define CVM_TARGET_EXP 15    // largest |texel| of a map is scaled into [2^14, 2^15)
#define CVM_UV_MAX_VIEWS MNERF_MAX_VIEWS

struct CvmLayout {
  int nrp[2], rs[2];          // row pairs (incl. the zero one past the end), columns per row pair (fw + 3 zero columns)
  size_t map_bytes[2], off[2], total;
};

__host__ __device__ inline CvmLayout cvm_layout(const mnerf_scene& sc) {
  CvmLayout L;
  const size_t maps = (size_t)sc.n_views * (sc.n_views - 1);
  size_t o = CVM_HDR_BYTES;
  for (int s = 0; s < 2; ++s) {
    if (s < sc.n_scales) {
      L.nrp[s] = (sc.fh[s] + 1) / 2 + 1;  // + the zero pair a chunk's second half reads past the last row
      L.rs[s] = sc.fw[s] + 3;
      L.map_bytes[s] = (size_t)L.nrp[s] * L.rs[s] * CVM_COL_BYTES;
    } else {
      L.nrp[s] = L.rs[s] = 0;
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
// one thread per (map, row pair, column, channel): the two texels of a column -> one hi dword, one lo dword
__global__ __launch_bounds__(256) void cvm_split_kernel(mnerf_scene sc, char* __restrict__ opnd) {
  const CvmLayout L = cvm_layout(sc);
  const int maps = sc.n_views * (sc.n_views - 1);
  const int s = (int)blockIdx.y >= maps ? 1 : 0, m = (int)blockIdx.y - s * maps;
  const int fh = sc.fh[s], fw = sc.fw[s];
  const int cell = (int)blockIdx.x * 2 + (int)(threadIdx.x >> 7), ch = threadIdx.x & 127;
  if (cell >= L.nrp[s] * L.rs[s]) return;
  const int rp = cell / L.rs[s], x = cell - rp * L.rs[s];
  float* gain = reinterpret_cast<float*>(opnd);
  const unsigned* absmax_bits = reinterpret_cast<const unsigned*>(opnd + 2 * 2 * CVM_MAX_MAPS * 4);
  const int e = cvm_gain_exp(__uint_as_float(absmax_bits[s * CVM_MAX_MAPS + m]));
  const float g = ldexpf(1.0f, e);
  if (cell == 0 && ch == 0) {
    gain[s * CVM_MAX_MAPS + m] = g;
    gain[2 * CVM_MAX_MAPS + s * CVM_MAX_MAPS + m] = ldexpf(1.0f, -e);
  }
  const float* src = sc.feat[s] + (size_t)m * fh * fw * FEAT_C + ch;
  cvm_h2 hi, lo;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int y = 2 * rp + r;
    float v = 0.0f;
    if (y < fh && x < fw) v = src[((size_t)y * fw + x) * FEAT_C];
    v = v * g;                                        // exact (power of two) unless the product leaves the fp32 range
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);         // (only non-finite texels reach the clamp: NaN -> -65504)
    const _Float16 h = (_Float16)v;                   // RNE
    hi[r] = h;
    lo[r] = (_Float16)(v - (float)h);
  }
  char* dst = opnd + L.off[s] + (size_t)m * L.map_bytes[s] + (size_t)cell * CVM_COL_BYTES + (size_t)ch * 4;
  *reinterpret_cast<cvm_h2*>(dst) = hi;
  *reinterpret_cast<cvm_h2*>(dst + 512) = lo;
}

// ============================================================================ the kernel
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

// bilinear footprint of one ray in one map: top-left texel (xy = y0 << 16 | x0) and the four weights as two packed fp16 pairs
// (column x0 / column x0 + 1: (weight of row y0, weight of row y0 + 1)), hi and lo terms.  bilin_setup()'s arithmetic
// (cv_walk.hpp).
struct CvmTap {
  unsigned xy;
  unsigned c0_hi, c1_hi, c0_lo, c1_lo;
};

__device__ __forceinline__ unsigned cvm_pack_h2(float a, float b) {
  const cvm_f2 ab = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(ab, cvm_h2));  // v_cvt_pk_f16_f32 (RNE)
}
// w - half(hpk.lo16 / hi16) in one v_fma_mix_f32 (split_f16.hpp: resid_lo / resid_hi)
__device__ __forceinline__ float cvm_resid_lo(float v, unsigned hpk) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(hpk));
  return r;
}
__device__ __forceinline__ float cvm_resid_hi(float v, unsigned hpk) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(hpk));
  return r;
}

__device__ __forceinline__ CvmTap cvm_tap(float u, float v, int h, int w, int& x0, int& y0) {
  const float gx = u * 2.0f - 1.0f, gy = v * 2.0f - 1.0f;
  float x = ((gx + 1.0f) * 0.5f) * (float)(w - 1);
  float y = ((gy + 1.0f) * 0.5f) * (float)(h - 1);
  x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
  y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
  const float x0f = floorf(x), y0f = floorf(y);
  const float fx = x - x0f, fy = y - y0f;
  const float w00 = (1.0f - fx) * (1.0f - fy), w01 = fx * (1.0f - fy), w10 = (1.0f - fx) * fy, w11 = fx * fy;
  CvmTap t;
  x0 = (int)x0f;
  y0 = (int)y0f;
  t.xy = ((unsigned)y0 << 16) | (unsigned)x0;
  t.c0_hi = cvm_pack_h2(w00, w10);
  t.c1_hi = cvm_pack_h2(w01, w11);
  t.c0_lo = cvm_pack_h2(cvm_resid_lo(w00, t.c0_hi), cvm_resid_hi(w10, t.c0_hi));
  t.c1_lo = cvm_pack_h2(cvm_resid_lo(w01, t.c1_hi), cvm_resid_hi(w11, t.c1_hi));
  return t;
}

// Wave-uniform chunk range of one map.  Chunks are addressed on a grid anchored at the rays' smallest row pair and leftmost
// texel column: chunk (cr, cx) = row pairs p_lo + 2 cr, + 1 and columns xb_lo + 4 cx .. + 3  (xb_lo is a COLUMN, any alignment).  `mask` holds one bit per chunk of the first 8 x 8 window
// of that grid (bit 8 cr + cx) that some ray's footprint touches: the chunk loop walks its set bits with scalar instructions.
// A footprint wider than the window (`big`: magnifying or degenerate projections) takes the general loop over [p_lo, p_hi] x
// [xb_lo, xb_hi].  (A footprint's second row / column beyond the map's last one carries weight exactly 0 - the coordinate was
// clamped onto the last texel - and is left out, as bilin_setup()'s min(x0 + 1, w - 1) does: chunks never start past the map.)
struct CvmBox {
  int p_lo, xb_lo;
  unsigned long long mask;
  int big;
};

// One SIDE of a unit = one map of a (pair, scale): what the chunk loop needs, all wave-uniform (scalar registers); the rays'
// footprints (CvmTap, per lane) are read from the wave's LDS scratch when the side becomes the current one.
struct CvmSide {
  const char* map;  // the map's operand image
  int rs;           // columns per row pair in the operand image
  int item;         // view * n_scales + scale: index of the footprints / chunk range in the scratch
  CvmBox box;
  int p, xb;        // the side's first chunk
};

// a ray touches chunk (row pairs p, p + 1; columns xb .. xb + 3) iff its 2 x 2 footprint intersects rows [2 p, 2 p + 4) x
// columns [xb, xb + 4)   (general loop only)
__device__ __forceinline__ bool cvm_occupied(const CvmTap& t, int p, int xb) {
  const int y0 = (int)(t.xy >> 16), x0 = (int)(t.xy & 0xffffu);
  const bool mine = (unsigned)(y0 - 2 * p + 1) <= 4u && (unsigned)(x0 - xb + 1) <= 4u;
  return __builtin_amdgcn_ballot_w64(mine) != 0;
}

// lowest set bit of a non-zero mask -> chunk position; the bit is cleared
__device__ __forceinline__ void cvm_pop(unsigned long long& m, const CvmBox& box, int& p, int& xb) {
  const int bit = __builtin_ctzll(m);
  m &= m - 1;
  p = box.p_lo + 2 * (bit >> 3);
  xb = box.xb_lo + 4 * (bit & 7);
}

// B operand (hi and lo) of one chunk for this lane: rows 2 (p + half), 2 (p + half) + 1; columns xb .. xb + 3; operand dword c =
// (weight of the upper row, weight of the lower row) in column xb + c  (K index 2 c + row, as the A operand).
// A column's packed pair C = (w(y0), w(y0 + 1)) lands in the dword of column dx (C0) / dx + 1 (C1), shifted by the row offset dy
// of the footprint's top row in the lane's row pair: every output dword is ONE v_perm_b32 of (C0, C1) with a byte selector
// that depends on (dy, dx) only (0x0c = constant zero, so a footprint that misses the chunk gives zeros by itself).  The four
// selectors of a (dy, dx) come from a 35-entry table in LDS with one ds_read_b128.
#define CVM_LUT_ROWS 5  // dy clamped to [-2, 2]
#define CVM_LUT_COLS 7  // dx clamped to [-2, 4]
#define CVM_LUT_BYTES 576
__device__ inline unsigned cvm_lut_word(int dyv, int dxv, int c) {
  const int t = c - dxv;  // 0: column x0 (C0 = first source of v_perm_b32: bytes 4-7), 1: column x0 + 1 (C1: bytes 0-3)
  if (t != 0 && t != 1) return 0x0c0c0c0cu;
  const unsigned lo = t == 0 ? 0x0504u : 0x0100u, hi = t == 0 ? 0x0706u : 0x0302u;  // selectors of the pair's two halves
  // dy = 0: the lane's rows are (y0, y0 + 1) -> (upper, lower); dy = 1: (-, y0) -> (0, upper); dy = -1: (y0 + 1, -) -> (lower, 0)
  if (dyv == 0) return (hi << 16) | lo;
  if (dyv == 1) return (lo << 16) | 0x0c0cu;
  if (dyv == -1) return (0x0c0cu << 16) | hi;
  return 0x0c0c0c0cu;
}
__device__ __forceinline__ void cvm_lut_init(unsigned* lut) {  // whole workgroup; the caller synchronises
  for (int t = threadIdx.x; t < CVM_LUT_ROWS * CVM_LUT_COLS * 4; t += blockDim.x) {
    const int e = t >> 2, c = t & 3;
    lut[t] = cvm_lut_word(e / CVM_LUT_COLS - 2, e % CVM_LUT_COLS - 2, c);
  }
}

__device__ __forceinline__ int cvm_med3(int a, int lo, int hi) { return max(lo, min(a, hi)); }

__device__ __forceinline__ void cvm_weights(const CvmTap& t, int p, int xb, int half, const cvm_u4* __restrict__ lut, cvm_h8& bh,
                                            cvm_h8& bl) {
  const int y0 = (int)(t.xy >> 16), x0 = (int)(t.xy & 0xffffu);
  const int dy = y0 - 2 * (p + half);
  const int dx = x0 - xb;
  const cvm_u4 sel = lut[(cvm_med3(dy, -2, 2) + 2) * CVM_LUT_COLS + cvm_med3(dx, -2, 4) + 2];
  cvm_u4 H, Lo;
  H.x = __builtin_amdgcn_perm(t.c0_hi, t.c1_hi, sel.x), H.y = __builtin_amdgcn_perm(t.c0_hi, t.c1_hi, sel.y);
  H.z = __builtin_amdgcn_perm(t.c0_hi, t.c1_hi, sel.z), H.w = __builtin_amdgcn_perm(t.c0_hi, t.c1_hi, sel.w);
  Lo.x = __builtin_amdgcn_perm(t.c0_lo, t.c1_lo, sel.x), Lo.y = __builtin_amdgcn_perm(t.c0_lo, t.c1_lo, sel.y);
  Lo.z = __builtin_amdgcn_perm(t.c0_lo, t.c1_lo, sel.z), Lo.w = __builtin_amdgcn_perm(t.c0_lo, t.c1_lo, sel.w);
  bh = __builtin_bit_cast(cvm_h8, H);
  bl = __builtin_bit_cast(cvm_h8, Lo);
}

#if CVM_EXP == 5
__device__ __forceinline__ cvm_f16 cvm_mfma(cvm_h8 a, cvm_h8 b, cvm_f16 c) {
  c[0] += (float)a[0] * (float)b[0], c[5] += (float)a[2] * (float)b[3], c[9] += (float)a[4] * (float)b[5], c[14] += (float)a[6] * (float)b[7];
  return c;
}
#else
__device__ __forceinline__ cvm_f16 cvm_mfma(cvm_h8 a, cvm_h8 b, cvm_f16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
#endif

// A operands of a chunk for this lane (lanes 0-31: row pair p, lanes 32-63: row pair p + 1): + 1024 column + 512 (hi | lo) +
// 128 channel tile.  Wave-uniform part (scalar registers) + the lane's offset inside the chunk (one register per side)
__device__ __forceinline__ const char* cvm_chunk_base(const CvmSide& s, int p, int xb) {
#if CVM_EXP == 6
  return s.map + (size_t)((unsigned)((p & 1) * s.rs + (xb & 3)) * (unsigned)CVM_COL_BYTES);
#endif
  return s.map + (size_t)((unsigned)(p * s.rs + xb) * (unsigned)CVM_COL_BYTES);
}
__device__ __forceinline__ unsigned cvm_lane_off(int rs, int n, int half) {
  return (unsigned)(half * rs) * (unsigned)CVM_COL_BYTES + (unsigned)n * 4u;
}
// the A operands of channel tile ct: hi = the four columns' hi dwords, lo likewise (16 dword loads with immediate offsets)
__device__ __forceinline__ void cvm_load_ct(cvm_u4& h, cvm_u4& l, const char* src, int ct) {
  const char* q = src + ct * 128;
  h.x = *reinterpret_cast<const unsigned*>(q), h.y = *reinterpret_cast<const unsigned*>(q + 1024);
  h.z = *reinterpret_cast<const unsigned*>(q + 2048), h.w = *reinterpret_cast<const unsigned*>(q + 3072);
  l.x = *reinterpret_cast<const unsigned*>(q + 512), l.y = *reinterpret_cast<const unsigned*>(q + 1536);
  l.z = *reinterpret_cast<const unsigned*>(q + 2560), l.w = *reinterpret_cast<const unsigned*>(q + 3584);
}

// ---------------------------------------------------------------------------------------------------------------------------
// One UNIT = one (pair, scale) for the wave's 32 rays at one depth index, CHANNEL TILE BY CHANNEL TILE and software-pipelined:
//   the A operands of BOTH sides and all four 32-channel tiles live in registers (ha, la, hb, lb: 64 registers) and are
//   re-requested FOR THE NEXT UNIT tile by tile, as soon as the matrix instructions that read them have been issued - every
//   operand load has a whole unit (several thousand cycles) to arrive, instead of one side's matrix run (~400);
//   the matrix instructions of tile t + 1 (two accumulators of 16 registers per side, double-buffered: 64 instead of 128) are
//   issued in the same straight-line block as the dot products of tile t, so that ONE wave keeps the matrix pipe and the
//   vector ALU busy at the same time (before: 24 matrix instructions, then ~290 vector instructions, the two waves of a SIMD
//   rarely in complementary phases: removing the matrix instructions altogether did not change the kernel's time).
// Removal experiments behind this form (tools/exp/patches/README.md, 3 views, 7.1 ms): no matrix instructions 7.3 ms, no dot
// products 6.1, every operand load a cache hit 6.5, no row write-out 6.5, no units at all 1.1.
// A unit whose footprints need more than one chunk per side (rare at the DTU shape: tools/exp/cvmm_chunks.py) takes the same
// code with the further chunks added per tile (operands loaded on demand, weights rebuilt).

__device__ __forceinline__ cvm_f16 cvm_mfma3(cvm_u4 ah, cvm_u4 al, cvm_h8 bh, cvm_h8 bl, cvm_f16 c) {
  const cvm_h8 h = __builtin_bit_cast(cvm_h8, ah), l = __builtin_bit_cast(cvm_h8, al);
  c = cvm_mfma(l, bh, c);
  c = cvm_mfma(h, bl, c);
  return cvm_mfma(h, bh, c);
}

// every chunk of a side for channel tile ct, operands loaded on demand (the unit copy for sides with more than one chunk)
__device__ __forceinline__ void cvm_all_chunks(cvm_f16& acc, const CvmSide& s, const CvmTap& tap, unsigned loff, int ct,
                                               const cvm_u4* __restrict__ lut, int half, int fh, int fw) {
  cvm_h8 bh, bl;
  cvm_u4 hu, lu;
  if (!s.box.big) {
    unsigned long long m = s.box.mask;
    while (m) {
      int p, xb;
      cvm_pop(m, s.box, p, xb);
      cvm_load_ct(hu, lu, cvm_chunk_base(s, p, xb) + loff, ct);
      cvm_weights(tap, p, xb, half, lut, bh, bl);
      acc = cvm_mfma3(hu, lu, bh, bl, acc);
    }
    return;
  }
  // a footprint wider than the 8 x 8 chunk window: every occupied chunk of the rays' range
  int ymax = (int)(tap.xy >> 16), xmax = (int)(tap.xy & 0xffffu);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ymax = max(ymax, __shfl_xor(ymax, off, 64)), xmax = max(xmax, __shfl_xor(xmax, off, 64));
  const int p_hi = __builtin_amdgcn_readfirstlane(min(ymax + 1, fh - 1) >> 1), x_hi = __builtin_amdgcn_readfirstlane(min(xmax + 1, fw - 1));
  for (int p = s.box.p_lo; p <= p_hi; p += 2)
    for (int xb = s.box.xb_lo; xb <= x_hi; xb += 4) {
      if (!cvm_occupied(tap, p, xb)) continue;
      cvm_load_ct(hu, lu, cvm_chunk_base(s, p, xb) + loff, ct);
      cvm_weights(tap, p, xb, half, lut, bh, bl);
      acc = cvm_mfma3(hu, lu, bh, bl, acc);
    }
}

// One cosine from the two half-wave partial sums of (dot, |a|^2, |b|^2) of TWO groups x / y: lower half-wave <- group x,
// upper half-wave <- group y.  CosineSimilarity: x1 / max(|x1|, eps) . x2 / max(|x2|, eps), eps = 1e-8, on the UN-scaled maps:
// with the maps' gains ga, gb in the sums, dot / (ga gb) / (max(|a| / ga, eps) max(|b| / gb, eps))
// = dot rsqrt(max(|a|^2, (eps ga)^2)) rsqrt(max(|b|^2, (eps gb)^2)): the gains only move the clamps (ea2, eb2, wave-uniform).
__device__ __forceinline__ float cvm_cos_pair(float dx, float dy, float ax, float ay, float bx, float by, float ea2, float eb2) {
  const float d = cvm_fold_pair(dx, dy);
  const float a = cvm_fold_pair(ax, ay);
  const float b = cvm_fold_pair(bx, by);
  return d * (__builtin_amdgcn_rsqf(fmaxf(a, ea2)) * __builtin_amdgcn_rsqf(fmaxf(b, eb2)));
}

// three dot products per 16-channel granule of one channel tile (this lane's 8 channels of a granule: registers 0-7 / 8-15)
__device__ __forceinline__ void cvm_dots(const cvm_f16& fa, const cvm_f16& fb, float* dot, float* na, float* nb) {
#pragma unroll
  for (int g2 = 0; g2 < 2; ++g2) {
    const int r0 = 8 * g2;
#if CVM_EXP == 4
    dot[g2] = fa[r0] + fb[r0 + 1], na[g2] = fa[r0 + 2], nb[g2] = fb[r0 + 3];
#else
    float d = fa[r0] * fb[r0], a = fa[r0] * fa[r0], b = fb[r0] * fb[r0];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      d = __builtin_fmaf(fa[r0 + r], fb[r0 + r], d);
      a = __builtin_fmaf(fa[r0 + r], fa[r0 + r], a);
      b = __builtin_fmaf(fb[r0 + r], fb[r0 + r], b);
    }
    dot[g2] = d, na[g2] = a, nb[g2] = b;
#endif
  }
}

// cosines of one unit from the granules' sums: slot i of a lane = group 2 i + half.  G = channel groups of the scale (1, 2, 4, 8)
__device__ __forceinline__ void cvm_cosines(float (&cacc)[4], float (&dot)[8], float (&na)[8], float (&nb)[8], float ea2, float eb2, int G) {
  if (G == 8) {  // granule = group
#pragma unroll
    for (int i = 0; i < 4; ++i) cacc[i] += cvm_cos_pair(dot[2 * i], dot[2 * i + 1], na[2 * i], na[2 * i + 1], nb[2 * i], nb[2 * i + 1], ea2, eb2);
    return;
  }
  asm volatile("" : "+v"(dot[0]), "+v"(na[0]), "+v"(nb[0]));  // (keeps the sums below out of the G = 8 path)
#pragma unroll
  for (int q = 0; q < 4; ++q) dot[q] = dot[2 * q] + dot[2 * q + 1], na[q] = na[2 * q] + na[2 * q + 1], nb[q] = nb[2 * q] + nb[2 * q + 1];
  if (G == 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i) cacc[i] += cvm_cos_pair(dot[2 * i], dot[2 * i + 1], na[2 * i], na[2 * i + 1], nb[2 * i], nb[2 * i + 1], ea2, eb2);
    return;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) dot[q] = dot[2 * q] + dot[2 * q + 1], na[q] = na[2 * q] + na[2 * q + 1], nb[q] = nb[2 * q] + nb[2 * q + 1];
  if (G == 2) {
    cacc[0] += cvm_cos_pair(dot[0], dot[1], na[0], na[1], nb[0], nb[1], ea2, eb2);
    return;
  }
  cacc[0] += cvm_cos_pair(dot[0] + dot[1], 0.0f, na[0] + na[1], 0.0f, nb[0] + nb[1], 0.0f, ea2, eb2);  // (upper half-wave: 0)
}


// The unit.  On entry (ha, la) / (hb, lb) hold the A operands of the two sides' first chunks, all four channel tiles; on exit
// (REQ) those of the NEXT unit's sides (`next_a`, `next_b`: complete per-lane addresses of their first chunks).  REQ = false is
// the copy for a depth index's last unit (peeled off the unit loop: a branch around the requests inside the unit cost 40 %).
// MORE: some side has further chunks (wave-uniform; compiled as a second copy of the unit so that the common copy stays
// straight-line code).
template <bool MORE, bool REQ>
__device__ __forceinline__ void cvm_unit_run(float (&cacc)[4], cvm_u4 (&ha)[4], cvm_u4 (&la)[4], cvm_u4 (&hb)[4], cvm_u4 (&lb)[4],
                                             const CvmSide& sa, const CvmSide& sb, const CvmTap& tap_a, const CvmTap& tap_b,
                                             const char* next_a, const char* next_b, float ea2, float eb2, int G,
                                             const cvm_u4* __restrict__ lut, int n, int half, int fh, int fw) {
  cvm_h8 wah, wal, wbh, wbl;
  cvm_weights(tap_a, sa.p, sa.xb, half, lut, wah, wal);
  cvm_weights(tap_b, sb.p, sb.xb, half, lut, wbh, wbl);
  float dot[8], na[8], nb[8];
  cvm_f16 fa0, fb0, fa1, fb1;
  // matrix instructions of tile ct (first chunks), then the next unit's operands of that tile
  auto tile = [&](int ct, cvm_f16& fa, cvm_f16& fb) {
    const cvm_h8 h_a = __builtin_bit_cast(cvm_h8, ha[ct]), l_a = __builtin_bit_cast(cvm_h8, la[ct]);
    const cvm_h8 h_b = __builtin_bit_cast(cvm_h8, hb[ct]), l_b = __builtin_bit_cast(cvm_h8, lb[ct]);
    cvm_f16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    fa = cvm_mfma(l_a, wah, z);
    fb = cvm_mfma(l_b, wbh, z);
    fa = cvm_mfma(h_a, wal, fa);
    fb = cvm_mfma(h_b, wbl, fb);
    fa = cvm_mfma(h_a, wah, fa);
    fb = cvm_mfma(h_b, wbh, fb);
  };
  auto request = [&](int ct) {
    if constexpr (REQ) {
      cvm_load_ct(ha[ct], la[ct], next_a, ct);
      cvm_load_ct(hb[ct], lb[ct], next_b, ct);
    }
  };
  // the dot products of a tile stay where they are written (next to the following tile's matrix instructions) instead of being
  // sunk to the cosines
#define CVM_PIN(q) asm volatile("" : "+v"(dot[q]), "+v"(dot[q + 1]), "+v"(na[q]), "+v"(na[q + 1]), "+v"(nb[q]), "+v"(nb[q + 1]));
  if constexpr (MORE) {
    // tile after tile in a rolled loop, every chunk's operands loaded on demand (the prefetched first chunks are not used: the
    // loop could only index them dynamically), then the next unit's operands
    const unsigned loff_a = cvm_lane_off(sa.rs, n, half), loff_b = cvm_lane_off(sb.rs, n, half);
#pragma unroll 1
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
      for (int i = 0; i < 16; ++i) fa0[i] = 0.0f, fb0[i] = 0.0f;
      cvm_all_chunks(fa0, sa, tap_a, loff_a, ct, lut, half, fh, fw);
      cvm_all_chunks(fb0, sb, tap_b, loff_b, ct, lut, half, fh, fw);
      float d2[2], a2[2], b2[2];
      cvm_dots(fa0, fb0, d2, a2, b2);
      switch (ct) {  // (wave-uniform; keeps the sums in registers)
        case 0: dot[0] = d2[0], dot[1] = d2[1], na[0] = a2[0], na[1] = a2[1], nb[0] = b2[0], nb[1] = b2[1]; break;
        case 1: dot[2] = d2[0], dot[3] = d2[1], na[2] = a2[0], na[3] = a2[1], nb[2] = b2[0], nb[3] = b2[1]; break;
        case 2: dot[4] = d2[0], dot[5] = d2[1], na[4] = a2[0], na[5] = a2[1], nb[4] = b2[0], nb[5] = b2[1]; break;
        default: dot[6] = d2[0], dot[7] = d2[1], na[6] = a2[0], na[7] = a2[1], nb[6] = b2[0], nb[7] = b2[1]; break;
      }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) request(ct);
  } else {
    tile(0, fa0, fb0);
    request(0);
    tile(1, fa1, fb1);
    cvm_dots(fa0, fb0, dot + 0, na + 0, nb + 0);
    CVM_PIN(0)
    request(1);
    tile(2, fa0, fb0);
    cvm_dots(fa1, fb1, dot + 2, na + 2, nb + 2);
    CVM_PIN(2)
    request(2);
    tile(3, fa1, fb1);
    cvm_dots(fa0, fb0, dot + 4, na + 4, nb + 4);
    CVM_PIN(4)
    request(3);
    cvm_dots(fa1, fb1, dot + 6, na + 6, nb + 6);
  }
#undef CVM_PIN
  cvm_cosines(cacc, dot, na, nb, ea2, eb2, G);
}


struct CvmGrid {
  int tile_y0, ntx, n_tiles, nsg, spw;  // first tile row, tiles per row, tiles, sample groups per tile, samples per wave and item
  int stage_rows;                       // rows assembled in LDS (0: many views - the scratch would cost the second workgroup of a CU)
  int pair_begin, pair_end, a0, b0;     // PAIR BLOCK of this launch (cv_walk.hpp "PAIR BLOCKS"): pairs [pair_begin, pair_end), the first one = views (a0, b0)
};

// per-wave LDS scratch of one depth index: projections (u, v) [V][32] | footprints [V * n_scales][32] x (uint4 weights, xy) |
// chunk ranges [V * n_scales]
// | the 32 rays' conditioning rows of the depth index [32][cond_stride] (assembled here, stored with 16-byte stores)
__host__ __device__ inline size_t cvm_wave_lds_bytes(int n_views, int n_scales, int cond_stride, int stage_rows) {
  const size_t items = (size_t)n_views * n_scales;
  return (size_t)n_views * 32 * 8 + items * (32 * 16 + 32 * 4 + 16) + (stage_rows ? (size_t)32 * cond_stride * 4 : 0);
}

#ifndef CVM_WAVES_PER_SIMD
#define CVM_WAVES_PER_SIMD 2
#endif
#ifndef CVM_WG_WAVES
#define CVM_WG_WAVES 4  // waves per workgroup: wave w runs on SIMD w % 4, so waves w and w + 4 share a SIMD
#endif
#ifndef CVM_EXP
#define CVM_EXP 0  // timing experiments (tools/exp): 1 no row write-out, 2 no colour taps, 3 no units, 4 no dot products, 5 no matrix instructions, 6 cached operands, 8 no direct row writes (many views)
#endif
#ifndef CVM_PRIO
#define CVM_PRIO 0
#endif
// -DCVM_STATS (tools/exp/cvmm_stats.py): per-phase s_memtime sums of every wave, added into the buffer MNERF_CVDBG_PTR names
#ifdef CVM_STATS
#define CVM_DBG_PARAM , unsigned long long* __restrict__ dbg
#define CVM_T(i)                                                  \
  {                                                               \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
    st[i] += t_ - t_last;                                         \
    t_last = t_;                                                  \
  }
#else
#define CVM_DBG_PARAM
#define CVM_T(i)
#endif
__global__ __launch_bounds__(64 * CVM_WG_WAVES, CVM_WAVES_PER_SIMD) void cost_volume_mm_kernel(mnerf_scene sc, mnerf_rays R, int cond_stride,
                                                                                 float* __restrict__ cond, const char* __restrict__ opnd,
                                                                                 CvmGrid grid CVM_DBG_PARAM) {
#ifdef CVM_STATS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) char cvm_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: the depth index and
  const int n = lane & 31, half = lane >> 5;                                                      //  what derives from it stay scalar)
  const int S = R.n_samples, V = sc.n_views, NS = sc.n_scales;
  const int W = R.width, H = R.height;
  const int items = V * NS;
  const CvmLayout L = cvm_layout(sc);
  const float* gain = reinterpret_cast<const float*>(opnd);
  const cvm_u4* lut = reinterpret_cast<const cvm_u4*>(cvm_smem);
  cvm_lut_init(reinterpret_cast<unsigned*>(cvm_smem));
  __syncthreads();
#if CVM_PRIO
  // The two waves of a SIMD run the same program and start together; with equal priority the arbiter alternates between them
  // and they stay in phase - both in their matrix runs, then both in their dot products - so neither pipe overlaps the other.
  // A fixed priority for the second team lets it run as if alone and the first team fill whichever pipe it leaves free.
  if (wave >= CVM_WG_WAVES / 2) __builtin_amdgcn_s_setprio(3);
#endif
  char* wl = cvm_smem + CVM_LUT_BYTES + (size_t)wave * cvm_wave_lds_bytes(V, NS, cond_stride, grid.stage_rows);
  float2* uv = reinterpret_cast<float2*>(wl);                                     // [V][32]
  cvm_u4* tapw = reinterpret_cast<cvm_u4*>(wl + (size_t)V * 256);                 // [items][32]
  unsigned* tapxy = reinterpret_cast<unsigned*>(wl + (size_t)V * 256 + (size_t)items * 512);  // [items][32]
  int4* boxes = reinterpret_cast<int4*>(wl + (size_t)V * 256 + (size_t)items * 640);          // [items]: p_lo, xb_lo, mask (0: big)
  float* rows = reinterpret_cast<float*>(wl + (size_t)V * 256 + (size_t)items * 656);        // [32][cond_stride]

  // XCD-major contiguous runs of items (see cost_volume_kernel): item = (tile, sample group), sample groups of a tile adjacent
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin;
  const int tile = item / grid.nsg, sg = item - tile * grid.nsg;
  const int tyi = tile / grid.ntx, txi = tile - tyi * grid.ntx;

  // this lane's ray: pixel (8 txi + n % 8, 4 (tile_y0 + tyi) + n / 8).  A pixel of the image that lies outside the launch's
  // range is still evaluated (and not stored): a tile's chunk ranges, and with them the order of its sums, then do not depend on
  // how a frame is cut into launches.  Lanes outside the image work on the nearest pixel inside.
  const int px = txi * 8 + (n & 7), py = (grid.tile_y0 + tyi) * 4 + (n >> 3);
  const int pix = min(py, H - 1) * W + min(px, W - 1);
  const int ray_geo = pix - R.ray_begin;                                           // make_ray: pixel = ray_begin + ray
  const int ray = max(0, min(ray_geo, R.n_rays - 1));                              // rows / stratified offsets: a ray of the launch
#if CVM_EXP == 8  // no direct row writes (many views)
  const bool row_wr = grid.stage_rows;
#else
  const bool row_wr = grid.stage_rows || (px < W && py < H && ray_geo >= 0 && ray_geo < R.n_rays);  // direct rows: live rays only
#endif
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  const int G0 = sc.n_group[0], G1 = NS > 1 ? sc.n_group[1] : 0;
  const int sumG = G0 + G1;
  const int n_pairs = V * (V - 1) / 2;
  const float inv_pairs = 1.0f / (float)n_pairs;
  const bool first_block = grid.pair_begin == 0, last_block = grid.pair_end == n_pairs;

  for (int k = 0; k < grid.spw; ++k) {
    const int j = (sg * grid.spw + k) * CVM_WG_WAVES + wave;
    if (j >= S) break;  // wave-uniform
    // this ray's row of the depth index: assembled in LDS, or (many views) written to memory value by value
    float* out = grid.stage_rows ? rows + n * cond_stride : cond + ((size_t)ray * S + j) * cond_stride;
    float wx, wy, wz;
    {
      int rg = ray_geo;
      asm volatile("" : "+v"(rg));  // the ray is rebuilt per depth index (15 instructions) instead of living in six registers
      ray_point(make_ray(R, rg), sample_depth(R, ray, j), wx, wy, wz);
    }

    CVM_T(7)
    // ---- pass 1 (cv_pass1's arithmetic): projections, colours, masks; half-wave `half` takes views half, half + 2, ..
    cvw_handoff();  // the previous depth index's reads of the scratch are done
    for (int v = half; v < V; v += 2) {
      float u, w_, z;
      project(sc.views[v], wx, wy, wz, wm1, hm1, u, w_, z);
      uv[v * 32 + n] = make_float2(u, w_);
      if (row_wr && first_block) {  // (a later pair block only needs the projections)
        const Bilin b = bilin_setup(u, w_, H, W);
        const float4* img = reinterpret_cast<const float4*>(sc.images) + (size_t)v * H * W;
#if CVM_EXP == 2
        const float4 t00 = make_float4(u, w_, z, 0.f), t01 = t00, t10 = t00, t11 = t00;
#else
        const float4 t00 = img[b.o00], t01 = img[b.o01], t10 = img[b.o10], t11 = img[b.o11];
#endif
        const float gx = u * 2.0f - 1.0f, gy = w_ * 2.0f - 1.0f;
        const float m = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;
        out[sumG + 3 * v + 0] = bilin4(t00.x, t01.x, t10.x, t11.x, b);
        out[sumG + 3 * v + 1] = bilin4(t00.y, t01.y, t10.y, t11.y, b);
        out[sumG + 3 * v + 2] = bilin4(t00.z, t01.z, t10.z, t11.z, b);
        out[sumG + 3 * V + v] = m;
      }
    }
    if (half == 0 && row_wr && first_block) {
      const int dc = sumG + 4 * V;
      out[dc] = 1.0f;  // constant input of the packed FiLM bias column
      for (int c = dc + 1; c < cond_stride; ++c) out[c] = 0.0f;
    }
    cvw_handoff();
    CVM_T(0)

    // ---- pass 1b: footprint and chunk range of every (view, scale) = item; half-wave `half` takes items half, half + 2, ..
    // (the range is a minimum / maximum over the 32 rays = the two DPP rows of the half-wave)
    for (int it = 0; 2 * it < items; ++it) {
      const int i = min(2 * it + half, items - 1);  // (an odd item count: the upper half-wave repeats the last item)
      const int v = NS == 2 ? i >> 1 : i, s = i - v * NS;  // (n_scales is 1 or 2)
      const float2 p2 = uv[v * 32 + n];
      int x0, y0;
      const int fh_s = s == 0 ? sc.fh[0] : sc.fh[1], fw_s = s == 0 ? sc.fw[0] : sc.fw[1];
      const CvmTap t = cvm_tap(p2.x, p2.y, fh_s, fw_s, x0, y0);
      cvm_u4 wv;
      wv.x = t.c0_hi, wv.y = t.c1_hi, wv.z = t.c0_lo, wv.w = t.c1_lo;
      tapw[i * 32 + n] = wv;
      tapxy[i * 32 + n] = t.xy;
      // chunk grid anchor and occupancy: minimum / OR over the 32 rays = the two DPP rows of the half-wave
      const int y1 = min(y0 + 1, fh_s - 1), x1 = min(x0 + 1, fw_s - 1);
      int xmin = x0, ymin = y0;
#define CVM_MM_STEP(CTRL)                                                                                                    \
  {                                                                                                                          \
    xmin = min(xmin, __builtin_amdgcn_update_dpp(0x7fffffff, xmin, CTRL, 0xF, 0xF, false)); /* old = the identity: one v_min_i32_dpp */ \
    ymin = min(ymin, __builtin_amdgcn_update_dpp(0x7fffffff, ymin, CTRL, 0xF, 0xF, false));                                         \
  }
      CVM_MM_STEP(0xB1)   // quad_perm [1,0,3,2]
      CVM_MM_STEP(0x4E)   // quad_perm [2,3,0,1]
      CVM_MM_STEP(0x141)  // row_half_mirror
      CVM_MM_STEP(0x140)  // row_mirror
#undef CVM_MM_STEP
      // the other row of this half-wave: lanes 16 apart (ds_swizzle BitMode: and 0x1f, or 0, xor 0x10)
      xmin = min(xmin, __builtin_amdgcn_ds_swizzle(xmin, 0x401F));
      ymin = min(ymin, __builtin_amdgcn_ds_swizzle(ymin, 0x401F));
      const int p_lo = ymin >> 1, xb_lo = xmin;  // first row pair, leftmost COLUMN
      // this ray's chunks on the grid anchored at (p_lo, xb_lo): rows cra, crb x columns cxa, cxb (bit 8 cr + cx; window 8 x 8)
      const int cra = ((y0 >> 1) - p_lo) >> 1, crb = ((y1 >> 1) - p_lo) >> 1;
      const int cxa = (x0 - xb_lo) >> 2, cxb = (x1 - xb_lo) >> 2;
      const unsigned long long ob = __builtin_amdgcn_ballot_w64(crb > 7 || cxb > 7);
      const bool big = (half ? (unsigned)(ob >> 32) : (unsigned)ob) != 0u;  // uniform over the half-wave
      unsigned long long mk = (1ull << ((8 * cra + cxa) & 63)) | (1ull << ((8 * cra + cxb) & 63)) | (1ull << ((8 * crb + cxa) & 63)) |
                              (1ull << ((8 * crb + cxb) & 63));
      unsigned mlo = (unsigned)mk, mhi = (unsigned)(mk >> 32);
#define CVM_OR_STEP(CTRL)                                                                             \
  {                                                                                                   \
    mlo |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mlo, CTRL, 0xF, 0xF, false);                 \
    mhi |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mhi, CTRL, 0xF, 0xF, false);                 \
  }
      CVM_OR_STEP(0xB1)
      CVM_OR_STEP(0x4E)
      CVM_OR_STEP(0x141)
      CVM_OR_STEP(0x140)
#undef CVM_OR_STEP
      mlo |= (unsigned)__builtin_amdgcn_ds_swizzle((int)mlo, 0x401F);
      mhi |= (unsigned)__builtin_amdgcn_ds_swizzle((int)mhi, 0x401F);
      if (n == 0 && 2 * it + half < items) boxes[i] = make_int4(p_lo, xb_lo, big ? 0 : (int)mlo, big ? 0 : (int)mhi);
    }
    cvw_handoff();

    CVM_T(1)
    // ---- pass 2: the units in the kernel's order (pair, scale), each one's operands requested during the unit before it
    float c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
    if (!first_block && row_wr) {  // a later pair block continues the raw sums the blocks before it left in the rows
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gi = 2 * i + half;
        if (gi < G0) c0[i] = out[gi];
        if (gi < G1) c1[i] = out[G0 + gi];
      }
    }
    auto tap_of = [&](int i) {
      CvmTap t;
      const cvm_u4 wv = tapw[i * 32 + n];
      t.xy = tapxy[i * 32 + n];
      t.c0_hi = wv.x, t.c1_hi = wv.y, t.c0_lo = wv.z, t.c1_lo = wv.w;
      return t;
    };
    auto side_of = [&](const char* map, int view, int s) {
      CvmSide o;
      o.map = map;
      o.rs = s == 0 ? L.rs[0] : L.rs[1];  // (no dynamic index: the layout would be read from scratch memory, and a scratch
                                          // load's wait also waits for every operand load in flight)
      o.item = view * NS + s;
      const int4 bx = boxes[o.item];
      o.box.p_lo = __builtin_amdgcn_readfirstlane(bx.x), o.box.xb_lo = __builtin_amdgcn_readfirstlane(bx.y);
      o.box.mask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(bx.z) |
                   ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(bx.w) << 32);
      o.box.big = o.box.mask == 0;  // (pass 1b stores no mask for a footprint wider than the window)
      o.p = o.box.p_lo, o.xb = o.box.xb_lo;  // general loop: the range's first chunk (occupied or not)
      if (!o.box.big) {
        unsigned long long m = o.box.mask;
        cvm_pop(m, o.box, o.p, o.xb);
      }
      return o;
    };
    auto first_of = [&](const CvmSide& sd) { return cvm_chunk_base(sd, sd.p, sd.xb) + cvm_lane_off(sd.rs, n, half); };
    auto has_more = [&](const CvmSide& sd) { return sd.box.big || (sd.box.mask & (sd.box.mask - 1)) != 0; };
    // running map pointers per scale: side a of the current pair (side b follows it in memory, the next pair's sides after that)
    const char* mp0 = opnd + L.off[0] + (size_t)(2 * grid.pair_begin) * L.map_bytes[0];
    const char* mp1 = opnd + L.off[1] + (size_t)(2 * grid.pair_begin) * L.map_bytes[1];
    int pr = grid.pair_begin, a = grid.a0, b = grid.b0, s = 0;
    CvmSide sa = side_of(mp0, a, 0), sb = side_of(mp0 + L.map_bytes[0], b, 0);
    cvm_u4 ha[4], la[4], hb[4], lb[4];
    {
      const char* fa_ = first_of(sa);
      const char* fb_ = first_of(sb);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) cvm_load_ct(ha[ct], la[ct], fa_, ct), cvm_load_ct(hb[ct], lb[ct], fb_, ct);
    }
#if CVM_EXP == 3
    const int n_units = 0;
#else
    const int n_units = (grid.pair_end - grid.pair_begin) * NS;
#endif
    // one unit; REQ: not the depth index's last one - the next unit's sides are looked up and their operands requested
    auto unit = [&](auto req_tag) {
      constexpr bool REQ = decltype(req_tag)::value;
      CVM_T(2)
      const float ga = gain[s * CVM_MAX_MAPS + 2 * pr], gb = gain[s * CVM_MAX_MAPS + 2 * pr + 1];
      const float ea = 1e-8f * ga, eb = 1e-8f * gb;  // clamps of the two norms in the scaled sums: (eps gain)^2
      const int G = s == 0 ? G0 : G1;
      const int s_cur = s;
      const int fh_u = s == 0 ? sc.fh[0] : sc.fh[1], fw_u = s == 0 ? sc.fw[0] : sc.fw[1];
      CvmSide na_ = sa, nb_ = sb;
      if constexpr (REQ) {  // the unit after this one: next scale of the pair, else the next pair
        if (++s == NS) {
          s = 0, ++pr;
          mp0 += 2 * L.map_bytes[0], mp1 += 2 * L.map_bytes[1];
          if (++b == V) ++a, b = a + 1;
        }
        const char* mpn = s == 0 ? mp0 : mp1;
        na_ = side_of(mpn, a, s);
        nb_ = side_of(mpn + (s == 0 ? L.map_bytes[0] : L.map_bytes[1]), b, s);
      }
      const CvmTap tap_a = tap_of(sa.item), tap_b = tap_of(sb.item);
      float cu[4] = {0.f, 0.f, 0.f, 0.f};
      if (has_more(sa) || has_more(sb))
        cvm_unit_run<true, REQ>(cu, ha, la, hb, lb, sa, sb, tap_a, tap_b, first_of(na_), first_of(nb_), ea * ea, eb * eb, G, lut, n, half, fh_u,
                                fw_u);
      else
        cvm_unit_run<false, REQ>(cu, ha, la, hb, lb, sa, sb, tap_a, tap_b, first_of(na_), first_of(nb_), ea * ea, eb * eb, G, lut, n, half, fh_u,
                                 fw_u);
      if (s_cur == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c0[i] += cu[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) c1[i] += cu[i];
      }
      sa = na_, sb = nb_;
      CVM_T(5)
    };
    for (int u = 0; u + 1 < n_units; ++u) unit(std::true_type{});
    if (n_units > 0) unit(std::false_type{});
    if (row_wr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gi = 2 * i + half;
        if (gi < G0) out[gi] = last_block ? c0[i] * inv_pairs : c0[i];
        if (gi < G1) out[G0 + gi] = last_block ? c1[i] * inv_pairs : c1[i];
      }
    }
    // the 32 rows leave as 16-byte pieces, consecutive lanes on consecutive pieces of a row (a row is cond_stride / 4 pieces:
    // the scattered 4-byte stores of one value per lane cost the memory pipeline ~20 instructions of 32 partial lines each)
    cvw_handoff();
    if (grid.stage_rows && CVM_EXP != 1) {
      const int ppr = cond_stride >> 2;  // pieces per row (cond_stride is a multiple of 8)
      const int ppr_sh = (ppr & (ppr - 1)) == 0 ? __builtin_ctz(ppr) : -1;  // a power of two: shifts instead of a division
      for (int c = lane; c < 32 * ppr; c += 64) {
        const int r = ppr_sh >= 0 ? c >> ppr_sh : c / ppr, part = c - r * ppr;
        const int rpx = txi * 8 + (r & 7), rpy = (grid.tile_y0 + tyi) * 4 + (r >> 3);
        const int rpix = rpy * W + rpx;
#if CVM_EXP == 7  // same bytes, one contiguous block per wave and depth index (is it the access pattern?)
        if (rpx < W && rpy < H && rpix >= R.ray_begin && rpix < R.ray_begin + R.n_rays)
          *reinterpret_cast<v4f*>(cond + (((size_t)tile * S + j) * 32) * cond_stride + 4 * c) =
              *reinterpret_cast<const v4f*>(rows + r * cond_stride + 4 * part);
#else
        if (rpx < W && rpy < H && rpix >= R.ray_begin && rpix < R.ray_begin + R.n_rays)
          *reinterpret_cast<v4f*>(cond + ((size_t)(rpix - R.ray_begin) * S + j) * cond_stride + 4 * part) =
              *reinterpret_cast<const v4f*>(rows + r * cond_stride + 4 * part);
#endif
      }
    }
    CVM_T(6)
  }
#ifdef CVM_STATS
  if (dbg && lane == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(dbg + i, st[i]);
    atomicAdd(dbg + 8, 1ull);
  }
#endif
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
  for (int s = 0; s < scene->n_scales; ++s) cells = cells > L.nrp[s] * L.rs[s] ? cells : L.nrp[s] * L.rs[s];
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
  g.nsg = (rays->n_samples + CVM_WG_WAVES * spw - 1) / (CVM_WG_WAVES * spw);
  const long long items = (long long)g.n_tiles * g.nsg;
  MNERF_REQUIRE(items < (1ll << 31), MNERF_E_RANGE, "mnerf_cost_volume: %lld work items", items);
  // Many views: one launch per BLOCK of view pairs over all rays (cv_walk.hpp "PAIR BLOCKS": the maps a launch gathers from
  // then fit the Infinity Cache and, per XCD band, come close to its L2); the raw cosine sums travel through the rows.
  const int n_pairs = scene->n_views * (scene->n_views - 1) / 2;
  // (the matrix form requests its operands a whole unit ahead and does not need the maps of a launch to fit a cache: all pairs in
  // one launch unless MNERF_CV_PAIR_BLOCK / the knob asks for blocks - 10 views: 67.8 ms in one launch, 77.8 in blocks of 8)
  int blk = mnerf_tune().cv_pair_block > 0 ? mnerf_tune().cv_pair_block : n_pairs;
  if (scene->n_views <= 5) blk = n_pairs;  // (what the walk does: up to 10 pairs x 2 sides x 13.1 MB at 512x640 = 262 MB)
  // rows are assembled in LDS unless that scratch is what keeps a second workgroup off the CU (160 KiB of LDS; from 7 views on)
  // - or the launch is one of several pair blocks (only the first writes colours and masks)
  g.stage_rows = blk >= n_pairs &&
                 (2 * (CVM_LUT_BYTES + CVM_WG_WAVES * cvm_wave_lds_bytes(scene->n_views, scene->n_scales, cond_stride, 1)) <= 160 * 1024 ||
                  2 * (CVM_LUT_BYTES + CVM_WG_WAVES * cvm_wave_lds_bytes(scene->n_views, scene->n_scales, cond_stride, 0)) > 160 * 1024);
  const size_t lds = CVM_LUT_BYTES + CVM_WG_WAVES * cvm_wave_lds_bytes(scene->n_views, scene->n_scales, cond_stride, g.stage_rows);
  MNERF_REQUIRE(lds <= 160 * 1024, MNERF_E_UNSUPPORTED, "mnerf_cost_volume: %d views need %zu B of LDS", scene->n_views, lds);
  static std::atomic<int> lds_set[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& seen = lds_set[dev & 63];
  if ((int)lds > seen.load(std::memory_order_relaxed)) {
    (void)hipFuncSetAttribute((const void*)cost_volume_mm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    seen.store((int)lds, std::memory_order_relaxed);
  }
#ifdef CVM_STATS
  unsigned long long* dbg = nullptr;
  if (const char* e = getenv("MNERF_CVDBG_PTR")) dbg = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  int a = 0, b = 1;
  for (int p0 = 0; p0 < n_pairs; p0 += blk) {
    g.pair_begin = p0, g.pair_end = p0 + blk < n_pairs ? p0 + blk : n_pairs, g.a0 = a, g.b0 = b;
#ifdef CVM_STATS
    hipLaunchKernelGGL(cost_volume_mm_kernel, dim3((unsigned)items), dim3(64 * CVM_WG_WAVES), lds, (hipStream_t)stream, *scene, *rays,
                       cond_stride, cond, reinterpret_cast<const char*>(scene->feat_op), g, dbg);
#else
    hipLaunchKernelGGL(cost_volume_mm_kernel, dim3((unsigned)items), dim3(64 * CVM_WG_WAVES), lds, (hipStream_t)stream, *scene, *rays,
                       cond_stride, cond, reinterpret_cast<const char*>(scene->feat_op), g);
#endif
    for (int i = 0; i < blk; ++i)  // the pair after the block's last one
      if (++b == scene->n_views) ++a, b = a + 1;
  }
  return mnerf_check_launch("mnerf_cost_volume");
}
