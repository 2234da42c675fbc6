// K3+K4+K5 — conditional radiance MLP + per-ray transformer + alpha compositing as ONE
// ray-chunk kernel for gfx950: fp32 data and accumulation; matrix products either on the exact-f32 MFMA
// (FMT = 0) or as fp32-grade products of three bf16 terms per operand on the bf16 MFMA (FMT = 1, default;
// see "split-bf16 matrix path" below).
//
// Replaces, per chunk of rays (paths relative to /root/reference):
//   models/matchnerf.py:118-132        NDC warp w.r.t. source view 0, view-dir rotation
//   models/rfdecoder/cond_nerf.py:52-100   CondNeRF.forward (posenc, FiLM-modulated MLP, heads)
//   models/rfdecoder/ray_transformer.py:14-26, 49-79   4-head attention along the ray + LN
//   models/rfdecoder/nerf.py:101-124   NeRF.composite
// The reference runs these as ~60 eager ops that materialise [R,S,128] activations per layer
// (134 MB each at R=4096,S=64) and a [R,4,S,S] score tensor (268 MB).  Here a workgroup owns
// a tile of TILE = 32*NW samples (whole rays), activations never leave registers, the
// ray-attention K/V and per-sample (rgb,sigma) live in LDS, and only 5 floats per ray are
// written to HBM.
//
// ---- MFMA formulation (the part that is specific to CDNA; described for FMT = 0) --------------
// Every Linear is evaluated TRANSPOSED:  Y^T[out, sample] = W[out, in] . H^T[in, sample]
// with v_mfma_f32_32x32x2_f32:  A = W tile (32 outs x 2 ins), B = H^T (2 ins x 32 samples).
// A wave owns 32 samples (N = lane&31) and all 128 outputs (4 M-blocks -> 4 x 16 accumulator
// VGPRs).  The C/D layout puts output row (r&3)+8*(r>>2)+4*(lane>>5) of block m in register
// r of lane (n, half) — which is exactly the B-operand layout the NEXT layer needs if its
// K-steps are taken in the order "register r of block m": lower half-wave supplies input
// feature f_lo(m,r), upper half supplies f_hi(m,r) = f_lo + 4.  Since a dot product does not
// care about the order of its terms, the host packs each weight matrix with its columns
// permuted to that order (matchnerf_amd/cond_nerf.py: pack_wstream, pack_wstream16), and the whole
// 6-layer MLP + heads chains accumulator -> operand with NO transpose, shuffle or LDS round trip.
// Biases ride along as one extra K-step whose B operand is the constant (1 | 0); the FiLM
// multiplier (pts_bias(cond), cond_nerf.py:62) is itself computed by an MFMA stage and kept
// in 64 VGPRs; the epilogue of a layer is one v_mul + v_max per accumulator register.
//
// Weights: 130k floats (521 KB as fp32 fragments, 808 KB as three bf16 terms) cannot live in LDS, so the packed A-fragment stream is cut
// into segments of <= 33 KiB that every wave consumes in the same order; segment i+1 is
// DMA'd global->LDS (global_load_lds_dwordx4, no VGPRs) into the other half of a double
// buffer while segment i feeds the MFMAs; one workgroup barrier per segment.
// With NW=4 a workgroup needs 76 KiB of LDS and <=256 VGPRs, so two workgroups share a CU
// (2 waves/SIMD) and de-synchronise: one's VALU phases (posenc, attention, compositing)
// overlap the other's MFMA phases.
#include <stdlib.h>

// This file is compiled TWICE (matchnerf_amd/csrc/build.py): MNERF_DECODER_PART 0 = the staged decoder kernels and their
// entry points (decoder.o), 1 = the one-launch form with the cost-volume walk inside (decoder_fused.o).  The second
// part is built without packed-fp32 vector instructions in the walk (cv_walk.hpp + -fno-slp-vectorize), because its
// workgroups share SIMDs with workgroups that issue 16-bit 32x32x16 matrix instructions: see "packed fp32 next to
// 16-bit MFMA" in DESIGN.md (section 4) — what round 2 recorded as an unexplained race of co-resident fused workgroups.
#ifndef MNERF_DECODER_PART
#define MNERF_DECODER_PART 0
#endif

#include "cv_walk.hpp"
#include "split_f16.hpp"


#define SEG_CAP_FLOATS (33 * 256)  // one LDS weight buffer: 33 KiB
#define MAX_SEGS 64
#define SMALL_FIXED 32    // floats of `small` (LayerNorm weight | bias) before the ray-posenc table
// tail segment (resident across the attention phase): float offsets of its sub-stages
#define TAIL_QKV 0
#define TAIL_FCO 1024
#define TAIL_OA0 1536
#define TAIL_OA2 2112
#define TAIL_FLOATS 2688

#ifdef MNERF_TIMELINE
// debug build (tools/exp/timeline.py): per-wave s_memtime stamps at phase boundaries
#define TL_POINTS 20
#define TL_STAMP(k)                                                                     \
  do {                                                                                  \
    if (sch.tl && lane == 0 && tl_slot >= 0 && tl_tile < 4)                             \
      sch.tl[(((size_t)tl_slot * 4 + tl_tile) * NW + wave) * TL_POINTS + (k)] =         \
          __builtin_amdgcn_s_memtime();                                                 \
  } while (0)
#else
#define TL_STAMP(k) do {} while (0)
#endif

struct DecSched {
#ifdef MNERF_TIMELINE
  unsigned long long* tl;
#endif
#ifdef MNERF_FUSED_DEBUG
  // debug build of the one-launch form (tools/exp/race_probe.py): what the trunk consumed, where each tile ran
  unsigned dbg_flags;   // 2 drain DMA + barrier before the walk, 4 first weight segment requested after the walk,
                        // 8 full wait + barrier after the FiLM inputs are read, 32 NaN-poison of the walk's LDS,
                        // 64 weight segments copied with plain loads + LDS stores instead of LDS-DMA
  float* dbg_rows;      // [rays*S][32] conditioning inputs as read by the trunk
  float* dbg_nv;        // [rays*S] mask sum as read by the trunk
  unsigned* dbg_tile;   // [tiles][4] blockIdx, HW_ID, XCC_ID, low clock word
#endif
  int stagger_sleeps;  // one-time start delay (x s_sleep 127) of the 2nd resident workgroup of a CU
  int stagger_mode;    // which workgroups wait: 0 odd HW wave slot, 1 upper half of grid, 2 (b>>3)&1, 3 all
  int n_seg;
  int film_steps, enc_steps;
  int seg_off[MAX_SEGS];     // float offset of the segment in wstream (multiple of 256)
  int seg_floats[MAX_SEGS];  // padded to a multiple of 256 floats (1 KiB DMA pieces)
  int seg_steps[MAX_SEGS];
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int NW>
__device__ __forceinline__ void prefetch_segment(const float* __restrict__ wstream,
                                                 const DecSched& sch, int seg, unsigned base /* LDS byte address */,
                                                 int wave, int lane) {
  if (seg >= sch.n_seg) return;
  const float* src = wstream + sch.seg_off[seg] + lane * 4;
  const int pieces = sch.seg_floats[seg] >> 8;
#ifdef MNERF_FUSED_DEBUG
  if (sch.dbg_flags & 64u) {  // no LDS-DMA at all: plain loads + LDS stores (compiler-tracked)
    typedef v4f32 __attribute__((address_space(3)))* lds_v4f32_ptr;
    for (int p = wave; p < pieces; p += NW) {
      const v4f32 t = *reinterpret_cast<const v4f32*>(src + p * 256);
      *((lds_v4f32_ptr)(size_t)(base + (unsigned)p * 1024u + (unsigned)lane * 16u)) = t;
    }
    return;
  }
#endif
  for (int p = wave; p < pieces; p += NW)
    glds16(src + p * 256, __builtin_amdgcn_readfirstlane(base + (unsigned)p * 1024u));
}

// one K-step against 4 / 2 / 1 M-blocks; A fragments laid out [step][lane][nmb]
__device__ __forceinline__ void step4(f32x16 (&acc)[4], const float* seg, int step, int lane,
                                      float b) {
  const float4 a = reinterpret_cast<const float4*>(seg)[step * 64 + lane];
  acc[0] = mfma(a.x, b, acc[0]);
  acc[1] = mfma(a.y, b, acc[1]);
  acc[2] = mfma(a.z, b, acc[2]);
  acc[3] = mfma(a.w, b, acc[3]);
}
__device__ __forceinline__ void step2(f32x16 (&acc)[2], const float* seg, int step, int lane,
                                      float b) {
  const float2 a = reinterpret_cast<const float2*>(seg)[step * 64 + lane];
  acc[0] = mfma(a.x, b, acc[0]);
  acc[1] = mfma(a.y, b, acc[1]);
}
__device__ __forceinline__ void step1(f32x16& acc, const float* seg, int step, int lane, float b) {
  acc = mfma(seg[step * 64 + lane], b, acc);
}

// 32 K-steps fed from two 16-register accumulator blocks of the previous layer
template <int NMB>
__device__ __forceinline__ void steps_from_regs(f32x16 (&acc)[NMB], const float* seg, int step0,
                                                int lane, const f32x16& h0, const f32x16& h1);
template <>
__device__ __forceinline__ void steps_from_regs<4>(f32x16 (&acc)[4], const float* seg, int step0,
                                                   int lane, const f32x16& h0, const f32x16& h1) {
  // A fragments are double-buffered in registers: the ds_read_b128 of step t+1 is issued before
  // the four MFMAs of step t, so its LDS latency hides under 256 cycles of matrix work.
  const float4* a4 = reinterpret_cast<const float4*>(seg) + step0 * 64 + lane;
  float4 cur = a4[0];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const float4 nxt = a4[(r + 1) * 64];  // r == 31 reads the following fragment (bias step or pad)
    __builtin_amdgcn_sched_barrier(0);    // keep the prefetch ABOVE this step's MFMAs
    const float b = r < 16 ? h0[r & 15] : h1[r & 15];
    acc[0] = mfma(cur.x, b, acc[0]);
    acc[1] = mfma(cur.y, b, acc[1]);
    acc[2] = mfma(cur.z, b, acc[2]);
    acc[3] = mfma(cur.w, b, acc[3]);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
}
// 16 pipelined K-steps against 4 M-blocks fed from one 16-register block
__device__ __forceinline__ void steps16_from_regs(f32x16 (&acc)[4], const float* seg, int step0, int lane,
                                                  const f32x16& e) {
  const float4* a4 = reinterpret_cast<const float4*>(seg) + step0 * 64 + lane;
  float4 cur = a4[0];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float4 nxt = a4[(r + 1) * 64];
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = mfma(cur.x, e[r], acc[0]);
    acc[1] = mfma(cur.y, e[r], acc[1]);
    acc[2] = mfma(cur.z, e[r], acc[2]);
    acc[3] = mfma(cur.w, e[r], acc[3]);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
}

template <>
__device__ __forceinline__ void steps_from_regs<2>(f32x16 (&acc)[2], const float* seg, int step0,
                                                   int lane, const f32x16& h0, const f32x16& h1) {
#pragma unroll
  for (int r = 0; r < 16; ++r) step2(acc, seg, step0 + r, lane, h0[r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) step2(acc, seg, step0 + 16 + r, lane, h1[r]);
}
template <>
__device__ __forceinline__ void steps_from_regs<1>(f32x16 (&acc)[1], const float* seg, int step0,
                                                   int lane, const f32x16& h0, const f32x16& h1) {
#pragma unroll
  for (int r = 0; r < 16; ++r) step1(acc[0], seg, step0 + r, lane, h0[r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) step1(acc[0], seg, step0 + 16 + r, lane, h1[r]);
}

// B operand of positional-encoding step t for this lane (cond_nerf.py:108-116 legacy /
// nerf.py:126-133 non-legacy; the packer maps weight columns accordingly):
//   t < 3L : arg = x_{t%3} * 2^{t/3} (* pi)  ->  lower half sin(arg), upper half cos(arg)
//   t = 3L : (x | y)      t = 3L+1 : (z | 1)   [the 1 multiplies the packed bias column]
__device__ __forceinline__ float enc_operand(int t, int L3, int hl, float x, float y, float z,
                                             float freq_mul) {
  if (t < L3) {
    const int l = t / 3, c = t - 3 * l;
    const float xc = (c == 0) ? x : ((c == 1) ? y : z);
    const float arg = xc * (ldexpf(1.0f, l) * freq_mul);
    return sin_quarter(arg, hl);
  }
  if (t == L3) return hl ? y : x;
  return hl ? 1.0f : z;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16 independent 4x4 outer products per wave: D[r](lane l) += A(lane 4*(l/4)+r) * B(lane l)
// (layout verified on MI355X, tools/exp/mfma4x4.hip)
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// 16 consecutive positional-encoding operands (steps t0 .. t0+15) evaluated into registers, so
// that the MFMA loop that consumes them is the same software-pipelined loop as a hidden layer.
// L_3D = 10 (every shipped config) is a compile-time case: with a run-time L every one of the 64 operand slots of a
// tile carries three comparisons against L whose results the compiler hoists out of the tile loop as exec-sized
// masks - ~300 spilled SGPRs, one v_readlane + v_cndmask per use (12 % of the kernel's VALU issue slots).
struct EncBase {   // x_c * freq_mul / (2 pi) as two floats per coordinate (exactly scalable by 2^l)
  float th[3], tl[3];
};
__device__ __forceinline__ EncBase enc_base(float x, float y, float z, float freq_mul) {
  EncBase b;
  turns_two_float(x * freq_mul, b.th[0], b.tl[0]);
  turns_two_float(y * freq_mul, b.th[1], b.tl[1]);
  turns_two_float(z * freq_mul, b.th[2], b.tl[2]);
  return b;
}
template <int T0>
__device__ __forceinline__ f32x16 enc_block16_L10(const EncBase& b, int hl, float x, float y, float z) {
  constexpr int L3 = 30;
  f32x16 e;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = T0 + i;
    if (t < L3) {
      const int l = t / 3, c = t - 3 * l;
      const float sc = (float)(1 << l);
      e[i] = sin_quarter_turns(b.th[c] * sc, b.tl[c] * sc, hl);
    } else if (t == L3) {
      e[i] = hl ? y : x;
    } else {
      e[i] = hl ? 1.0f : z;
    }
  }
  return e;
}

// ---------------------------------------------------------------- split-bf16 matrix path ("bf16x6")
// Same transposed chain on v_mfma_f32_32x32x16_bf16 (32 cycles per SIMD for 16 K-elements: 16x the
// K-rate of the f32 MFMA).  Each fp32 weight is stored as three bf16 terms (host,
// cond_nerf.py:pack_wstream16) and each fp32 activation is split the same way right before it is
// used; a product is accumulated in fp32 from six terms (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi,
// mid.mid) — what is dropped is < 2^-24 of the product, so results are fp32-grade (measured on
// MI355X, tools/exp/bf16x6.hip: max error below that of an fp32 FMA chain) at 16/6 of the f32 rate.
// Operand layout (verified by the same micro-test): lane (n, half) supplies k = 8*half + j, j < 8,
// for A row / B column n; C/D as for the f32 32x32 MFMA.  So a 16-register accumulator block of
// the previous layer is consumed as two K16-steps (registers 0-7, 8-15) with no data movement.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE)
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

struct Parts {
  bf16x8 hi, mid, lo;
};

__device__ __forceinline__ Parts split8(const float (&v)[8]) {
  u32x4 H, M, L;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned h = pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    const unsigned m = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    H[i] = h;
    M[i] = m;
    L[i] = pk_bf16(sa, sb);
  }
  Parts p;
  p.hi = __builtin_bit_cast(bf16x8, H);
  p.mid = __builtin_bit_cast(bf16x8, M);
  p.lo = __builtin_bit_cast(bf16x8, L);
  return p;
}

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// accumulators <- fp32 bias fragment [half][4][16] (exact fp32 biases, no K-step spent on them)
// LDS operands of the split-bf16 path are addressed by LDS byte offset through explicit address_space(3)
// pointers made from integers.

template <int NMB>
__device__ __forceinline__ void bias_init(f32x16 (&acc)[NMB], unsigned frag_lds, int hl) {
  lds_v4f32_cptr p = (lds_v4f32_cptr)(size_t)frag_lds + hl * 16;
#pragma unroll
  for (int m = 0; m < NMB; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f32 t = p[m * 4 + q];
      acc[m][4 * q] = t.x;
      acc[m][4 * q + 1] = t.y;
      acc[m][4 * q + 2] = t.z;
      acc[m][4 * q + 3] = t.w;
    }
}

// NS K16-steps against NMB output blocks.  `base`: fragments [step][block][hi|mid|lo][64 lanes][8 bf16];
// v: the lane's 8*NS operands.  A fragments of unit (step, block) i+1 are read before the six MFMAs
// of unit i (192 cycles of matrix work hide the LDS latency).
template <int NMB, int NS>
__device__ __forceinline__ void ksteps(f32x16 (&acc)[NMB], unsigned base_lds, int lane, const float (&v)[8 * NS]) {
  lds_u32x4_cptr a = (lds_u32x4_cptr)(size_t)base_lds + lane;
  u32x4 ch = a[0], cm = a[64], cl = a[128];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    float vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vv[j] = v[8 * u + j];
    const Parts b = split8(vv);
#pragma unroll
    for (int m = 0; m < NMB; ++m) {
      const int i = u * NMB + m;
      const int nx = (i + 1 < NS * NMB) ? (i + 1) * 192 : i * 192;  // the last unit re-reads itself
      const u32x4 nh = a[nx], nm = a[nx + 64], nl = a[nx + 128];
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 ah = __builtin_bit_cast(bf16x8, ch), am = __builtin_bit_cast(bf16x8, cm),
                   al = __builtin_bit_cast(bf16x8, cl);
      acc[m] = mfma16(ah, b.lo, acc[m]);
      acc[m] = mfma16(al, b.hi, acc[m]);
      acc[m] = mfma16(am, b.mid, acc[m]);
      acc[m] = mfma16(ah, b.mid, acc[m]);
      acc[m] = mfma16(am, b.hi, acc[m]);
      acc[m] = mfma16(ah, b.hi, acc[m]);
      __builtin_amdgcn_sched_barrier(0);
      ch = nh;
      cm = nm;
      cl = nl;
    }
  }
}

// two K16-steps fed from one 16-register block (an accumulator block of the previous layer)
template <int NMB>
__device__ __forceinline__ void kblock(f32x16 (&acc)[NMB], unsigned base_lds, int lane, const f32x16& h) {
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = h[r];
  ksteps<NMB, 2>(acc, base_lds, lane, v);
}
#define K16_UNIT_BYTES 3072  // one (step, block): hi | mid | lo fragments

// Experiment knob, OFF: streaming the conditioning rows with non-temporal stores (cost volume) and loads (here).
// Measured on MI355X (profiles/r2_nt_*): HBM-side counters unchanged (the scratch lines it was meant to protect are
// written back either way), cost volume 9.8 -> 11.2 ms, decoder unchanged.  The write-then-read pair of the staged
// form is served best by the default cache policy.
__device__ __forceinline__ float4 ld_stream4(const float* p, bool lds) {
  return *reinterpret_cast<const float4*>(p);
}

template <int NW, int SP>
struct Smem {
  static constexpr int TILE = NW * 32;
  static constexpr int W_FLOATS = 2 * SEG_CAP_FLOATS;
  static constexpr int RS_FLOATS = TILE * 4;
  static constexpr int LN_FLOATS = 64;
  // Ray-attention scratch lives in the weight buffer that does NOT hold the resident tail segment.
  //   MFMA form (SP <= 128): K [rays][4][SP][4], V^T [rays][4][4][SP], Q [TILE][16], O [TILE][16]
  //   VALU form (SP  = 256): K|V interleaved [rays][4][SP][8]
  static constexpr bool MFMA_ATT = SP <= 128;
  static constexpr int KV_FLOATS = MFMA_ATT ? TILE * 64 : TILE * 32;
  static_assert(KV_FLOATS <= SEG_CAP_FLOATS, "attention scratch must fit one weight buffer");
  static constexpr int TOTAL_FLOATS = W_FLOATS + RS_FLOATS + LN_FLOATS;
};

// CVF = 1: the FUSED ray-chunk form (K1..K5 in one launch): the workgroup first produces the conditioning rows of
// its own tile with the register-quad walk of cv_walk.hpp (8-sample walks, one per 16-lane slot) straight into LDS
// — no [rays*S, cond_stride] hand-off through HBM — and only then starts the MFMA trunk.  The walk is texture /
// VALU work with no matrix instruction; with two workgroups per CU one workgroup's walk runs under the other's MFMA
// stages.  LDS: walk scratch in weight buffer 1, the tile's rows in the part of weight buffer 0 above segment 0
// (the FiLM weights: 17 KiB at <= 32 conditioning inputs); both are dead before the weight pipeline needs them.
#define CVF_SEG 8
#define CVF_COND_OFF_FLOATS (17 * 256)
// Two workgroups of the fused form share a CU like those of the staged decoder (68.25 KiB of LDS each).  Round 2 saw a
// handful of wrong rays per frame in that configuration and reserved the CU (84 KiB) without finding the cause.  Round 3
// found it (tools/exp/race_probe.py): the conditioning rows were wrong, always in lanes 48-63 of a
// wave, one walk step (or one pass-1 view) at a time — packed-fp32 vector instructions (v_pk_fma_f32 / v_pk_mul_f32) of
// the walk lose their result in the last lane quarter while ANOTHER wave of the SIMD issues v_mfma_f32_32x32x16_{f16,
// bf16}; the stand-alone cost volume shows the same faults when it runs next to this decoder on a second stream, not next
// to the exact-f32 decoder, and none once it is built without packed-fp32 instructions.  This part of the file is
// therefore compiled with -fno-slp-vectorize (build.py; the walk's own arithmetic is unpacked in cv_walk.hpp), and the
// occupancy restriction is gone.
#define CVF_LDS_BYTES 0  // 0: the natural footprint (two workgroups per CU); bytes: reserve more (probes)
#ifndef MNERF_DECODER_MINBLOCKS
#define MNERF_DECODER_MINBLOCKS 2  // experiments: 1 = 512 registers per wave (one workgroup per CU), no spills
#endif
template <int NW, int SP, int FMT, int CVF>
__global__ __launch_bounds__(NW * 64, MNERF_DECODER_MINBLOCKS) void decoder_kernel(
    mnerf_decoder D, DecSched sch, mnerf_view view0, mnerf_rays R,
    const float* __restrict__ cond, float* __restrict__ out_rgb, float* __restrict__ out_depth,
    float* __restrict__ out_opacity, float* __restrict__ dbg_rgb_s, float* __restrict__ dbg_sigma,
    const float* __restrict__ ext_ndc, const float* __restrict__ ext_dir, mnerf_scene scene) {
  static_assert(!CVF || (FMT == 2 && NW == 4), "the fused form is built for the split-fp16 trunk");
  constexpr int Sp = SP;
  using SM = Smem<NW, SP>;
  constexpr int TILE = SM::TILE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wbuf0 = smem;
  float* wbuf1 = smem + SEG_CAP_FLOATS;
  // LDS byte addresses of the two weight buffers, for the DMA and for the split-bf16 operand reads.  (No
  // generic->LDS pointer casts anywhere near the hot loops: they trip a gfx950 code-generation bug in hipcc
  // 7.2 — "Illegal instruction detected: Operand has incorrect register class" on a V_CMP against
  // src_shared_base — depending on unrelated code around them.)
  const unsigned wbuf0_lds = __builtin_amdgcn_groupstaticsize();  // the dynamic array starts after the static LDS
  const unsigned wbuf1_lds = wbuf0_lds + SEG_CAP_FLOATS * 4u;
  float* rs_lds = smem + SM::W_FLOATS;                // [TILE][4]   rgb.xyz, sigma.w
  float* ln_lds = rs_lds + SM::RS_FLOATS;             // LayerNorm weight[16] | bias[16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int n = lane & 31, hl = lane >> 5;
  const int S = R.n_samples;
  const int rays_per_tile = TILE / Sp;
  const int n_tiles = (R.n_rays + rays_per_tile - 1) / rays_per_tile;
  const int L3 = 3 * D.L_3D;
  const int CS = D.cond_stride;
  const float freq_mul = R.legacy_coord ? 1.0f : 3.14159265358979323846f;
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);

  // Phase stagger.  With <=256 VGPRs and 76 KiB of LDS two workgroups share a CU (one wave of
  // each per SIMD).  Launched together they run IN PHASE: their VALU-only phases (prologue,
  // ray attention, compositing) coincide and the matrix pipe idles for both (measured: MFMA busy
  // 69.6 % = 2M/(2M+V)).  The grid is persistent (2 workgroups per CU), and the workgroup whose
  // waves sit in the odd hardware wave slot waits half a tile ONCE, so that from then on one
  // workgroup's VALU phases overlap the other's MFMA phases.  Speed only: any placement is correct.
  if (sch.stagger_sleeps > 0) {
    const unsigned hw_id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11));  // HW_ID.WAVE_ID
    bool late = (hw_id & 1u) != 0;
    if (sch.stagger_mode == 1) late = blockIdx.x >= (gridDim.x >> 1);
    if (sch.stagger_mode == 2) late = ((blockIdx.x >> 3) & 1) != 0;
    if (sch.stagger_mode == 3) late = true;
    if (late)
      for (int i = 0; i < sch.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  if (tid < SMALL_FIXED) ln_lds[tid] = D.small_[tid];
  __syncthreads();

  // Per-tile global inputs of a lane: its half-wave's FiLM operands (cond row) and the sum of the
  // visibility masks (cond_nerf.py:79-80).  (Fetching them one tile ahead was tried: the 33 extra
  // loop-carried VGPRs cost more in spills than the hidden latency gained.)
  float4 cpre[8];
  float n_valid = 0.0f;
  auto load_tile_inputs = [&](int t) {
    const int s_l = wave * 32 + n;
    const int r_t = s_l / Sp;
    const int j_p = s_l - r_t * Sp;
    int r = t * rays_per_tile + r_t;
    if (r >= R.n_rays) r = R.n_rays - 1;
    // row of this lane's sample: in the staged form a row of the [rays*S, CS] buffer in global memory, in the fused
    // form a row of the tile's [TILE, CS] block in LDS (written by this workgroup a moment ago)
    const float* crow_base = CVF ? (wbuf0 + CVF_COND_OFF_FLOATS) + (size_t)(r_t * Sp + (j_p < S ? j_p : (S - 1))) * CS
                                 : cond + ((size_t)r * S + (j_p < S ? j_p : (S - 1))) * CS;
    if constexpr (FMT >= 1) {  // K16 steps 0,1: cond[16 t + 8 hl + 4 q .. +4), q = i & 1, t = i >> 1
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = 16 * (i >> 1) + 8 * hl + 4 * (i & 1);
        cpre[i] = (o + 4 <= CS && (i >> 1) < sch.film_steps) ? ld_stream4(crow_base + o, CVF)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      const float4* crow4 = reinterpret_cast<const float4*>(crow_base + (size_t)hl * sch.film_steps);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        cpre[i] = (4 * i < sch.film_steps) ? crow4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* mrow = crow_base + (D.cond_dim - D.n_views);
    float nv = 0.0f;
    for (int v = 0; v < D.n_views; ++v) nv += mrow[v];
    n_valid = nv;
  };
  bool seg0_in_flight = false;

#ifdef MNERF_TIMELINE
  int tl_tile = -1;
  // record blocks [0,64) and their presumed CU partners [256,320)
  const int tl_slot = blockIdx.x < 64 ? (int)blockIdx.x : ((blockIdx.x >= 256 && blockIdx.x < 320) ? (int)blockIdx.x - 192 : -1);
#endif
  // Tile -> workgroup mapping.  Workgroup b runs on XCD b % 8 (observed dispatch order; speed only): XCD x takes the
  // contiguous tile range [x n/8, (x+1) n/8) and its workgroups step through it together, so the epipolar texels
  // the fused form gathers for concurrently processed tiles sit in that XCD's own L2.
  int tile_begin = blockIdx.x, tile_end = n_tiles, tile_step = gridDim.x;
  if (gridDim.x >= 8) {
    const int xcd = blockIdx.x & 7;
    tile_begin = (int)((long long)n_tiles * xcd / 8) + (int)(blockIdx.x >> 3);
    tile_end = (int)((long long)n_tiles * (xcd + 1) / 8);
    tile_step = ((int)gridDim.x - xcd + 7) >> 3;  // workgroups with this b % 8
  }
  for (int tile = tile_begin; tile < tile_end; tile += tile_step) {
#ifdef MNERF_TIMELINE
    ++tl_tile;
    if (sch.tl && lane == 0 && tl_slot >= 0 && tl_tile < 4)
      sch.tl[(((size_t)tl_slot * 4 + tl_tile) * NW + wave) * TL_POINTS + 19] =
          __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_ID
#endif
    TL_STAMP(0);
#ifdef MNERF_TIMELINE
    unsigned long long tl_dma_wait = 0, tl_bar_wait = 0;
#endif
    // ------------------------------------------------------------ per-lane sample identity
    const int s_local = wave * 32 + n;
    const int ray_t = s_local / Sp;                 // ray within the tile
    const int jp = s_local - ray_t * Sp;            // padded sample slot
    const int ray_raw = tile * rays_per_tile + ray_t;
    const bool ray_ok = ray_raw < R.n_rays;
    const int ray = ray_ok ? ray_raw : (R.n_rays - 1);
    const int j = jp < S ? jp : (S - 1);            // padded slots recompute the last sample
    const size_t gs = (size_t)ray * S + j;          // global sample index

    float x, y, z, dx, dy, dz;
    if (ext_ndc) {
      // mnerf_decoder_samples: the caller supplies the decoder inputs of CondNeRF.forward (cond_nerf.py:52) —
      // sample coordinates w.r.t. source view 0 and the (already rotated) unit view direction per sample
      x = ext_ndc[gs * 3 + 0];
      y = ext_ndc[gs * 3 + 1];
      z = ext_ndc[gs * 3 + 2];
      dx = ext_dir[gs * 3 + 0];
      dy = ext_dir[gs * 3 + 1];
      dz = ext_dir[gs * 3 + 2];
    } else {
      const RayGeom g = make_ray(R, ray);
      const float dpt = sample_depth(R, ray, j);
      float wx_, wy_, wz_;
      ray_point(g, dpt, wx_, wy_, wz_);
      project(view0, wx_, wy_, wz_, wm1, hm1, x, y, z);
      // view direction in the frame of source view 0 (matchnerf.py:129-131)
      const float rn = fmaxf(sqrtf(g.rx * g.rx + g.ry * g.ry + g.rz * g.rz), 1e-12f);
      const float ux = g.rx / rn, uy = g.ry / rn, uz = g.rz / rn;
      dx = ux * view0.extr[0] + uy * view0.extr[1] + uz * view0.extr[2];
      dy = ux * view0.extr[4] + uy * view0.extr[5] + uz * view0.extr[6];
      dz = ux * view0.extr[8] + uy * view0.extr[9] + uz * view0.extr[10];
    }

    const EncBase encb = enc_base(x, y, z, freq_mul);  // shared by the two positional-encoding stages (L0, L5)
    int seg = 0;   // running segment index; segment k lives in buffer (k & 1)
#ifdef MNERF_FUSED_DEBUG
    const unsigned dflags = CVF ? sch.dbg_flags : 0u;
    if (CVF && sch.dbg_tile && tid == 0) {
      sch.dbg_tile[tile * 4 + 0] = blockIdx.x;
      sch.dbg_tile[tile * 4 + 1] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_ID
      sch.dbg_tile[tile * 4 + 2] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // XCC_ID
      sch.dbg_tile[tile * 4 + 3] = (unsigned)__builtin_amdgcn_s_memtime();
    }
    if (!(dflags & 4u))
#endif
    if (!seg0_in_flight) prefetch_segment<NW>(D.wstream, sch, 0, wbuf0_lds, wave, lane);
#ifdef MNERF_FUSED_DEBUG
    if (dflags & 2u) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (dflags & 32u) {
      __syncthreads();
      for (int i = tid; i < SEG_CAP_FLOATS; i += NW * 64) wbuf1[i] = __builtin_nanf("");
      for (int i = CVF_COND_OFF_FLOATS + tid; i < SEG_CAP_FLOATS; i += NW * 64) wbuf0[i] = __builtin_nanf("");
      __syncthreads();
    }
#endif
    if constexpr (CVF) {
      // ---- K1+K2 for this tile: slot = 16 lanes, unit = CVF_SEG consecutive samples of one ray
      const int nv_ = scene.n_views;
      const int sumG_ = scene.n_group[0] + (scene.n_scales > 1 ? scene.n_group[1] : 0);
      const int per_slot = cv_slot_lds_floats(CVF_SEG, nv_, sumG_);
      const int slot = tid >> 4, sub = tid & 15;
      float* sl = wbuf1 + slot * per_slot;
      float* cond_lds = wbuf0 + CVF_COND_OFF_FLOATS;
      for (int unit = slot; unit < TILE / CVF_SEG; unit += NW * 4) {
        const int ls0 = unit * CVF_SEG;             // first local sample of the unit
        const int r_t = ls0 / Sp, jp0 = ls0 - r_t * Sp;
        const int rr = tile * rays_per_tile + r_t;
        const bool live = rr < R.n_rays;
        cv_walk_unit<8, CVF_SEG>(scene, R, live ? rr : R.n_rays - 1, live, jp0,
                                 cond_lds + (size_t)(r_t * Sp + (jp0 < S ? jp0 : S - 1)) * CS, CS, sl,
                                 reinterpret_cast<float4*>(sl + CVF_SEG * nv_ * 2), sl + CVF_SEG * (nv_ * 2 + 16), sub);
      }
      __syncthreads();  // the tile's rows are complete
    }
    load_tile_inputs(tile);  // issued before the geometry above is consumed: latency overlaps it
    const bool q_valid = n_valid > 1.0f;
#ifdef MNERF_FUSED_DEBUG
    if constexpr (CVF && FMT >= 1) {
      if (sch.dbg_rows && ray_ok && jp < S) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int o = 16 * (i >> 1) + 8 * hl + 4 * (i & 1);
          *reinterpret_cast<float4*>(sch.dbg_rows + gs * 32 + o) = cpre[i];
        }
        if (hl == 0) sch.dbg_nv[gs] = n_valid;
      }
      if (dflags & 8u) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (dflags & 4u) prefetch_segment<NW>(D.wstream, sch, 0, wbuf0_lds, wave, lane);
    }
#endif
    segment_wait();
    __syncthreads();  // publishes weight segment 0; in the fused form also: every lane holds its FiLM inputs, so
                      // weight buffer 1 (walk scratch) and the rows above segment 0 may be overwritten from here on

#define CUR_BUF ((seg & 1) ? wbuf1 : wbuf0)
#define NXT_BUF ((seg & 1) ? wbuf0 : wbuf1)
#define SEG_BEGIN() prefetch_segment<NW>(D.wstream, sch, seg + 1, (seg & 1) ? wbuf0_lds : wbuf1_lds, wave, lane)
#ifdef MNERF_TIMELINE
#define SEG_END()                                                  \
  do {                                                             \
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();   \
    segment_wait();                                                \
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime();   \
    __syncthreads();                                               \
    const unsigned long long t2_ = __builtin_amdgcn_s_memtime();   \
    tl_dma_wait += t1_ - t0_;                                      \
    tl_bar_wait += t2_ - t1_;                                      \
    ++seg;                                                         \
  } while (0)
#else
#define SEG_END()     \
  do {                \
    segment_wait();   \
    __syncthreads();  \
    ++seg;            \
  } while (0)
#endif

    TL_STAMP(1);
    float av[8];  // alpha-head activations: rows 0..15 <-> registers 0..7, feature (r&3) + 8*(r>>2) + 4*hl
    if constexpr (FMT == 2) {
      // ============================================================ trunk, split-fp16 matrix path
      // Scale bookkeeping: register values carry an integer exponent per lane (the same in the two lanes of a
      // sample): true value = register * 2^ec.  A stage picks the operand gain 2^em from the sample's largest
      // operand, the accumulator then holds 2^(ew + em - ec) (W h_true + b), i.e. its exponent is ec - em - ew.
      unsigned wb;
#define CUR_LDS ((seg & 1) ? wbuf1_lds : wbuf0_lds)
      // ------------------------------------------------------------ FiLM = pts_bias(cond); inputs in [-1, 1]
      f32x16 film[4];
      int ecf = 0;  // film_true = film * 2^ecf (never multiplied out: it rides in the exponent of each layer)
      // (defined on every path before the segment loop: a value that is only assigned under `done == 0` inside the loop
      // looks possibly-undefined to the register allocator, which then keeps its 64 registers reserved from the top of
      // the tile loop - across the whole fused cost-volume phase)
#pragma unroll
      for (int m = 0; m < 4; ++m) film[m] = (f32x16)(0.0f);
      {
        int done = 0;
        while (done < sch.film_steps) {
          const int ns = sch.seg_steps[seg];
          SEG_BEGIN();
          wb = CUR_LDS;
          if (done == 0) {
            const int ew = header_ew(CUR_LDS);
            ecf = -(ew + (H16_TARGET_EXP - 1));
            bias_init_h<4>(film, CUR_LDS, hl, pow2i(ew + (H16_TARGET_EXP - 1)));
            wb += 1024;
          }
          for (int u = 0; u < ns; ++u) {
            const int t = done + u;
            float v[8];
            float4 c0, c1;
            if (t == 0) {
              c0 = cpre[0];
              c1 = cpre[1];
            } else if (t == 1) {
              c0 = cpre[2];
              c1 = cpre[3];
            } else {  // more than 32 conditioning inputs (n_src_views > 5): straight from global
              const int o = 16 * t + 8 * hl;
              const float* crow = cond + gs * CS;
              c0 = (o + 4 <= CS) ? *reinterpret_cast<const float4*>(crow + o) : make_float4(0.f, 0.f, 0.f, 0.f);
              c1 = (o + 8 <= CS) ? *reinterpret_cast<const float4*>(crow + o + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w;
            v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
            ksteps_h<4, 1>(film, wb + u * 4 * H16_UNIT_BYTES, lane, v, (float)(1 << (H16_TARGET_EXP - 1)));
          }
          done += ns;
          SEG_END();
        }
      }
      TL_STAMP(2);
      // ------------------------------------------------------------ positional-encoding stages (L0, L5)
      f32x16 acc[4], h[4];
      const float enc_max = fmaxf(fmaxf(1.0f, fabsf(x)), fmaxf(fabsf(y), fabsf(z)));  // sin / cos <= 1; raw x, y, z
      int ew_cur = 0;
      // acc (+)= W_enc . enc(x) with operand gain 2^em; the first call of a stage loads bias * 2^(ew + em)
      auto enc_stage = [&](int em) {
        const float mult = pow2i(em);
        if (D.L_3D == 10) {  // one segment of four K16-steps, register-fed in two halves
          SEG_BEGIN();
          ew_cur = header_ew(CUR_LDS);
          bias_init_h<4>(acc, CUR_LDS, hl, pow2i(ew_cur + em));
          {
            const f32x16 e0 = enc_block16_L10<0>(encb, hl, x, y, z);
            kblock_h<4>(acc, CUR_LDS + 1024, lane, e0, mult);
          }
          {
            const f32x16 e1 = enc_block16_L10<16>(encb, hl, x, y, z);
            kblock_h<4>(acc, CUR_LDS + 1024 + 8 * H16_UNIT_BYTES, lane, e1, mult);
          }
          SEG_END();
        } else {
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
          int done = 0;
          while (done < sch.enc_steps) {
            const int ns = sch.seg_steps[seg];
            SEG_BEGIN();
            wb = CUR_LDS;
            if (done == 0) {
              ew_cur = header_ew(CUR_LDS);
              bias_init_h<4>(acc, CUR_LDS, hl, pow2i(ew_cur + em));
              wb += 1024;
            }
            for (int u = 0; u < ns; ++u) {
              float v[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = enc_operand(8 * (done + u) + j, L3, hl, x, y, z, freq_mul);
              ksteps_h<4, 1>(acc, wb + u * 4 * H16_UNIT_BYTES, lane, v, mult);
            }
            done += ns;
            SEG_END();
          }
        }
      };
      // acc (+)= W . h with register gain `mult`; with a header: acc <- bias * 2^bexp first (bexp - ew given)
      auto hidden_stage = [&](bool with_hdr, float mult, int bexp_minus_ew) {
#pragma unroll
        for (int sgi = 0; sgi < 2; ++sgi) {
          SEG_BEGIN();
          wb = CUR_LDS;
          if (sgi == 0 && with_hdr) {
            ew_cur = header_ew(CUR_LDS);
            bias_init_h<4>(acc, CUR_LDS, hl, pow2i(ew_cur + bexp_minus_ew));
            wb += 1024;
          }
          kblock_h<4>(acc, wb, lane, h[2 * sgi], mult);
          kblock_h<4>(acc, wb + 8 * H16_UNIT_BYTES, lane, h[2 * sgi + 1], mult);
          SEG_END();
        }
      };
      // h <- max(acc * film, 0); returns the sample's largest new activation (register units)
      auto film_relu = [&]() -> float {
        float mx = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float t = fmaxf(acc[m][r] * film[m][r], 0.0f);
            h[m][r] = t;
            mx = fmaxf(mx, t);
          }
        return fmaxf(mx, __shfl_xor(mx, 32, 64));
      };
      int ec;  // exponent of the values in h
      {
        const int em = gain_exp(enc_max);
        enc_stage(em);
        ec = -em - ew_cur + ecf;
      }
      float hmax = film_relu();
      TL_STAMP(3);
      // ------------------------------------------------------------ layers 1..4: 128 -> 128
      for (int layer = 1; layer <= 4; ++layer) {
        const int em = gain_exp(hmax);
        hidden_stage(true, pow2i(em), em - ec);
        ec = ec - em - ew_cur + ecf;
        hmax = film_relu();
      }
      TL_STAMP(4);
      // ------------------------------------------------------------ layer 5: [enc, h] -> 128, one accumulator:
      // both operand sets share one TRUE gain 2^eg, from the larger of the two maxima
      {
        const int eg = gain_exp(fmaxf(enc_max, hmax * pow2i(ec)));
        enc_stage(eg);
        hidden_stage(false, pow2i(eg + ec), 0);
        ec = -eg - ew_cur + ecf;
        hmax = film_relu();
      }
      // ------------------------------------------------------------ alpha head: 128 -> 16 (its activations wait in 8 registers for the ray transformer)
      const int em5 = gain_exp(hmax);
      const float mult5 = pow2i(em5);
      {
        f32x16 al[1];
        SEG_BEGIN();
        wb = CUR_LDS + 1024;
        const int ew = header_ew(CUR_LDS);
        bias_init_h<1>(al, CUR_LDS, hl, pow2i(ew + em5 - ec));
        const float ca = pow2i(ec - em5 - ew);
#pragma unroll
        for (int sgi = 0; sgi < 4; ++sgi) kblock_h<1>(al, wb + sgi * 2 * H16_UNIT_BYTES, lane, h[sgi], mult5);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float t = al[0][r] * ca;
          t = D.raytrans_elu ? (t > 0.0f ? t : (expf(t) - 1.0f)) : fmaxf(t, 0.0f);
          av[r] = t;
        }
        if (D.raytrans_posenc) {
          const float* tab = D.small_ + SMALL_FIXED + (size_t)j * 16;
#pragma unroll
          for (int r = 0; r < 8; ++r) av[r] += tab[(r & 3) + 8 * (r >> 2) + 4 * hl];
        }
        SEG_END();
      }
      TL_STAMP(5);
      // ------------------------------------------------------------ feature_linear: 128 -> 128 (no activation)
      hidden_stage(true, mult5, em5 - ec);
      const int ecfeat = ec - em5 - ew_cur;
      TL_STAMP(6);
      // ------------------------------------------------------------ views_linear: [feat, dir] -> 64
      f32x16 hv[2];
      int ecv;
      {
        const int eg = gain_exp(fmaxf(1.0f, sample_absmax<4>(acc) * pow2i(ecfeat)));  // |dir| <= 1
        const float multf = pow2i(eg + ecfeat), multd = pow2i(eg);
        SEG_BEGIN();
        wb = CUR_LDS + 1024;
        const int ew = header_ew(CUR_LDS);
        bias_init_h<2>(hv, CUR_LDS, hl, pow2i(ew + eg));
        kblock_h<2>(hv, wb, lane, acc[0], multf);
        kblock_h<2>(hv, wb + 4 * H16_UNIT_BYTES, lane, acc[1], multf);
        SEG_END();
        SEG_BEGIN();
        wb = CUR_LDS;
        kblock_h<2>(hv, wb, lane, acc[2], multf);
        kblock_h<2>(hv, wb + 4 * H16_UNIT_BYTES, lane, acc[3], multf);
        const float v[8] = {hl ? 0.0f : dx, hl ? 0.0f : dy, hl ? 0.0f : dz, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        ksteps_h<2, 1>(hv, wb + 8 * H16_UNIT_BYTES, lane, v, multd);
        SEG_END();
        ecv = -eg - ew;
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[m][r] = fmaxf(hv[m][r], 0.0f);
      TL_STAMP(7);
      // ------------------------------------------------------------ rgb_linear: 64 -> 3, sigmoid
      {
        const int em = gain_exp(sample_absmax<2>(hv));
        const float mult = pow2i(em);
        f32x16 c3[1];
        SEG_BEGIN();
        wb = CUR_LDS + 1024;
        const int ew = header_ew(CUR_LDS);
        bias_init_h<1>(c3, CUR_LDS, hl, pow2i(ew + em - ecv));
        const float cc = pow2i(ecv - em - ew);
        kblock_h<1>(c3, wb, lane, hv[0], mult);
        kblock_h<1>(c3, wb + 2 * H16_UNIT_BYTES, lane, hv[1], mult);
        if (hl == 0) {
          const float cr = 1.0f / (1.0f + expf(-c3[0][0] * cc));
          const float cg = 1.0f / (1.0f + expf(-c3[0][1] * cc));
          const float cb = 1.0f / (1.0f + expf(-c3[0][2] * cc));
          rs_lds[s_local * 4 + 0] = cr;
          rs_lds[s_local * 4 + 1] = cg;
          rs_lds[s_local * 4 + 2] = cb;
          if (dbg_rgb_s && ray_ok && jp < S) {
            dbg_rgb_s[gs * 3 + 0] = cr;
            dbg_rgb_s[gs * 3 + 1] = cg;
            dbg_rgb_s[gs * 3 + 2] = cb;
          }
        }
        SEG_END();
      }
      TL_STAMP(8);
#undef CUR_LDS
    } else if constexpr (FMT == 1) {
      // ============================================================ trunk, split-bf16 matrix path
      unsigned wb;  // LDS byte cursor inside the current weight segment
#define CUR_LDS ((seg & 1) ? wbuf1_lds : wbuf0_lds)
      // ------------------------------------------------------------ FiLM = pts_bias(cond)
      f32x16 film[4];
      {
        int done = 0;
        while (done < sch.film_steps) {
          const int ns = sch.seg_steps[seg];
          SEG_BEGIN();
          wb = CUR_LDS;
          if (done == 0) {
            bias_init<4>(film, CUR_LDS, hl);
            wb += 1024;
          }
          for (int u = 0; u < ns; ++u) {
            const int t = done + u;
            float v[8];
            float4 c0, c1;
            if (t == 0) {
              c0 = cpre[0];
              c1 = cpre[1];
            } else if (t == 1) {
              c0 = cpre[2];
              c1 = cpre[3];
            } else {  // more than 32 conditioning inputs (n_src_views > 5): straight from global
              const int o = 16 * t + 8 * hl;
              const float* crow = cond + gs * CS;
              c0 = (o + 4 <= CS) ? *reinterpret_cast<const float4*>(crow + o) : make_float4(0.f, 0.f, 0.f, 0.f);
              c1 = (o + 8 <= CS) ? *reinterpret_cast<const float4*>(crow + o + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w;
            v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
            ksteps<4, 1>(film, wb + u * 4 * K16_UNIT_BYTES, lane, v);
          }
          done += ns;
          SEG_END();
        }
      }
      TL_STAMP(2);
      // ------------------------------------------------------------ positional-encoding stages (L0, L5)
      f32x16 acc[4], h[4];
      auto enc_stage = [&]() {  // acc <- bias + W_enc . enc(x)
        if (D.L_3D == 10) {  // two segments of two K16-steps, register-fed
          {
            const f32x16 e0 = enc_block16_L10<0>(encb, hl, x, y, z);
            SEG_BEGIN();
            bias_init<4>(acc, CUR_LDS, hl);
            kblock<4>(acc, CUR_LDS + 1024, lane, e0);
            SEG_END();
          }
          {
            const f32x16 e1 = enc_block16_L10<16>(encb, hl, x, y, z);
            SEG_BEGIN();
            kblock<4>(acc, CUR_LDS, lane, e1);
            SEG_END();
          }
        } else {
          int done = 0;
          while (done < sch.enc_steps) {
            const int ns = sch.seg_steps[seg];
            SEG_BEGIN();
            wb = CUR_LDS;
            if (done == 0) {
              bias_init<4>(acc, CUR_LDS, hl);
              wb += 1024;
            }
            for (int u = 0; u < ns; ++u) {
              float v[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = enc_operand(8 * (done + u) + j, L3, hl, x, y, z, freq_mul);
              ksteps<4, 1>(acc, wb + u * 4 * K16_UNIT_BYTES, lane, v);
            }
            done += ns;
            SEG_END();
          }
        }
      };
      auto hidden_stage = [&](bool with_bias) {  // acc (+)= W . h : four segments, one per input block
#pragma unroll
        for (int sgi = 0; sgi < 4; ++sgi) {
          SEG_BEGIN();
          wb = CUR_LDS;
          if (sgi == 0 && with_bias) {
            bias_init<4>(acc, CUR_LDS, hl);
            wb += 1024;
          }
          kblock<4>(acc, wb, lane, h[sgi]);
          SEG_END();
        }
      };
      enc_stage();
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f);
      TL_STAMP(3);
      // ------------------------------------------------------------ layers 1..4: 128 -> 128
      for (int layer = 1; layer <= 4; ++layer) {
        hidden_stage(true);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f);
      }
      TL_STAMP(4);
      // ------------------------------------------------------------ layer 5: [enc, h] -> 128
      enc_stage();
      hidden_stage(false);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f);
      // ------------------------------------------------------------ alpha head: 128 -> 16 (its activations wait in 8 registers for the ray transformer)
      {
        f32x16 al[1];
        SEG_BEGIN();
        wb = CUR_LDS + 1024;
        bias_init<1>(al, CUR_LDS, hl);
#pragma unroll
        for (int sgi = 0; sgi < 4; ++sgi) kblock<1>(al, wb + sgi * 2 * K16_UNIT_BYTES, lane, h[sgi]);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float t = al[0][r];
          t = D.raytrans_elu ? (t > 0.0f ? t : (expf(t) - 1.0f)) : fmaxf(t, 0.0f);
          av[r] = t;
        }
        if (D.raytrans_posenc) {
          const float* tab = D.small_ + SMALL_FIXED + (size_t)j * 16;
#pragma unroll
          for (int r = 0; r < 8; ++r) av[r] += tab[(r & 3) + 8 * (r >> 2) + 4 * hl];
        }
        SEG_END();
      }
      TL_STAMP(5);
      // ------------------------------------------------------------ feature_linear: 128 -> 128
      hidden_stage(true);
      TL_STAMP(6);
      // ------------------------------------------------------------ views_linear: [feat, dir] -> 64
      f32x16 hv[2];
      {
        SEG_BEGIN();
        wb = CUR_LDS + 1024;
        bias_init<2>(hv, CUR_LDS, hl);
        kblock<2>(hv, wb, lane, acc[0]);
        kblock<2>(hv, wb + 4 * K16_UNIT_BYTES, lane, acc[1]);
        SEG_END();
        SEG_BEGIN();
        wb = CUR_LDS;
        kblock<2>(hv, wb, lane, acc[2]);
        kblock<2>(hv, wb + 4 * K16_UNIT_BYTES, lane, acc[3]);
        const float v[8] = {hl ? 0.0f : dx, hl ? 0.0f : dy, hl ? 0.0f : dz, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        ksteps<2, 1>(hv, wb + 8 * K16_UNIT_BYTES, lane, v);
        SEG_END();
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[m][r] = fmaxf(hv[m][r], 0.0f);
      TL_STAMP(7);
      // ------------------------------------------------------------ rgb_linear: 64 -> 3, sigmoid
      {
        f32x16 c3[1];
        SEG_BEGIN();
        wb = CUR_LDS + 1024;
        bias_init<1>(c3, CUR_LDS, hl);
        kblock<1>(c3, wb, lane, hv[0]);
        kblock<1>(c3, wb + 2 * K16_UNIT_BYTES, lane, hv[1]);
        if (hl == 0) {
          const float cr = 1.0f / (1.0f + expf(-c3[0][0]));
          const float cg = 1.0f / (1.0f + expf(-c3[0][1]));
          const float cb = 1.0f / (1.0f + expf(-c3[0][2]));
          rs_lds[s_local * 4 + 0] = cr;
          rs_lds[s_local * 4 + 1] = cg;
          rs_lds[s_local * 4 + 2] = cb;
          if (dbg_rgb_s && ray_ok && jp < S) {
            dbg_rgb_s[gs * 3 + 0] = cr;
            dbg_rgb_s[gs * 3 + 1] = cg;
            dbg_rgb_s[gs * 3 + 2] = cb;
          }
        }
        SEG_END();
      }
      TL_STAMP(8);
#undef CUR_LDS
    } else {
      // ============================================================ trunk, exact-f32 MFMA path
    // ------------------------------------------------------------ FiLM = pts_bias(cond)
    f32x16 film[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) film[m] = (f32x16)(0.0f);
    {
      // cond_stride <= 64 => film_steps <= 32 => one segment; inputs preloaded in cpre[]
      const int ns = sch.seg_steps[seg];
      SEG_BEGIN();
      const float* wseg = CUR_BUF;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (4 * i < ns) {
          step4(film, wseg, 4 * i + 0, lane, cpre[i].x);
          step4(film, wseg, 4 * i + 1, lane, cpre[i].y);
          step4(film, wseg, 4 * i + 2, lane, cpre[i].z);
          step4(film, wseg, 4 * i + 3, lane, cpre[i].w);
        }
      }
      SEG_END();
    }

    TL_STAMP(2);
    // ------------------------------------------------------------ layer 0: enc -> 128
    f32x16 acc[4], h[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
    if (D.L_3D == 10) {  // every shipped config: one segment, register-fed
      const f32x16 e0 = enc_block16_L10<0>(encb, hl, x, y, z);
      const f32x16 e1 = enc_block16_L10<16>(encb, hl, x, y, z);
      SEG_BEGIN();
      steps_from_regs<4>(acc, CUR_BUF, 0, lane, e0, e1);
      SEG_END();
    } else {
      int done = 0;
      while (done < sch.enc_steps) {
        const int ns = sch.seg_steps[seg];
        SEG_BEGIN();
        const float* wseg = CUR_BUF;
        for (int t = 0; t < ns; ++t)
          step4(acc, wseg, t, lane, enc_operand(done + t, L3, hl, x, y, z, freq_mul));
        done += ns;
        SEG_END();
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f);

    TL_STAMP(3);
    // ------------------------------------------------------------ layers 1..4: 128 -> 128
    for (int layer = 1; layer <= 4; ++layer) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
      SEG_BEGIN();
      steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[0], h[1]);
      SEG_END();
      SEG_BEGIN();
      steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[2], h[3]);
      step4(acc, CUR_BUF, 32, lane, hl ? 0.0f : 1.0f);  // bias column
      SEG_END();
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f);
    }

    TL_STAMP(4);
    // ------------------------------------------------------------ layer 5: [enc, h] -> 128
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
    if (D.L_3D == 10) {  // register-fed in two halves of 16 (film + h + acc are live here)
      SEG_BEGIN();
      {
        const f32x16 e = enc_block16_L10<0>(encb, hl, x, y, z);
        steps16_from_regs(acc, CUR_BUF, 0, lane, e);
      }
      {
        const f32x16 e = enc_block16_L10<16>(encb, hl, x, y, z);
        steps16_from_regs(acc, CUR_BUF, 16, lane, e);
      }
      SEG_END();
    } else {
      int done = 0;
      while (done < sch.enc_steps) {
        const int ns = sch.seg_steps[seg];
        SEG_BEGIN();
        const float* wseg = CUR_BUF;
        for (int t = 0; t < ns; ++t)
          step4(acc, wseg, t, lane, enc_operand(done + t, L3, hl, x, y, z, freq_mul));
        done += ns;
        SEG_END();
      }
    }
    SEG_BEGIN();
    steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[0], h[1]);
    SEG_END();
    SEG_BEGIN();
    steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[2], h[3]);
    SEG_END();
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f);

    TL_STAMP(5);
    // ------------------------------------------------------------ feature_linear: 128 -> 128
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
    SEG_BEGIN();
    steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[0], h[1]);
    SEG_END();
    SEG_BEGIN();
    steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[2], h[3]);
    step4(acc, CUR_BUF, 32, lane, hl ? 0.0f : 1.0f);
    SEG_END();

    TL_STAMP(6);
    // ------------------------------------------------------------ views_linear: [feat, dir] -> 64
    f32x16 hv[2];
    hv[0] = (f32x16)(0.0f);
    hv[1] = (f32x16)(0.0f);
    {
      SEG_BEGIN();
      const float* wseg = CUR_BUF;
      steps_from_regs<2>(hv, wseg, 0, lane, acc[0], acc[1]);
      steps_from_regs<2>(hv, wseg, 32, lane, acc[2], acc[3]);
      step2(hv, wseg, 64, lane, hl ? dy : dx);
      step2(hv, wseg, 65, lane, hl ? 1.0f : dz);
      SEG_END();
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) hv[m][r] = fmaxf(hv[m][r], 0.0f);

    TL_STAMP(7);
    // ------------------------------------------------------------ rgb_linear: 64 -> 3, sigmoid
    {
      f32x16 c3[1];
      c3[0] = (f32x16)(0.0f);
      SEG_BEGIN();
      const float* wseg = CUR_BUF;
      steps_from_regs<1>(c3, wseg, 0, lane, hv[0], hv[1]);
      step1(c3[0], wseg, 32, lane, hl ? 0.0f : 1.0f);
      if (hl == 0) {
        const float cr = 1.0f / (1.0f + expf(-c3[0][0]));
        const float cg = 1.0f / (1.0f + expf(-c3[0][1]));
        const float cb = 1.0f / (1.0f + expf(-c3[0][2]));
        rs_lds[s_local * 4 + 0] = cr;
        rs_lds[s_local * 4 + 1] = cg;
        rs_lds[s_local * 4 + 2] = cb;
        if (dbg_rgb_s && ray_ok && jp < S) {
          dbg_rgb_s[gs * 3 + 0] = cr;
          dbg_rgb_s[gs * 3 + 1] = cg;
          dbg_rgb_s[gs * 3 + 2] = cb;
        }
      }
      SEG_END();
    }

    TL_STAMP(8);
    // ------------------------------------------------------------ alpha head: 128 -> 16 (last trunk stage:
    // its activations stay in registers and feed the ray transformer's MFMA stages directly)
    // rows 0..15 <-> registers 0..7: feature o = (r&3) + 8*(r>>2) + 4*hl
    {
      f32x16 al[1];
      al[0] = (f32x16)(0.0f);
      SEG_BEGIN();  // DMA of the tail segment
      const float* wseg = CUR_BUF;
      steps_from_regs<1>(al, wseg, 0, lane, h[0], h[1]);
      steps_from_regs<1>(al, wseg, 32, lane, h[2], h[3]);
      step1(al[0], wseg, 64, lane, hl ? 0.0f : 1.0f);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float t = al[0][r];
        t = D.raytrans_elu ? (t > 0.0f ? t : (expf(t) - 1.0f)) : fmaxf(t, 0.0f);
        av[r] = t;
      }
      if (D.raytrans_posenc) {
        const float* tab = D.small_ + SMALL_FIXED + (size_t)j * 16;
#pragma unroll
        for (int r = 0; r < 8; ++r) av[r] += tab[(r & 3) + 8 * (r >> 2) + 4 * hl];
      }
      SEG_END();
    }

    }

    // ============================================================ ray transformer (K4)
    // The tail segment [w_qs;w_ks;w_vs | fc | out_alpha.0 | out_alpha.2] stays resident in its
    // weight buffer; the OTHER buffer (last read before the barrier above) is the K/V/Q/O scratch.
    TL_STAMP(17);
    const float* tail = CUR_BUF;
    float* att = NXT_BUF;
#undef CUR_BUF
#undef NXT_BUF
#undef SEG_BEGIN
#undef SEG_END
    float* kv_lds = att;                // VALU form: [rays][4 heads][Sp][8] (k0..3, v0..3)
    float* k_lds = att;                 // MFMA form
    float* vt_lds = att + TILE * 16;
    float* q_lds = att + TILE * 32;
    float* o_lds = att + TILE * 48;
    (void)kv_lds; (void)k_lds; (void)vt_lds; (void)q_lds; (void)o_lds;

    // ---- q|k|v = [Wq;Wk;Wv] a : 8 K-steps x 2 M-blocks, operands straight from the alpha registers.
    // Result rows: block 0 = q (regs 0..7) | k (regs 8..15), block 1 = v (regs 0..7); this lane
    // holds heads {hl, 2+hl} of its sample (register quad hh <-> head hl + 2*hh).
    f32x16 qkv[2];
    qkv[0] = (f32x16)(0.0f);
    qkv[1] = (f32x16)(0.0f);
#pragma unroll
    for (int r = 0; r < 8; ++r) step2(qkv, tail + TAIL_QKV, r, lane, av[r]);
    TL_STAMP(18);
    // temperature sqrt(d_k) = 2; masked query row -> uniform.  The MFMA form keeps its scores in the log2 domain
    // (log2 e folded into the query scale): the softmax numerator is then one v_exp_f32 per key
    const float qs = q_valid ? (SM::MFMA_ATT ? 0.5f * 1.4426950408889634f : 0.5f) : 0.0f;

    float ofc[8];  // attention output features [8*hl, 8*hl+8) of this lane's sample (head-major)
    if constexpr (SM::MFMA_ATT) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int head = hl + 2 * hh;
        *reinterpret_cast<float4*>(k_lds + ((ray_t * 4 + head) * Sp + jp) * 4) =
            make_float4(qkv[0][8 + 4 * hh], qkv[0][9 + 4 * hh], qkv[0][10 + 4 * hh], qkv[0][11 + 4 * hh]);
        float* vcol = vt_lds + (ray_t * 4 + head) * 4 * Sp + jp;
        vcol[0] = qkv[1][4 * hh];
        vcol[Sp] = qkv[1][4 * hh + 1];
        vcol[2 * Sp] = qkv[1][4 * hh + 2];
        vcol[3 * Sp] = qkv[1][4 * hh + 3];
        *reinterpret_cast<float4*>(q_lds + s_local * 16 + head * 4) =
            make_float4(qkv[0][4 * hh] * qs, qkv[0][4 * hh + 1] * qs, qkv[0][4 * hh + 2] * qs, qkv[0][4 * hh + 3] * qs);
      }
      __syncthreads();
      TL_STAMP(9);
      TL_STAMP(10);
      // ---- attention proper on the matrix pipe: lane = query.  v_mfma_f32_4x4x1_16b runs 16
      // independent 4x4 outer products per instruction, D[r](lane) += A(lane 4*(l/4)+r) B(lane):
      //   scores of 4 keys  s4[r] += K[k0+r][d] * Q[query][d]     (A = K row l%4, 4 steps over d)
      //   output            o4[d] += V[key][d] * P[query][key]    (A = V^T row l%4, 1 step per key)
      // Scores / probabilities of all S keys stay in VGPRs (the MLP's registers are dead here).
      int a_ray, a_hp, a_jq;
      if constexpr (SP >= 64) {
        constexpr int CH = SP / 64;
        int idx = wave;
        const int chunk = idx % CH;
        idx /= CH;
        a_hp = idx & 1;
        a_ray = idx >> 1;
        a_jq = chunk * 64 + lane;
      } else {
        a_ray = wave;  // SP == 32: one ray per wave, the two head pairs in the two half-waves
        a_hp = lane >> 5;
        a_jq = lane & 31;
      }
      const int s_q = a_ray * Sp + a_jq;
      // the scores of one head take SP registers per lane: at SP = 128 the two heads must not be unrolled into one
      // schedule (their score sets would be live together: 300 spilled VGPRs)
      constexpr int HEAD_UNROLL = SP >= 128 ? 1 : 2;
#pragma unroll HEAD_UNROLL
      for (int hh = 0; hh < 2; ++hh) {
        const int head = 2 * a_hp + hh;
        const float4 q4 = *reinterpret_cast<const float4*>(q_lds + s_q * 16 + head * 4);
        const float* kb = k_lds + ((a_ray * 4 + head) * Sp + (lane & 3)) * 4;
        f32x4 sc[SP / 4];
#pragma unroll
        for (int g = 0; g < SP / 4; ++g) {
          const float4 kk = *reinterpret_cast<const float4*>(kb + g * 16);
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
          t = mfma4(kk.x, q4.x, t);
          t = mfma4(kk.y, q4.y, t);
          t = mfma4(kk.z, q4.z, t);
          t = mfma4(kk.w, q4.w, t);
          sc[g] = t;
        }
        // four independent partial maxima / sums instead of one 64-long dependent chain
        float mx4[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        if (S == Sp) {  // no padded key slots (the usual case): no per-key masks (SP run-time comparisons otherwise)
#pragma unroll
          for (int g = 0; g < SP / 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx4[r] = fmaxf(mx4[r], sc[g][r]);
        } else {
          // (an opaque copy of S: otherwise the SP comparisons are hoisted out of the tile loop as SP lane masks in
          // 2 SP scalar registers - spilled, and paid for by the unpadded case too)
          int s_keys = S;
          asm volatile("" : "+s"(s_keys));
#pragma unroll
          for (int g = 0; g < SP / 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = (4 * g + r < s_keys) ? sc[g][r] : -3.0e38f;  // padded key slots
              sc[g][r] = v;
              mx4[r] = fmaxf(mx4[r], v);
            }
        }
        const float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
        float ls4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < SP / 4; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(sc[g][r] - mx);
            sc[g][r] = pr;
            ls4[r] += pr;
          }
        const float lsum = (ls4[0] + ls4[1]) + (ls4[2] + ls4[3]);
        const float* vb = vt_lds + ((a_ray * 4 + head) * 4 + (lane & 3)) * Sp;
        f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < SP / 4; g += 2) {
          const float4 va = *reinterpret_cast<const float4*>(vb + 4 * g);
          const float4 vc = *reinterpret_cast<const float4*>(vb + 4 * g + 4);
          oa = mfma4(va.x, sc[g][0], oa);
          ob = mfma4(vc.x, sc[g + 1][0], ob);
          oa = mfma4(va.y, sc[g][1], oa);
          ob = mfma4(vc.y, sc[g + 1][1], ob);
          oa = mfma4(va.z, sc[g][2], oa);
          ob = mfma4(vc.z, sc[g + 1][2], ob);
          oa = mfma4(va.w, sc[g][3], oa);
          ob = mfma4(vc.w, sc[g + 1][3], ob);
        }
        const float il = 1.0f / lsum;
        *reinterpret_cast<float4*>(o_lds + s_q * 16 + head * 4) =
            make_float4((oa[0] + ob[0]) * il, (oa[1] + ob[1]) * il, (oa[2] + ob[2]) * il, (oa[3] + ob[3]) * il);
      }
      __syncthreads();
      {
        const float4* src = reinterpret_cast<const float4*>(o_lds + s_local * 16 + 8 * hl);
        const float4 t0 = src[0], t1 = src[1];
        ofc[0] = t0.x; ofc[1] = t0.y; ofc[2] = t0.z; ofc[3] = t0.w;
        ofc[4] = t1.x; ofc[5] = t1.y; ofc[6] = t1.z; ofc[7] = t1.w;
      }
    } else {
      // ---- VALU form (S > 128): two lanes per sample, heads {hl, 2+hl}, K/V broadcast from LDS
      float ov[8];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float4* dst = reinterpret_cast<float4*>(kv_lds + ((size_t)(ray_t * 4 + hl + 2 * hh) * Sp + jp) * 8);
        dst[0] = make_float4(qkv[0][8 + 4 * hh], qkv[0][9 + 4 * hh], qkv[0][10 + 4 * hh], qkv[0][11 + 4 * hh]);
        dst[1] = make_float4(qkv[1][4 * hh], qkv[1][4 * hh + 1], qkv[1][4 * hh + 2], qkv[1][4 * hh + 3]);
      }
      __syncthreads();
      TL_STAMP(9);
      TL_STAMP(10);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float4* base = reinterpret_cast<const float4*>(kv_lds + (size_t)(ray_t * 4 + hl + 2 * hh) * Sp * 8);
        const float q0 = qkv[0][4 * hh] * qs, q1 = qkv[0][4 * hh + 1] * qs, q2 = qkv[0][4 * hh + 2] * qs,
                    q3 = qkv[0][4 * hh + 3] * qs;
        float mx = -3.0e38f;
        for (int jj = 0; jj < S; ++jj) {
          const float4 k4 = base[jj * 2];
          mx = fmaxf(mx, q0 * k4.x + q1 * k4.y + q2 * k4.z + q3 * k4.w);
        }
        float l = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        for (int jj = 0; jj < S; ++jj) {
          const float4 k4 = base[jj * 2], v4 = base[jj * 2 + 1];
          const float p = __expf((q0 * k4.x + q1 * k4.y + q2 * k4.z + q3 * k4.w) - mx);
          l += p;
          o0 += p * v4.x;
          o1 += p * v4.y;
          o2 += p * v4.z;
          o3 += p * v4.w;
        }
        const float il = 1.0f / l;
        ov[hh * 4] = o0 * il;
        ov[hh * 4 + 1] = o1 * il;
        ov[hh * 4 + 2] = o2 * il;
        ov[hh * 4 + 3] = o3 * il;
      }
      // this lane has heads {hl, 2+hl}; the fc stage wants features [8 hl, 8 hl + 8) = heads {2hl, 2hl+1}
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float mine_lo = ov[d], mine_hi = ov[4 + d];
        const float oth_lo = __shfl_xor(mine_lo, 32, 64), oth_hi = __shfl_xor(mine_hi, 32, 64);
        // hl = 0: heads 0 (mine_lo), 1 (partner's lo);  hl = 1: heads 2 (partner's hi), 3 (mine_hi)
        ofc[d] = hl ? oth_hi : mine_lo;
        ofc[4 + d] = hl ? mine_hi : oth_lo;
      }
    }

    TL_STAMP(11);
    // ---- fc (16x16, no bias) as one more MFMA stage + residual + LayerNorm(eps 1e-6); the 16
    // features of a sample are split over its two lanes exactly like the alpha registers.
    float yv[8];
    {
      f32x16 t1 = (f32x16)(0.0f);
#pragma unroll
      for (int t = 0; t < 8; ++t) step1(t1, tail + TAIL_FCO, t, lane, ofc[t]);
      float xs = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        yv[r] = t1[r] + av[r];
        xs += yv[r];
      }
      xs += __shfl_xor(xs, 32, 64);
      const float mean = xs * (1.0f / 16.0f);
      float var = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float dlt = yv[r] - mean;
        var += dlt * dlt;
      }
      var += __shfl_xor(var, 32, 64);
      const float rstd = 1.0f / sqrtf(var * (1.0f / 16.0f) + 1e-6f);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int o = (r & 3) + 8 * (r >> 2) + 4 * hl;
        yv[r] = (yv[r] - mean) * rstd * ln_lds[o] + ln_lds[16 + o];
      }
    }
    // ---- out_alpha_linear: 16 -> 16 (act) -> 1 (ReLU) as two MFMA stages (cond_nerf.py:33-36, 84)
    float sigma;
    {
      f32x16 t2 = (f32x16)(0.0f);
#pragma unroll
      for (int r = 0; r < 8; ++r) step1(t2, tail + TAIL_OA0, r, lane, yv[r]);
      step1(t2, tail + TAIL_OA0, 8, lane, hl ? 0.0f : 1.0f);
      f32x16 t3 = (f32x16)(0.0f);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float t = t2[r];
        t = D.raytrans_elu ? (t > 0.0f ? t : (expf(t) - 1.0f)) : fmaxf(t, 0.0f);
        step1(t3, tail + TAIL_OA2, r, lane, t);
      }
      step1(t3, tail + TAIL_OA2, 8, lane, hl ? 0.0f : 1.0f);
      sigma = fmaxf(t3[0], 0.0f);  // output row 0 <-> register 0 of the lower half-wave
    }
    if (D.density_maskfill && n_valid < 1.0f) sigma = 0.0f;
    if (hl == 0) {
      rs_lds[s_local * 4 + 3] = sigma;
      if (dbg_sigma && ray_ok && jp < S) dbg_sigma[gs] = sigma;
    }
    __syncthreads();
    // every weight / scratch read of this tile is complete: start the DMA of the next tile's first
    // weight segment now, under the compositing
    {
      seg0_in_flight = tile + tile_step < tile_end;
#ifdef MNERF_FUSED_DEBUG
      if (dflags & 4u) seg0_in_flight = false;
#endif
      if (seg0_in_flight) prefetch_segment<NW>(D.wstream, sch, 0, wbuf0_lds, wave, lane);
    }

    TL_STAMP(12);
    // ============================================================ compositing (K5)
    for (int rt = wave; rt < rays_per_tile; rt += NW) {
      const int rr = tile * rays_per_tile + rt;
      if (rr >= R.n_rays || !out_rgb) continue;  // wave-uniform (no compositing in the per-sample entry point)
      float rlen = 1.0f;
      if (!D.wo_render_interval) {
        const RayGeom gg = make_ray(R, rr);
        rlen = sqrtf(gg.rx * gg.rx + gg.ry * gg.ry + gg.rz * gg.rz);
      }
      float carry = 0.f, ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, ao = 0.f;
      for (int j0 = 0; j0 < S; j0 += 64) {
        const int jj = j0 + lane;
        const bool ok = jj < S;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        float dd = 0.f;
        if (ok) {
          c = reinterpret_cast<const float4*>(rs_lds)[rt * Sp + jj];
          dd = sample_depth(R, rr, jj);
          if (!D.wo_render_interval) {
            const float intv = (jj + 1 < S) ? (sample_depth(R, rr, jj + 1) - dd) : 1e10f;
            c.w = c.w * (intv * rlen);
          }
        }
        // exclusive prefix of sigma*delta: scan the lane-shifted values (see composite.hip)
        float incl = __shfl_up(c.w, 1, 64);
        if (lane == 0) incl = 0.0f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const float t = __shfl_up(incl, off, 64);
          if (lane >= off) incl += t;
        }
        const float excl = carry + incl;
        const float w = ok ? expf(-excl) * (1.0f - expf(-c.w)) : 0.0f;
        ar += w * c.x;
        ag += w * c.y;
        ab += w * c.z;
        ad += w * dd;
        ao += w;
        carry = __shfl(excl + c.w, 63, 64);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        ar += __shfl_xor(ar, off, 64);
        ag += __shfl_xor(ag, off, 64);
        ab += __shfl_xor(ab, off, 64);
        ad += __shfl_xor(ad, off, 64);
        ao += __shfl_xor(ao, off, 64);
      }
      if (lane == 0) {
        const float bg = D.setbg_opaque ? (1.0f - ao) : 0.0f;
        out_rgb[(size_t)rr * 3 + 0] = ar + bg;
        out_rgb[(size_t)rr * 3 + 1] = ag + bg;
        out_rgb[(size_t)rr * 3 + 2] = ab + bg;
        out_depth[rr] = ad;
        out_opacity[rr] = ao;
      }
    }
    TL_STAMP(13);
#ifdef MNERF_TIMELINE
    if (sch.tl && lane == 0 && tl_slot >= 0 && tl_tile < 4) {
      sch.tl[(((size_t)tl_slot * 4 + tl_tile) * NW + wave) * TL_POINTS + 15] = tl_dma_wait;
      sch.tl[(((size_t)tl_slot * 4 + tl_tile) * NW + wave) * TL_POINTS + 16] = tl_bar_wait;
    }
#endif
    // No barrier here: the next tile touches rs_lds only after several segment barriers,
    // and every read of the attention scratch (aliased on the weight buffers that the next
    // tile's first DMA overwrites) completed before the barrier in front of the compositing.
    TL_STAMP(14);
  }
}

#if MNERF_DECODER_PART == 0
// =====================================================================================================================
// Ping-pong form of the split-fp16 decoder (round 3): ONE 8-wave workgroup per CU, two TEAMS of four waves, each team
// owns 128 samples (whole rays) of a 256-sample tile.  Every SIMD holds one wave of each team, and the two teams run the
// same straight-line phase sequence ONE PHASE APART (team B enters the tile loop one barrier late):
//     phase 2s   = V_s: everything the vector ALU has to do before stage s (FiLM multiply + ReLU of the previous layer,
//                  per-sample gain, operand split into fp16 hi | lo fragments that stay in registers)
//     phase 2s+1 = M_s: stage s's matrix instructions, fed from those registers and from the weight fragments in LDS
// so while one wave of a SIMD issues back-to-back MFMAs its partner does VALU work, by construction instead of by the
// luck of two unsynchronised workgroups (the staged kernel above: matrix pipe 43 % busy, VALU 46 %).  One workgroup
// barrier per phase.  A weight segment is streamed ONCE per 256 samples (staged kernel: once per 128): stage s lives in
// ring slots 2 (s & 1) (+1), team A reads it in global step 2s+1, team B in step 2s+2; it is requested in two halves from VALU
// phases (team B the even 1-KiB pieces at the end of its V_{s-1}, step 2s-1: the slot pair's previous tenant, stage s-2, was
// last read in step 2s-2; team A the odd pieces at the start of its V_s, step 2s), each team waits for its own pieces in
// step 2s.  (Requests between the matrix instructions were measured, rounds 3 and 4: a request stalls its wave ~50 cycles
// there against ~95 in a vector phase, but the matrix phase is the other long side of a slot: no gain, MNERF_PP_DMA_IN_M.)  Resident for the whole kernel: the tail segment (ray-transformer weights)
// and the 1-KiB headers (biases, weight scales) of all stages, so that a V phase can also load the next accumulators.
// The conditioning rows of the next tile are copied to LDS during the tail phases (half of ring slot 1 per team).
// Same arithmetic in the same order as decoder_kernel<.., 2, 0>: results are bit-identical (tests).
// Scope: the shipped decoder shape (L_3D = 10, <= 32 conditioning inputs, S <= 256); anything else takes the kernel above.
// S in (128, 256] (round 5, `WIDE`): ONE ray per 256-sample tile, team A its samples 0..127, team B 128..255.  The trunk is per
// sample; the tail meets across the teams: each team's ray attention reads its own 128 keys at once and the other team's when
// that team's waves have all arrived at the sync behind their q|k|v phase (ray_attention_pp256: two key halves merged flash
// style), and the whole ray is composited by team B's first wave once team A's densities are there.  No extra LDS: a team still
// keeps only its own keys / values (in the 128-sample layout) and reads the other team's scratch in place; "arrived" is read off
// the other team's monotonic sync counter (3 team syncs per tile for A, 4 for B), so the teams keep running their tails freely.
#define PP_STAGES 12
#define PP_PHASES 28
struct PPSched {
  int seg_first[PP_STAGES];  // first weight segment of the stage (index into DecSched)
  int n_seg[PP_STAGES];      // 1 or 2
};

template <int SP>
struct SmemPP {
  static constexpr int TEAM = 128, TILE = 256;
  static constexpr int RING_FLOATS = 4 * SEG_CAP_FLOATS;
  static constexpr int TAIL_F = ((TAIL_FLOATS + 255) / 256) * 256;
  static constexpr int HDR_FLOATS = PP_STAGES * 256;     // the 1-KiB headers (biases, weight scale) of all stages, resident
  static constexpr int RS_FLOATS = TILE * 4;
  static constexpr int TOTAL_FLOATS = RING_FLOATS + TAIL_F + HDR_FLOATS + RS_FLOATS + 64;
  static_assert(TEAM * 64 <= SEG_CAP_FLOATS, "a team's attention scratch must fit one ring slot");
  static_assert(TOTAL_FLOATS * 4 <= 160 * 1024, "LDS budget of one CU");
};
// one-KiB pieces of the two weight segments of every stage (the shipped decoder shape; checked against the schedule on the host)
// MNERF_PP_L5_H_FIRST (default 1): layer 5 = [enc, h] -> 128 takes its ACTIVATION half first (stage 6 = stream segments 11, 12)
// and its encoding half second (stage 7 = segment 10, whose header carries the layer's biases and scale).  In stream order
// (0) phase V_6 holds the FiLM multiplier, layer 4's activations AND the new accumulators (3 x 64 registers) while it
// re-evaluates the encoding: the longest vector phase of the tile (6.6 k cycles against 3.9 k) and the origin of most
// spills.  Activation half first, the activations are consumed where they are produced and the encoding is evaluated in
// the short phase V_7.  The layer's sum is then accumulated in the other order, so this kernel's results differ from
// decoder_kernel's in the last bits (not from one launch to the next: tests/test_stress_gpu.py).
__device__ __host__ constexpr int pp_p0(int s, int fs = 2) {
  if (s == 0) return 1 + 8 * fs;  // FiLM: header + fs K16-steps x 4 blocks x (hi | lo)
  constexpr int t[PP_STAGES] = {17, 33, 33, 33, 33, 33, 32, 33, 17, 33, 17, 9};
  return t[s];
}
// The ping-pong kernel's schedule is ONE fixed schedule (the host refuses anything else), so the position of every segment in
// the weight stream is a compile-time constant too: stream order = FiLM | L0 | L1 a b | L2 a b | L3 a b | L4 a b | L5-enc |
// L5-h a b | alpha | feature a b | views a b | rgb | tail.  Taken from the kernel arguments (sch.seg_off[pps.seg_first[s] + i]:
// two dependent scalar loads per use) the 20 addresses were hoisted out of the tile loop and parked in VGPR lanes - 280
// spilled SGPRs, 850 v_readlane per tile (12 % of the kernel's vector instructions).
__device__ __host__ constexpr int pp_stream_pieces(int k, int fs = 2) {  // 1-KiB pieces of stream segment k
  if (k == 0) return 1 + 8 * fs;
  constexpr int t[20] = {17, 33, 33, 32, 33, 32, 33, 32, 33, 32, 33, 32, 32, 17, 33, 32, 17, 20, 9, 11};
  return t[k];
}
__device__ __host__ constexpr int pp_stream_off_floats(int k, int fs = 2) {
  int o = 0;
  for (int m = 0; m < k; ++m) o += pp_stream_pieces(m, fs) * 256;
  return o;
}
__device__ __host__ constexpr int pp_seg_first(int s) {
  constexpr int t[PP_STAGES] = {0, 1, 2, 4, 6, 8, 11, 10, 13, 14, 16, 18};
  return t[s];
}
__device__ __host__ constexpr int pp_seg_off_floats(int s, int i, int fs = 2) { return pp_stream_off_floats(pp_seg_first(s) + i, fs); }
// how many of a wave's requests for stage n are issued inside the matrix phase M_{n-1} (decoder_pp_kernel: pp_m_dma).  Stages
// whose previous matrix phase is not a plain ksteps_presplit2 call (M_0: FiLM + geometry, M_1 / M_7: the encoding halves with
// the operand split between them) or has the next tile's first stage in flight (M_11) keep every request in the vector phase.
// PP_DMA_IN_M = 0: no request is issued inside a matrix phase (4 measured 16.45-16.53, 2: 16.54-16.6 against 16.35-16.43 ms per
// frame with all of them in the vector phases: profiles/history/r4_variants.md); the mechanism stays for the next attempt
constexpr int PP_DMA_IN_M = 0;
#ifndef MNERF_PP_DMA_RUNS
#define MNERF_PP_DMA_RUNS 1  // the bursts' requests in runs of four consecutive pieces per M0 write (stage_dma); 0: one by one
#endif
static_assert(MNERF_PP_DMA_RUNS == 0 || PP_DMA_IN_M == 0, "requests inside matrix phases take single pieces 2 tw + half + 8 k");
__device__ __host__ constexpr int pp_km(int n) {
  const int enc_stage = 7;  // (layer 5 takes its activation half first) the stage AFTER the encoding half of layer 5 stays in the burst
  if (n < 3 || n > 11 || n == enc_stage + 1) return 0;
  // requests k = 0 .. km-1 of a wave are pieces 2 tw + half + 8 k of the stage's first segment: they must exist for tw = 3, half = 1
  const int fit = pp_p0(n) / 8;  // largest km with 7 + 8 (km - 1) < pp_p0
  return fit < PP_DMA_IN_M ? fit : PP_DMA_IN_M;
}
__device__ __host__ constexpr int pp_p1(int s) {
  constexpr int t[PP_STAGES] = {0, 0, 32, 32, 32, 32, 32, 0, 0, 32, 20, 0};
  return t[s];
}


// ---- wave-level reductions / scan of the compositing phase WITHOUT LDS round trips (MNERF_PP_T4_DPP).  __shfl_xor / __shfl_up
// are ds_bpermute_b32: every one of the ~40 exchanges of a ray's compositing (exclusive transmittance scan + five sums) is an
// LDS access of ~100 cycles that the next one depends on.  DPP row operations exchange inside a row of 16 lanes in the vector
// ALU itself; rows are joined by row_bcast (scan) or four v_readlane (sums).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_get0(float v) {  // the source lane's value, 0 where the row is masked / the source is outside
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float readlane_f(float v, int l) {  // (the builtin is an INTEGER operation: a float argument would be
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));  //  converted by value - 0.37f -> 0 - not by bits)
}
__device__ __forceinline__ float wave_sum_dpp(float v) {  // total over the 64 lanes (wave-uniform)
  v = dpp_group_sum<16>(v);  // every lane: the sum of its row
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
// lane l: v[0] + .. + v[l-1].  Row operations (row_shr) and v_readlane only (the wave-wide DPP forms wave_shr / row_bcast of
// earlier GCN parts assemble for gfx950; not relied upon here).
__device__ __forceinline__ float wave_exclusive_sum_dpp(float v, int lane) {
  float x = dpp_get0<0x111>(v);           // row_shr:1: v[l-1] inside a row, 0 at a row's first lane ..
  const float c15 = readlane_f(v, 15), c31 = readlane_f(v, 31), c47 = readlane_f(v, 47);
  x = lane == 16 ? c15 : (lane == 32 ? c31 : (lane == 48 ? c47 : x));  // .. which takes the last lane of the row before
  x += dpp_get0<0x111>(x);                // row_shr:1
  x += dpp_get0<0x112>(x);                // row_shr:2
  x += dpp_get0<0x114>(x);                // row_shr:4
  x += dpp_get0<0x118>(x);                // row_shr:8     -> inclusive scan inside every row
  const float t0 = readlane_f(x, 15), t1 = readlane_f(x, 31), t2 = readlane_f(x, 47);
  const float t01 = t0 + t1;
  return x + (lane >= 48 ? t01 + t2 : (lane >= 32 ? t01 : (lane >= 16 ? t0 : 0.0f)));
}

// Ray attention of the ping-pong kernel (phase T2): the arithmetic of decoder_kernel's MFMA form, every accumulator fed in the same
// order (bit-identical), but written so that no matrix instruction waits for its predecessor: the chains of TWO key groups (and,
// for S <= 64, of the wave's TWO heads) are interleaved — v_mfma_f32_4x4x1 is a 2-pass instruction, and as 4-long dependent
// chains issued one group after the other (round 3) every instruction paid the full pipeline latency plus a register copy —,
// and the softmax numerators of a group pair (v_exp_f32: quarter rate, the bound of this phase) are evaluated right in front of
// that pair's output products, which then run under the next pair's exponentials.
// PADDED = false: S == SP, no masked key slots.  A template parameter, not a run-time test inside: with `if (S == SP)` around the
// maxima the compiler if-converted the two cases into ONE stream that evaluated the padded-key selects always — 128 v_cndmask +
// 128 v_mov + 256 scalar compare / select instructions per wave next to 128 exponentials, in the unpadded case too.
template <int SP, bool PADDED>
__device__ __forceinline__ void ray_attention_pp(const float* q_lds, const float* k_lds, const float* vt_lds, float* o_lds,
                                                 int a_ray, int a_hp, int s_q, int lane, int S) {
  constexpr int G = SP / 4;
  constexpr int NH = SP >= 128 ? 1 : 2;  // heads side by side (their score sets are live together: SP registers each)
#pragma unroll 1
  for (int h0 = 0; h0 < 2; h0 += NH) {
    f32x4 sc[NH][G];
    float4 q4[NH];
    const float* kb[NH];
    const float* vb[NH];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
      const int head = 2 * a_hp + h0 + hh;
      q4[hh] = *reinterpret_cast<const float4*>(q_lds + s_q * 16 + head * 4);
      kb[hh] = k_lds + ((a_ray * 4 + head) * SP + (lane & 3)) * 4;
      vb[hh] = vt_lds + ((a_ray * 4 + head) * 4 + (lane & 3)) * SP;
    }
    // ---- scores of 4 keys per group: s4[r] += K[4 g + r][d] * Q[query][d]
#pragma unroll
    for (int g = 0; g < G; g += 2) {
      float4 ka[NH], kc[NH];
      f32x4 ta[NH], tc[NH];
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        ka[hh] = *reinterpret_cast<const float4*>(kb[hh] + g * 16);
        kc[hh] = *reinterpret_cast<const float4*>(kb[hh] + g * 16 + 16);
      }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        ta[hh] = mfma4(ka[hh].x, q4[hh].x, (f32x4){0.f, 0.f, 0.f, 0.f});
        tc[hh] = mfma4(kc[hh].x, q4[hh].x, (f32x4){0.f, 0.f, 0.f, 0.f});
      }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        ta[hh] = mfma4(ka[hh].y, q4[hh].y, ta[hh]);
        tc[hh] = mfma4(kc[hh].y, q4[hh].y, tc[hh]);
      }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        ta[hh] = mfma4(ka[hh].z, q4[hh].z, ta[hh]);
        tc[hh] = mfma4(kc[hh].z, q4[hh].z, tc[hh]);
      }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        sc[hh][g] = mfma4(ka[hh].w, q4[hh].w, ta[hh]);
        sc[hh][g + 1] = mfma4(kc[hh].w, q4[hh].w, tc[hh]);
      }
    }
    // ---- row maxima (four partial maxima per head, as in decoder_kernel)
    float mx[NH];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
      float mx4[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
      if constexpr (!PADDED) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx4[r] = fmaxf(mx4[r], sc[hh][g][r]);
      } else {
        int s_keys = S;  // opaque: see decoder_kernel
        asm volatile("" : "+s"(s_keys));
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = (4 * g + r < s_keys) ? sc[hh][g][r] : -3.0e38f;
            sc[hh][g][r] = v;
            mx4[r] = fmaxf(mx4[r], v);
          }
      }
      mx[hh] = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
    }
    // ---- numerators of a group pair, then that pair's output products o4[d] += V[key][d] * P[query][key]
    float ls4[NH][4];
    f32x4 oa[NH], ob[NH];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ls4[hh][r] = 0.f;
      oa[hh] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ob[hh] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int g = 0; g < G; g += 2) {
      float4 va[NH], vc[NH];
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        va[hh] = *reinterpret_cast<const float4*>(vb[hh] + 4 * g);
        vc[hh] = *reinterpret_cast<const float4*>(vb[hh] + 4 * g + 4);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(sc[hh][g + u][r] - mx[hh]);
            sc[hh][g + u][r] = pr;
            ls4[hh][r] += pr;
          }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        oa[hh] = mfma4(va[hh].x, sc[hh][g][0], oa[hh]);
        ob[hh] = mfma4(vc[hh].x, sc[hh][g + 1][0], ob[hh]);
      }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        oa[hh] = mfma4(va[hh].y, sc[hh][g][1], oa[hh]);
        ob[hh] = mfma4(vc[hh].y, sc[hh][g + 1][1], ob[hh]);
      }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        oa[hh] = mfma4(va[hh].z, sc[hh][g][2], oa[hh]);
        ob[hh] = mfma4(vc[hh].z, sc[hh][g + 1][2], ob[hh]);
      }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        oa[hh] = mfma4(va[hh].w, sc[hh][g][3], oa[hh]);
        ob[hh] = mfma4(vc[hh].w, sc[hh][g + 1][3], ob[hh]);
      }
    }
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
      const int head = 2 * a_hp + h0 + hh;
      const float lsum = (ls4[hh][0] + ls4[hh][1]) + (ls4[hh][2] + ls4[hh][3]);
      const float il = 1.0f / lsum;
      *reinterpret_cast<float4*>(o_lds + s_q * 16 + head * 4) =
          make_float4((oa[hh][0] + ob[hh][0]) * il, (oa[hh][1] + ob[hh][1]) * il, (oa[hh][2] + ob[hh][2]) * il,
                      (oa[hh][3] + ob[hh][3]) * il);
    }
  }
}

// Ray attention for S in (128, 256] (decoder_pp_kernel<256>): a ray spans BOTH teams — team A owns samples 0..127, team B 128..255 — so
// a wave's 64 queries (of its own team) meet the keys in two halves of 128: k0 / vt0 = team A's scratch, k1 / vt1 = team B's, each in
// the 128-sample layout of ray_attention_pp<128>.  Per half the arithmetic is that function's (scores of 4 keys per 4x4x1 group, four
// partial maxima, numerators of a group pair in front of its output products); the halves are merged flash style: running maximum
// m, sum l and un-normalised output o per head, the older part rescaled by 2^(m_old - m_new).  `first` is the half this team can
// read at once (its own: the other team's q|k|v phase may still be running), `wait_other()` returns when the other half is there.
template <bool PADDED, class Wait>
__device__ __forceinline__ void ray_attention_pp256(const float* q_lds, const float* k0, const float* vt0, const float* k1,
                                                    const float* vt1, float* o_lds, int a_hp, int s_q, int lane, int S, int first,
                                                    Wait wait_other) {
  constexpr int HS = 128, G = HS / 4;
  // four parts = (half, head) in the order (first, h0), (first, h1), (other, h0), (other, h1): ONE copy of the code in a loop that is
  // not unrolled (two heads' score sets — 128 registers each — must never be live together); the two heads' running (m, l, o)
  // are selected by the part's parity.
  float m0 = -3.0e38f, m1 = -3.0e38f, l0 = 0.f, l1 = 0.f;
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    if (it == 2) wait_other();
    const int half = it < 2 ? first : 1 - first;
    const int hsel = it & 1;
    const int head = 2 * a_hp + hsel;
    const float* kh = half == 0 ? k0 : k1;
    const float* vh = half == 0 ? vt0 : vt1;
    int n_keys = S - half * HS;  // valid keys of this half (PADDED: S < 256)
    n_keys = n_keys < 0 ? 0 : (n_keys > HS ? HS : n_keys);
    const float m_run = hsel ? m1 : m0, l_run = hsel ? l1 : l0;
    const float4 q4 = *reinterpret_cast<const float4*>(q_lds + s_q * 16 + head * 4);
    const float* kb = kh + (head * HS + (lane & 3)) * 4;
    const float* vb = vh + (head * 4 + (lane & 3)) * HS;
    f32x4 sc[G];
#pragma unroll
    for (int g = 0; g < G; g += 2) {
      const float4 ka = *reinterpret_cast<const float4*>(kb + g * 16);
      const float4 kc = *reinterpret_cast<const float4*>(kb + g * 16 + 16);
      f32x4 ta = mfma4(ka.x, q4.x, (f32x4){0.f, 0.f, 0.f, 0.f});
      f32x4 tc = mfma4(kc.x, q4.x, (f32x4){0.f, 0.f, 0.f, 0.f});
      ta = mfma4(ka.y, q4.y, ta);
      tc = mfma4(kc.y, q4.y, tc);
      ta = mfma4(ka.z, q4.z, ta);
      tc = mfma4(kc.z, q4.z, tc);
      sc[g] = mfma4(ka.w, q4.w, ta);
      sc[g + 1] = mfma4(kc.w, q4.w, tc);
    }
    float mx4[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    if constexpr (!PADDED) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx4[r] = fmaxf(mx4[r], sc[g][r]);
    } else {
      int s_keys = __builtin_amdgcn_readfirstlane(n_keys);  // wave-uniform; opaque: see decoder_kernel
      asm volatile("" : "+s"(s_keys));
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = (4 * g + r < s_keys) ? sc[g][r] : -3.0e38f;
          sc[g][r] = v;
          mx4[r] = fmaxf(mx4[r], v);
        }
    }
    const float m_new = fmaxf(m_run, fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3])));
    float ls4[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; g += 2) {
      const float4 va = *reinterpret_cast<const float4*>(vb + 4 * g);
      const float4 vc = *reinterpret_cast<const float4*>(vb + 4 * g + 4);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = __builtin_amdgcn_exp2f(sc[g + u][r] - m_new);
          sc[g + u][r] = pr;
          ls4[r] += pr;
        }
      oa = mfma4(va.x, sc[g][0], oa);
      ob = mfma4(vc.x, sc[g + 1][0], ob);
      oa = mfma4(va.y, sc[g][1], oa);
      ob = mfma4(vc.y, sc[g + 1][1], ob);
      oa = mfma4(va.z, sc[g][2], oa);
      ob = mfma4(vc.z, sc[g + 1][2], ob);
      oa = mfma4(va.w, sc[g][3], oa);
      ob = mfma4(vc.w, sc[g + 1][3], ob);
    }
    // merge with what the other half left (first half: m_run = -3e38, l_run = o_run = 0: the factor is exp2(-huge) = 0)
    const float f_old = __builtin_amdgcn_exp2f(m_run - m_new);
    const float l_new = l_run * f_old + ((ls4[0] + ls4[1]) + (ls4[2] + ls4[3]));
    f32x4 o_new;
#pragma unroll
    for (int d = 0; d < 4; ++d) o_new[d] = (hsel ? o1[d] : o0[d]) * f_old + (oa[d] + ob[d]);
    if (hsel) {
      m1 = m_new, l1 = l_new, o1 = o_new;
    } else {
      m0 = m_new, l0 = l_new, o0 = o_new;
    }
  }
  {
    const float i0 = 1.0f / l0, i1 = 1.0f / l1;
    *reinterpret_cast<float4*>(o_lds + s_q * 16 + (2 * a_hp) * 4) = make_float4(o0[0] * i0, o0[1] * i0, o0[2] * i0, o0[3] * i0);
    *reinterpret_cast<float4*>(o_lds + s_q * 16 + (2 * a_hp + 1) * 4) = make_float4(o1[0] * i1, o1[1] * i1, o1[2] * i1, o1[3] * i1);
  }
}

// FS: K16-steps of the FiLM stage = ceil(conditioning inputs / 16): 2 up to 5 source views (the shipped 3-view case), 3 for
// 6-7, 4 for 8-11 (BASELINE config[4]: 10 views, 50 inputs).  Only the first stage, its operands and its place in the weight
// stream depend on it; with FS > 2 a team's rows of the next tile (128 x cond_stride floats) no longer fit its half of ring
// slot 1, so V_0 reads them from global memory.
// POSES: the launch carries a pose table (mnerf_rays.pose_table: several target poses of a small frame in one launch); a team's
// rays of a tile belong to one pose (rays_per_pose is a multiple of 64), whose camera constants replace the launch-wide ones
// once per tile.  A separate instance so that the default kernel's scalar-register budget is untouched.
// NPK: products per MAC of the trunk (3 = the fp32-grade parity path; 1 = the reduced-precision fast mode MNERF_WSTREAM_F16X1: fp16
// operands with per-sample gains, fp32 accumulation — same stream, same schedule, the lo halves unused).
template <int SP, int FS = 2, bool POSES = false, int NPK = 3>
__global__ __launch_bounds__(512, 2) void decoder_pp_kernel(
    mnerf_decoder D, DecSched sch, PPSched pps, mnerf_view view0, mnerf_rays Rl, const float* __restrict__ cond,
    float* __restrict__ out_rgb, float* __restrict__ out_depth, float* __restrict__ out_opacity,
    float* __restrict__ dbg_rgb_s, float* __restrict__ dbg_sigma, const float* __restrict__ ext_ndc,
    const float* __restrict__ ext_dir) {
  constexpr int Sp = SP;
  using SM = SmemPP<SP>;
  constexpr int TEAM = SM::TEAM;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned ring_lds = __builtin_amdgcn_groupstaticsize();
  const unsigned tail_lds = ring_lds + SM::RING_FLOATS * 4u;
  const unsigned hdr_lds = tail_lds + SM::TAIL_F * 4u;
  float* tail = smem + SM::RING_FLOATS;                 // resident [w_qs;w_ks;w_vs | fc | out_alpha.0 | out_alpha.2]
  float* rs_all = tail + SM::TAIL_F + SM::HDR_FLOATS;   // [TILE][4] rgb.xyz, sigma.w
  float* ln_lds = rs_all + SM::RS_FLOATS;

  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, tw = wave & 3;
  const mnerf_rays& R = Rl;  // launch-wide fields (counts, sizes); the camera constants are read through Rt inside the tile loop
  const int S = R.n_samples;
  // WIDE (SP = 256, round 5): ONE ray per tile — team A owns its samples 0..127, team B 128..255; the trunk is per sample and does not
  // care, the tail meets across the teams (ray attention in two key halves, compositing of the whole ray by team B).
  constexpr bool WIDE = SP == 256;
  static_assert(!WIDE || (!POSES && FS == 2), "the 256-sample instance exists for the shipped shape without a pose table");
  const int rays_per_team = WIDE ? 0 : TEAM / Sp, rays_per_tile = WIDE ? 1 : 2 * rays_per_team;
  const int n_tiles = (R.n_rays + rays_per_tile - 1) / rays_per_tile;
  const int CS = D.cond_stride;
  const float freq_mul = R.legacy_coord ? 1.0f : 3.14159265358979323846f;
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);
  float* rs_lds = rs_all + team * TEAM * 4;
  // a team's ray-attention scratch: ring slot 3 (team A) / 2 (team B), free while that team is in its tail phases
  float* att = smem + (team == 0 ? 3 : 2) * SEG_CAP_FLOATS;
  float* k_lds = att;
  float* vt_lds = att + TEAM * 16;
  float* q_lds = att + TEAM * 32;
  float* o_lds = att + TEAM * 48;
  // WIDE: the OTHER team's keys / values and arrival counter (a wave polls it where it needs the other team's phase to be over)
  const float* k_other = smem + (team == 0 ? 2 : 3) * SEG_CAP_FLOATS;
  const float* vt_other = k_other + TEAM * 16;
  // the team's conditioning rows of the NEXT tile: half of ring slot 1 (free between the views stage and layer 1 of the next tile)
  float* rows_lds = smem + 1 * SEG_CAP_FLOATS + team * 4096;
  const unsigned rows_lds_addr = ring_lds + (unsigned)(1 * SEG_CAP_FLOATS + team * 4096) * 4u;

#define PP_SLOT_LDS(s_, i_) (ring_lds + (unsigned)(2 * ((s_)&1) + (i_)) * (SEG_CAP_FLOATS * 4u))
#define PP_HDR_LDS(s_) (hdr_lds + (unsigned)(s_) * 1024u)
#define PP_SEG_SRC(s_, i_) (D.wstream + pp_seg_off_floats(s_, i_, FS))

  // ---- prologue: LayerNorm parameters, the resident tail segment, the stage headers, stage 0 of the first tile
  if (tid < SMALL_FIXED) ln_lds[tid] = D.small_[tid];
  if (tid < 2) reinterpret_cast<unsigned*>(ln_lds)[48 + tid] = 0u;  // the teams' arrival counters (wave_group_sync)
  // this wave's team counter: LDS byte address and the number of arrivals it has made (wave-uniform)
  const unsigned team_ctr_lds = hdr_lds + (unsigned)(SM::HDR_FLOATS + SM::RS_FLOATS + 48 + team) * 4u;
  int team_epoch = 0;
  const unsigned other_ctr_lds = hdr_lds + (unsigned)(SM::HDR_FLOATS + SM::RS_FLOATS + 48 + (1 - team)) * 4u;
  int tile_it = 0;  // tiles this workgroup has finished (both teams count alike): the other team's arrival count is a function of it
  auto wait_other_team = [&](int arrivals) {  // until the other team's waves have ALL made `arrivals` team syncs (wave_group_sync)
    const int want = 4 * arrivals;
    for (;;) {
      unsigned v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(other_ctr_lds) : "memory");
      if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - (unsigned)want) >= 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
  };
  {
    const unsigned voff = (unsigned)lane0 * 16u;
    for (int p = wave; p < pp_stream_pieces(19); p += 8)
      glds16_s(D.wstream + pp_stream_off_floats(19, FS) + p * 256, voff, __builtin_amdgcn_readfirstlane(tail_lds + (unsigned)p * 1024u));
    for (int s = wave; s < PP_STAGES; s += 8) glds16_s(PP_SEG_SRC(s, 0), voff, __builtin_amdgcn_readfirstlane(PP_HDR_LDS(s)));
    for (int p = wave; p < pp_p0(0, FS); p += 8)
      glds16_s(PP_SEG_SRC(0, 0) + p * 256, voff, __builtin_amdgcn_readfirstlane(PP_SLOT_LDS(0, 0) + (unsigned)p * 1024u));
  }
  segment_wait();
  __syncthreads();
  // (A static issue priority for team B - waves 4-7 are dispatched second, and between two waves of a SIMD at equal priority
  // the OLDER one wins the vector-issue arbitration - was measured in round 4: it only moves the loss to the other team, 18.99
  // against 18.82 ms per frame; profiles/history/r4_variants.md.)
  if (team == 1) __syncthreads();  // team B runs one phase behind team A from here on

  int tile_begin = blockIdx.x, tile_end = n_tiles, tile_step = gridDim.x;
  if (gridDim.x >= 8) {
    const int xcd = blockIdx.x & 7;
    tile_begin = (int)((long long)n_tiles * xcd / 8) + (int)(blockIdx.x >> 3);
    tile_end = (int)((long long)n_tiles * (xcd + 1) / 8);
    tile_step = ((int)gridDim.x - xcd + 7) >> 3;
  }
  // Synchronisation of a phase's end.  Trunk phases end at the workgroup barrier (PP_SYNC): that is what keeps the two teams
  // exactly one phase apart.  The four tail phases only exchange data INSIDE a team (attention scratch, per-sample densities),
  // so they end at a team-scoped sync (PP_TSYNC: wave_group_sync on the team's LDS counter) and the teams run through their
  // tails independently: in round 3's lock step the five slots around the tail cost max(A, B) each — T2 (9 k cycles) was paid
  // twice, 32 k of a tile's 120 k cycles —, now they cost max(sum A, sum B).  PP_NOSYNC: a phase end without any sync.
#ifdef MNERF_TIMELINE
  // debug build (tools/exp/pp_timeline.py): per wave and phase, s_memtime at the end of the work and after the sync
  int pp_tl_tile = -1, pp_ph = 0;
#define PP_STAMPED(sync_)                                                                                      \
  do {                                                                                                         \
    unsigned long long* o_ = (sch.tl && lane0 == 0 && blockIdx.x < 32 && pp_tl_tile >= 0 && pp_tl_tile < 4)     \
                                 ? sch.tl + ((((size_t)blockIdx.x * 4 + pp_tl_tile) * 8 + wave) * PP_PHASES + pp_ph) * 2 \
                                 : nullptr;                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (o_) o_[0] = __builtin_amdgcn_s_memtime();                                                              \
    sync_;                                                                                                     \
    if (o_) o_[1] = __builtin_amdgcn_s_memtime();                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    ++pp_ph;                                                                                                   \
  } while (0)
#else
  // (a scheduling fence on both sides: the compiler otherwise moves vector work across s_barrier — in round 3's ISA the 64 bias
  // multiplies and 16 operand conversions of every vector phase sat at the top of the NEXT matrix phase, in front of its first
  // matrix instruction: ~350 cycles of start-up in the longer phase of the slot)
#define PP_STAMPED(sync_)                    \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    sync_;                                   \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)
#endif
#define PP_SYNC() PP_STAMPED(__syncthreads())
#define PP_TSYNC() PP_STAMPED(wave_group_sync(team_ctr_lds, 4, team_epoch, (int)hw_lane()))
#define PP_NOSYNC() PP_STAMPED((void)0)
// MNERF_PP_DMA_VOFF (round 5, default 1): a stage's requests take ONE stream base (D.wstream) and add the piece's byte offset to the
// per-lane offset in the vector ALU (glds16_sv) instead of ~200 distinct loop-invariant `stream + constant` pointers, which the
// compiler hoisted out of the tile loop and parked in VGPR lanes (two v_readlane per request).  Bit-identical; same-box A/B
// (gpurun_out/r5_3): 17.11 -> 16.99 ms per frame at S = 64, 36.9 -> 36.5 at S = 128.
  // `half`: 0 / 1 = the even / odd 1-KiB pieces of a stage (the two teams share the issue cost), 2 = all of them
  auto stage_dma = [&](int s, int half) {
    const unsigned voff = hw_lane() * 16u;  // (not `lane0`: see hw_lane)
    int twl = tw;  // opaque: the requests' offsets are computed where they are used (see stage_piece)
    asm volatile("" : "+s"(twl));
    for (int i = 0; i < 2; ++i) {
      const int pieces = i == 0 ? pp_p0(s, FS) : pp_p1(s);
      const unsigned base = PP_SLOT_LDS(s, i);
      const int first = half == 2 ? twl : 2 * twl + half, step = half == 2 ? 4 : 8;
#if MNERF_PP_DMA_RUNS
      // round 6: the pieces are dealt in RUNS of four consecutive ones (run r to team r & 1, wave (r >> 1) & 3; all runs to one
      // team's waves for half = 2), a run is one glds16_sv4 = one M0 write for four requests; what is left of a segment beyond
      // its last full run (0 or 1 piece in this schedule) goes to the next (team, wave) of the pattern one by one
      const int n_runs = pieces >> 2;
      for (int r = first; r <= n_runs; r += step) {
        const unsigned off = (unsigned)r * 4096u;
        if (r < n_runs)
          glds16_sv4(D.wstream, voff, (unsigned)pp_seg_off_floats(s, i, FS) * 4u + off, __builtin_amdgcn_readfirstlane(base + off));
        else
          for (int p = 4 * n_runs; p < pieces; ++p)
            glds16_sv(D.wstream, voff, (unsigned)pp_seg_off_floats(s, i, FS) * 4u + (unsigned)p * 1024u,
                      __builtin_amdgcn_readfirstlane(base + (unsigned)p * 1024u));
      }
#else
      for (int p = first; p < pieces; p += step)
        glds16_sv(D.wstream, voff, (unsigned)pp_seg_off_floats(s, i, FS) * 4u + (unsigned)p * 1024u,
                  __builtin_amdgcn_readfirstlane(base + (unsigned)p * 1024u));
#endif
    }
  };
  // ---- weight requests of a vector phase.  MNERF_PP_DMA_SPREAD = 0 (round 3): team A asks for its half of stage s (the odd
  // 1-KiB pieces) in one burst at the START of V_s, team B for its half of stage s+1 (the even ones) in one burst at the END of
  // its V_s; the LDS-DMA queue drains ~1 piece per 90 cycles, so a wave's burst of 8 stalls it ~1.6 k cycles in which it does
  // nothing else.  = 1 (experiment): the requests are SPREAD over the phase, a few behind the FiLM / ReLU pass and one behind every
  // operand split, so that the queue drains under the wave's own vector work (PP_DMA sites; what is left goes out at the end).
  // A wave's share of a stage is at most 10 requests: k -> segment k / 5, piece 2 tw + half + 8 (k % 5).
  // MEASURED (gpurun_out r4b, profiles/r4_variants.md): slower. Every request still stalls its wave (the queue is full whenever
  // eight waves feed it), now ten times per phase behind a scheduling fence each: vector phases 3.9 k -> 4.7 k cycles, decoder
  // 18.2 -> 19.0 ms per frame.  The bursts stay.
  int dma_k = 0;  // requests of the current vector phase already made by this wave (compile-time after unrolling)
  auto stage_piece = [&](int s, int half, int k) {
    const int i = k / 5;
    const int pcs = i == 0 ? pp_p0(s, FS) : pp_p1(s);
    // (an opaque copy of the wave's index: the ~200 source / destination addresses of a tile's requests are loop-invariant,
    // and hoisted out of the tile loop they were parked in VGPR lanes — 480 spilled SGPRs, two v_readlane per request; computed
    // where they are used they are two scalar additions)
    int twl = tw;
    asm volatile("" : "+s"(twl));
    const int pce = 2 * twl + half + 8 * (k % 5);
    if (pcs > 0 && pce < pcs) {
      const unsigned vo = hw_lane() * 16u;
      glds16_s(PP_SEG_SRC(s, i) + pce * 256, vo, __builtin_amdgcn_readfirstlane(PP_SLOT_LDS(s, i) + (unsigned)pce * 1024u));
    }
  };
  auto dma_some = [&](int sa, int sb, int n) {
#pragma unroll
    for (int q = 0; q < n; ++q) {
      if (dma_k >= 10) break;
      __builtin_amdgcn_sched_barrier(0);
      if (team == 0) {
        if (sa >= 0) stage_piece(sa, 1, dma_k);
      } else {
        if (sb >= 0) stage_piece(sb, 0, dma_k);
      }
      __builtin_amdgcn_sched_barrier(0);
      ++dma_k;
    }
  };
  // MNERF_PP_B_EARLY (round 4, after the issue microbenchmark tools/exp/ubench/mfma_issue.hip): team B asks for its half of
  // stage s+1 at the START of its V_s instead of at its end.  The ring slots of stage s+1 are free from that point (their last
  // reader was team B's own M_{s-1}); asked for at the end of the phase, B's burst met team A's burst at the start of A's
  // V_{s+1} right behind the barrier — 65 KiB requested at once, and both teams stalled longer per request than one team's burst
  // alone does (~95 cycles).  Measured: 16.43-16.50 against 16.51-16.68 ms per frame (same box, three runs each).  V_0 keeps the late request: its rows sit in that buffer.
#define PP_SEGMENT_WAIT() segment_wait()
  // MNERF_PP_DMA_IN_M = KM (round 4 experiment, default 0): the first KM of a wave's (up to 10) requests for stage n go out INSIDE
  // the matrix phase M_{n-1}, behind its matrix instructions (pp_m_dma below), the rest in the burst of the vector phase.  A
  // request stalls the issuing wave ~95 cycles in a vector phase and ~50 among matrix instructions
  // (tools/exp/ubench/mfma_issue.hip).  MEASURED: no gain (KM = 4: 16.45-16.53, 2: 16.54-16.76, 0: 16.35-16.43 ms per frame) — the
  // two phases of a slot are equally long, what one side saves the other pays.  Stages whose previous matrix phase has no hook
  // keep all requests in the burst.
  auto burst_from = [&](int s, int half, int k0) {
    if (k0 == 0) {
      stage_dma(s, half);
      return;
    }
#pragma unroll
    for (int k = k0; k < 10; ++k) stage_piece(s, half, k);
  };
  // requests of matrix phase M_s (n_units units of three matrix instructions): stage s+1; team A spreads its KM over all units,
  // team B over the first half (its pieces are needed one slot earlier: team A's M_{s+1} starts when this phase ends)
  // (the per-lane offset of these requests is the matrix phase's own fragment address register: see ksteps_presplit2)
  // NO control flow here: a branch inside the matrix phase splits it into basic blocks, and the compiler then sinks the tail of
  // the previous vector phase (48 bias multiplies, the last operand conversions) into them, behind the first matrix
  // instructions, and spills around the blocks (23 -> 63 spilled registers in the first version).  So: pp_km(n) only counts
  // requests that exist for every wave of both teams (piece 2 tw + half + 8 k < the first segment's piece count), the teams
  // differ in a scalar operand (half) instead of a branch, and both front-load their requests into the first half of the units
  // (team B's must have landed when the phase ends: team A's M_{s+1} starts there).
  auto stage_piece_m = [&](int s, int k, unsigned lane_addr, unsigned lane_addr_base) {
    int twl = tw;
    asm volatile("" : "+s"(twl));
    const int pce = 2 * twl + (1 - team) + 8 * k;  // team A: the odd pieces, team B: the even ones (as in the bursts)
    glds16_s(reinterpret_cast<const float*>(reinterpret_cast<const char*>(PP_SEG_SRC(s, 0) + pce * 256) - lane_addr_base), lane_addr,
             __builtin_amdgcn_readfirstlane(PP_SLOT_LDS(s, 0) + (unsigned)pce * 1024u));
  };
  auto pp_m_dma = [&](int s, int n_units, int i, unsigned lane_addr, unsigned lane_addr_base) {
#pragma unroll
    for (int j = 0; j < pp_km(s + 1); ++j)
      if ((j * (n_units / 2)) / pp_km(s + 1) == i) stage_piece_m(s + 1, j, lane_addr, lane_addr_base);
  };
#define PP_BEGIN_V(stage_)                                                          \
  do {                                                                              \
    if (team == 0) burst_from(stage_, 1, pp_km(stage_));                            \
    else if ((stage_) + 1 < PP_STAGES) burst_from((stage_) + 1, 0, pp_km((stage_) + 1)); \
  } while (0)
  (void)dma_some;
#define PP_DMA(sa_, sb_, n_) do {} while (0)
#define PP_END_V(sa_, next_stage_)                                  \
  do {                                                              \
    if (team == 1) { if ((next_stage_) >= 0 && (sa_) < 0) stage_dma(next_stage_, 0); } \
    else PP_SEGMENT_WAIT();                                         \
    PP_SYNC();                                                      \
  } while (0)
#define PP_END_M()                   \
  do {                               \
    if (team == 1) PP_SEGMENT_WAIT();   \
    PP_SYNC();                       \
  } while (0)
  // M_s: stage s's matrix instructions (+ team A: the requests for stage NXT_)
// MNERF_PP_MPRIO (experiment, default 0): issue priority of a wave while it is in a matrix phase — its MFMAs then win the
// SIMD's arbitration against the other team's vector instructions.
#define PP_PRIO_UP() do {} while (0)
#define PP_PRIO_DOWN() do {} while (0)
#define PP_MFMA_(NMB_, NS0_, NS1_, acc_, base0_, base1_, hs_) ksteps_presplit2<NMB_, NS0_, NS1_, NPK>(acc_, base0_, base1_, lane, hs_)
#define PP_MFMA_HOOKED(NMB_, NS0_, NS1_, acc_, s_, hdr_bytes_, hs_)                                                      \
  ksteps_presplit2<NMB_, NS0_, NS1_, NPK>(acc_, PP_SLOT_LDS(s_, 0) + (hdr_bytes_), PP_SLOT_LDS(s_, 1), lane, hs_,        \
                                     [&](int i_, unsigned la_, unsigned lb_) { pp_m_dma(s_, ((NS0_) + (NS1_)) * (NMB_), i_, la_, lb_); })
#define PP_MFMA(NMB_, NS0_, NS1_, acc_, s_, hdr_bytes_, hs_)                                                   \
  do {                                                                                                         \
    PP_PRIO_UP();                                                                                              \
    PP_MFMA_HOOKED(NMB_, NS0_, NS1_, acc_, s_, hdr_bytes_, hs_);                                                \
    PP_PRIO_DOWN();                                                                                            \
  } while (0)
#define PP_MFMA1(NS0_, acc_, s_, hdr_bytes_, hs_)                                                              \
  do {                                                                                                         \
    PP_PRIO_UP();                                                                                              \
    PP_MFMA_HOOKED(1, NS0_, 0, acc_, s_, hdr_bytes_, hs_);                                                      \
    PP_PRIO_DOWN();                                                                                            \
  } while (0)

  bool rows_in_lds = false;  // this tile's conditioning rows were copied to LDS during the previous tile's tail
  for (int tile = tile_begin; tile < tile_end; tile += tile_step) {
#ifdef MNERF_TIMELINE
    ++pp_tl_tile;
    pp_ph = 0;
#endif
    const bool has_next = tile + tile_step < tile_end;
    // this team's target camera for the tile: the launch's, or (POSES) its pose's row of the table, fetched where it is used
    int tile_pose = 0;
    if constexpr (POSES)
      tile_pose = pose_of_ray(Rl, (tile * rays_per_tile + team * rays_per_team < Rl.n_rays) ? tile * rays_per_tile + team * rays_per_team : Rl.n_rays - 1);
#define PP_TILE_RAYS(NAME_)                                  \
  mnerf_rays NAME_ = Rl;                                     \
  if constexpr (POSES) rays_for_pose(NAME_, Rl, tile_pose)
    // Everything that depends only on the lane index is re-derived per tile from an OPAQUE copy of it: hoisted out of the
    // tile loop these values (sample indices, LDS addresses, pointers) stay live across all 28 phases, the register
    // allocator parks them in scratch, and phase V_0 became a chain of ~20 scratch reloads (7 k cycles, measured).
    const int lane = (int)hw_lane();
    const int n = lane & 31, hl = lane >> 5;
    const unsigned voff = (unsigned)lane * 16u;
    // ------------------------------------------------------------ per-lane sample identity
    const int s_local = tw * 32 + n;                  // sample within the team
    const int ray_t = WIDE ? 0 : s_local / Sp;        // ray within the team
    const int jp = WIDE ? team * TEAM + s_local : s_local - ray_t * Sp;
    const int ray_raw = tile * rays_per_tile + team * rays_per_team + ray_t;
    const bool ray_ok = ray_raw < R.n_rays;
    const int ray = ray_ok ? ray_raw : (R.n_rays - 1);
    const int j = jp < S ? jp : (S - 1);
    const size_t gs = (size_t)ray * S + j;

    // ============================================================ phase 0 = V_0: inputs, FiLM operands
    float4 cpre[2 * FS];
    float n_valid;
    {
      // the sample's row: in LDS when the previous tile's tail copied it there (FS == 2 only), else in global memory
      const float* crow = (FS == 2 && rows_in_lds) ? rows_lds + s_local * CS : cond + gs * CS;
#pragma unroll
      for (int i = 0; i < 2 * FS; ++i) {
        const int o = 16 * (i >> 1) + 8 * hl + 4 * (i & 1);
        cpre[i] = (o + 4 <= CS) ? *reinterpret_cast<const float4*>(crow + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float* mrow = crow + (D.cond_dim - D.n_views);
      float nv = 0.0f;
      for (int v = 0; v < D.n_views; ++v) nv += mrow[v];
      n_valid = nv;
    }
    // (only x, y, z stay in registers across the trunk: the two-float encoding bases are re-derived where the encoding is
    // evaluated, the view direction where the views stage needs it — kept alive from here they were spilled and reloaded.
    // The sample's coordinates themselves are NOT needed before V_1: they are evaluated in phase M_0, behind its 24 matrix
    // instructions — the shortest matrix phase of the tile, whose slot is as long as the other team's V_0 anyway.)
    float x, y, z, enc_max;
    const bool q_valid = n_valid > 1.0f;

    dma_k = 0;     // (V_0 has no PP_BEGIN_V: team A requests nothing here, team B its half of stage 1)
    PartsH hs[8];  // operands of the coming matrix stage: 8 K16-steps of fp16 hi | lo fragments (64 registers)
    f32x16 film[4];
    int ecf;
    {
#pragma unroll
      for (int t = 0; t < FS; ++t) {
        const float vt[8] = {cpre[2 * t].x, cpre[2 * t].y, cpre[2 * t].z, cpre[2 * t].w,
                             cpre[2 * t + 1].x, cpre[2 * t + 1].y, cpre[2 * t + 1].z, cpre[2 * t + 1].w};
        hs[t] = split8h(vt, (float)(1 << (H16_TARGET_EXP - 1)));
      }
      const int ew = header_ew(PP_HDR_LDS(0));
      ecf = -(ew + (H16_TARGET_EXP - 1));
      bias_init_h<4>(film, PP_HDR_LDS(0), hl, pow2i(ew + (H16_TARGET_EXP - 1)));
    }
    PP_END_V(-1, 1);
    // ============================================================ phase 1 = M_0: FiLM = pts_bias(cond)
    PP_MFMA(4, FS, 0, film, 0, 1024u, hs);
    __builtin_amdgcn_sched_barrier(0);  // the geometry stays in this phase (the scheduler moves vector work across s_barrier)
    if (ext_ndc) {
      x = ext_ndc[gs * 3 + 0];
      y = ext_ndc[gs * 3 + 1];
      z = ext_ndc[gs * 3 + 2];
    } else {
      PP_TILE_RAYS(Rt);
      const RayGeom g = make_ray(Rt, ray);
      const float dpt = sample_depth(Rt, ray, j);
      float wx_, wy_, wz_;
      ray_point(g, dpt, wx_, wy_, wz_);
      project(view0, wx_, wy_, wz_, wm1, hm1, x, y, z);
    }
    enc_max = fmaxf(fmaxf(1.0f, fabsf(x)), fmaxf(fabsf(y), fabsf(z)));
    __builtin_amdgcn_sched_barrier(0);
    PP_END_M();
    // ============================================================ phase 2 = V_1: positional encoding operands (layer 0)
    PP_BEGIN_V(1);
    f32x16 acc[4];
    int ew_cur = 0, ec, em;
    // 16 of the 32 encoding operands -> hs[2 half], hs[2 half + 1].  The two halves are evaluated in different phases: the first
    // in the vector phase in front of the stage, the second INSIDE the matrix phase, between the stage's first and second pair of
    // K16-steps (48 matrix instructions in all: the phase is short, while a vector phase with the whole encoding was the longest
    // of the trunk).  Same operands, same K-step order: same bits.
    auto split_enc_half = [&](int half, float mult) {
      // (opaque copies: otherwise the second evaluation, six stages later, is merged with the first one and the 32 values
      // travel through scratch — 33 stores in V_1, 53 scratch operations with 35 separate waits in V_6: 12 k cycles)
      float xo = x, yo = y, zo = z;
      asm volatile("" : "+v"(xo), "+v"(yo), "+v"(zo));
      const EncBase encb = enc_base(xo, yo, zo, freq_mul);
      const f32x16 e = half == 0 ? enc_block16_L10<0>(encb, hl, xo, yo, zo) : enc_block16_L10<16>(encb, hl, xo, yo, zo);
      float v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = e[r];
      hs[2 * half] = split8h(v, mult);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = e[8 + r];
      hs[2 * half + 1] = split8h(v, mult);
    };
    // the matrix phase of an encoding stage (4 K16-steps x 4 blocks in one segment behind a header): steps 0-1, second half of
    // the operands, steps 2-3
#define PP_ENC_MFMA(acc_, s_, mult_)                                                                             \
  do {                                                                                                           \
    PP_PRIO_UP();                                                                                                \
    PP_MFMA_(4, 2, 0, acc_, PP_SLOT_LDS(s_, 0) + 1024u, PP_SLOT_LDS(s_, 1), hs);                                 \
    PP_PRIO_DOWN();                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    split_enc_half(1, mult_);                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    PP_PRIO_UP();                                                                                                \
    PP_MFMA_(4, 2, 0, acc_, PP_SLOT_LDS(s_, 0) + 1024u + 8 * H16_UNIT_BYTES, PP_SLOT_LDS(s_, 1), hs + 2);         \
    PP_PRIO_DOWN();                                                                                              \
  } while (0)
    // 128 operands held in four 16-register blocks -> hs[0..7]; with MNERF_PP_DMA_SPREAD the wave's weight requests of the phase
    // (team A: stage sa, team B: stage sb) go out in front of the first split (three of them) and behind every split (one each)
    auto split_blocks = [&](const f32x16 (&b)[4], float mult, int sa, int sb) {
      PP_DMA(sa, sb, 3);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = b[m][r];
        hs[2 * m] = split8h(v, mult);
        PP_DMA(sa, sb, 1);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = b[m][8 + r];
        hs[2 * m + 1] = split8h(v, mult);
        PP_DMA(sa, sb, 1);
      }
    };
    // dst <- max(acc * film, 0) (dst may be acc itself: the accumulators are dead once the activations exist, and a
    // separate copy would cost 64 registers); returns the sample's largest new activation
    auto film_relu = [&](f32x16 (&dst)[4]) -> float {
      float mx = 0.0f;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float t = fmaxf(acc[m][r] * film[m][r], 0.0f);
          dst[m][r] = t;
          mx = fmaxf(mx, t);
        }
      return pair_max(mx);
    };
    em = gain_exp(enc_max);
    const float mult_l0 = pow2i(em);
    PP_DMA(1, 2, 4);
    split_enc_half(0, mult_l0);
    PP_DMA(1, 2, 4);
    ew_cur = header_ew(PP_HDR_LDS(1));
    bias_init_h<4>(acc, PP_HDR_LDS(1), hl, pow2i(ew_cur + em));
    ec = -em - ew_cur + ecf;
    PP_END_V(1, 2);
    // ============================================================ phase 3 = M_1: layer 0
    PP_ENC_MFMA(acc, 1, mult_l0);
    PP_END_M();
    // ============================================================ layers 1..4: V_s (FiLM, ReLU, gain, split, next biases), M_s
    float hmax;
#define PP_HIDDEN_LAYER(ST_)                                                \
  do {                                                                      \
    PP_BEGIN_V(ST_);                                                        \
    hmax = film_relu(acc);                                                  \
    em = gain_exp(hmax);                                                    \
    split_blocks(acc, pow2i(em), ST_, (ST_) + 1);                           \
    ew_cur = header_ew(PP_HDR_LDS(ST_));                                    \
    bias_init_h<4>(acc, PP_HDR_LDS(ST_), hl, pow2i(ew_cur + (em - ec)));    \
    ec = ec - em - ew_cur + ecf;                                            \
    PP_END_V(ST_, (ST_) + 1);                                               \
    PP_MFMA(4, 4, 4, acc, ST_, 1024u, hs);            \
    PP_END_M();                                                             \
  } while (0)
    PP_HIDDEN_LAYER(2);
    PP_HIDDEN_LAYER(3);
    PP_HIDDEN_LAYER(4);
    PP_HIDDEN_LAYER(5);
#undef PP_HIDDEN_LAYER
    // ============================================================ layer 5 = [enc, h] -> 128 (stage 6: activation half, 7: encoding half)
    // V_6: FiLM + ReLU of layer 4 in place, its operands with the gain BOTH halves share, the layer's biases (they sit in the
    // header of the encoding half's segment, resident as header 7)
    int eg;
    PP_BEGIN_V(6);
    {
      hmax = film_relu(acc);
      eg = gain_exp(fmaxf(enc_max, hmax * pow2i(ec)));
      split_blocks(acc, pow2i(eg + ec), 6, 7);
      ew_cur = header_ew(PP_HDR_LDS(7));
      bias_init_h<4>(acc, PP_HDR_LDS(7), hl, pow2i(ew_cur + eg));
    }
    PP_END_V(6, 7);
    PP_MFMA(4, 4, 4, acc, 6, 0u, hs);  // M_6: W5[:, enc:] . h
    PP_END_M();
    PP_BEGIN_V(7);
    const float mult_l5 = pow2i(eg);
    PP_DMA(7, 8, 4);
    split_enc_half(0, mult_l5);  // V_7: the encoding operands, re-evaluated (first half; the second one inside M_7)
    ec = -eg - ew_cur + ecf;
    PP_END_V(7, 8);
    PP_ENC_MFMA(acc, 7, mult_l5);  // M_7: += W5[:, :enc] . enc
    PP_END_M();
    // ============================================================ alpha head (stage 8) and feature_linear (stage 9): same operands
    PP_BEGIN_V(8);
    hmax = film_relu(acc);  // V_8
    const int em5 = gain_exp(hmax);
    split_blocks(acc, pow2i(em5), 8, 9);
    f32x16 al[1];
    const int ew_a = header_ew(PP_HDR_LDS(8));
    bias_init_h<1>(al, PP_HDR_LDS(8), hl, pow2i(ew_a + em5 - ec));
    PP_END_V(8, 9);
    PP_MFMA1(8, al, 8, 1024u, hs);  // M_8: alpha head 128 -> 16
    PP_END_M();
    float av[8];
    PP_BEGIN_V(9);
    {  // V_9: the alpha activations (they wait in 8 registers for the ray transformer); biases of feature_linear
      const float ca = pow2i(ec - em5 - ew_a);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float t = al[0][r] * ca;
        t = D.raytrans_elu ? (t > 0.0f ? t : (expf(t) - 1.0f)) : fmaxf(t, 0.0f);
        av[r] = t;
      }
      if (D.raytrans_posenc) {
        const float* tab = D.small_ + SMALL_FIXED + (size_t)j * 16;
#pragma unroll
        for (int r = 0; r < 8; ++r) av[r] += tab[(r & 3) + 8 * (r >> 2) + 4 * hl];
      }
      ew_cur = header_ew(PP_HDR_LDS(9));
      bias_init_h<4>(acc, PP_HDR_LDS(9), hl, pow2i(ew_cur + (em5 - ec)));
    }
    const int ecfeat = ec - em5 - ew_cur;
    PP_END_V(9, 10);
    PP_MFMA(4, 4, 4, acc, 9, 1024u, hs);  // M_9: feature_linear 128 -> 128, no activation
    PP_END_M();
    // ============================================================ views_linear (stage 10): [feat, dir] -> 64
    int egv, ecv;
    PartsH hsd;
    f32x16 hv[2];
    PP_BEGIN_V(10);
    {  // V_10
      float dx, dy, dz;  // view direction in the frame of source view 0 (matchnerf.py:129-131)
      if (ext_ndc) {
        dx = ext_dir[gs * 3 + 0];
        dy = ext_dir[gs * 3 + 1];
        dz = ext_dir[gs * 3 + 2];
      } else {
        PP_TILE_RAYS(Rt);
        const RayGeom g = make_ray(Rt, ray);
        const float rn = fmaxf(sqrtf(g.rx * g.rx + g.ry * g.ry + g.rz * g.rz), 1e-12f);
        const float ux = g.rx / rn, uy = g.ry / rn, uz = g.rz / rn;
        dx = ux * view0.extr[0] + uy * view0.extr[1] + uz * view0.extr[2];
        dy = ux * view0.extr[4] + uy * view0.extr[5] + uz * view0.extr[6];
        dz = ux * view0.extr[8] + uy * view0.extr[9] + uz * view0.extr[10];
      }
      egv = gain_exp(fmaxf(1.0f, sample_absmax<4>(acc) * pow2i(ecfeat)));
      split_blocks(acc, pow2i(egv + ecfeat), 10, 11);
      const float v[8] = {hl ? 0.0f : dx, hl ? 0.0f : dy, hl ? 0.0f : dz, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      hsd = split8h(v, pow2i(egv));
      const int ew = header_ew(PP_HDR_LDS(10));
      bias_init_h<2>(hv, PP_HDR_LDS(10), hl, pow2i(ew + egv));
      ecv = -egv - ew;
    }
    PP_END_V(10, 11);
    PP_MFMA(2, 4, 4, hv, 10, 1024u, hs);  // M_10
    ksteps_presplit<2, 1>(hv, PP_SLOT_LDS(10, 1) + 8 * H16_UNIT_BYTES, lane, &hsd);
    PP_END_M();
    // ============================================================ rgb_linear (stage 11): 64 -> 3, sigmoid
    f32x16 c3[1];
    float cc;
    PP_BEGIN_V(11);
    {  // V_11
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[m][r] = fmaxf(hv[m][r], 0.0f);
      const int emr = gain_exp(sample_absmax<2>(hv));
      const float mult = pow2i(emr);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = hv[m][r];
        hs[2 * m] = split8h(v, mult);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = hv[m][8 + r];
        hs[2 * m + 1] = split8h(v, mult);
      }
      const int ew = header_ew(PP_HDR_LDS(11));
      bias_init_h<1>(c3, PP_HDR_LDS(11), hl, pow2i(ew + emr - ecv));
      cc = pow2i(ecv - emr - ew);
    }
    PP_END_V(11, -1);
    {  // M_11
      PP_MFMA1(4, c3, 11, 1024u, hs);
      if (hl == 0) {
        const float cr = 1.0f / (1.0f + expf(-c3[0][0] * cc));
        const float cg = 1.0f / (1.0f + expf(-c3[0][1] * cc));
        const float cb = 1.0f / (1.0f + expf(-c3[0][2] * cc));
        rs_lds[s_local * 4 + 0] = cr;
        rs_lds[s_local * 4 + 1] = cg;
        rs_lds[s_local * 4 + 2] = cb;
        if (dbg_rgb_s && ray_ok && jp < S) {
          dbg_rgb_s[gs * 3 + 0] = cr;
          dbg_rgb_s[gs * 3 + 1] = cg;
          dbg_rgb_s[gs * 3 + 2] = cb;
        }
      }
    }
    // From here to the end of T4 a team only exchanges data with itself.  Team A leaves M_11 through the workgroup barrier
    // that team B leaves V_11 through (the last lock-step hand-over of the tile); team B's M_11 ends at its own team sync
    // (its attention scratch is ring slot 2: the rgb weights every wave of the team has just finished reading).
    if (team == 1) {
      segment_wait();
      PP_TSYNC();
    } else {
      PP_SYNC();
    }
    // ============================================================ phase 24 = T1: q | k | v of the ray transformer -> team scratch
    // The team's conditioning rows of the next tile travel to LDS meanwhile (its half of ring slot 1, free between the views
    // stage and layer 1 of the next tile): phase V_0 then starts from LDS instead of waiting ~8 k cycles for global loads.
    bool next_rows = false;
    if (FS == 2 && S == Sp && has_next) {
      const int first_ray = (tile + tile_step) * rays_per_tile + team * rays_per_team;
      next_rows = WIDE ? first_ray < R.n_rays : first_ray + rays_per_team <= R.n_rays;
      if (next_rows) {
        const float* src = cond + ((size_t)first_ray * S + (WIDE ? team * TEAM : 0)) * CS;
        const int pieces = (TEAM * CS) >> 8;  // CS is a multiple of 8: 128 rows = CS / 2 KiB
        for (int p = tw; p < pieces; p += 4) glds16_s_stream(src + p * 256, voff, __builtin_amdgcn_readfirstlane(rows_lds_addr + (unsigned)p * 1024u));
      }
    }
    f32x16 qkv[2];
    qkv[0] = (f32x16)(0.0f);
    qkv[1] = (f32x16)(0.0f);
#pragma unroll
    for (int r = 0; r < 8; ++r) step2(qkv, tail + TAIL_QKV, r, lane, av[r]);
    const float qs = q_valid ? 0.5f * 1.4426950408889634f : 0.0f;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int head = hl + 2 * hh;
      // (WIDE: a team's scratch holds ITS 128 samples in the 128-sample layout)
      constexpr int SPT = WIDE ? TEAM : SP;
      const int jt = WIDE ? s_local : jp;
      *reinterpret_cast<float4*>(k_lds + ((ray_t * 4 + head) * SPT + jt) * 4) =
          make_float4(qkv[0][8 + 4 * hh], qkv[0][9 + 4 * hh], qkv[0][10 + 4 * hh], qkv[0][11 + 4 * hh]);
      float* vcol = vt_lds + (ray_t * 4 + head) * 4 * SPT + jt;
      vcol[0] = qkv[1][4 * hh];
      vcol[SPT] = qkv[1][4 * hh + 1];
      vcol[2 * SPT] = qkv[1][4 * hh + 2];
      vcol[3 * SPT] = qkv[1][4 * hh + 3];
      *reinterpret_cast<float4*>(q_lds + s_local * 16 + head * 4) =
          make_float4(qkv[0][4 * hh] * qs, qkv[0][4 * hh + 1] * qs, qkv[0][4 * hh + 2] * qs, qkv[0][4 * hh + 3] * qs);
    }
    PP_TSYNC();
    // ============================================================ phase 25 = T2: ray attention on the matrix pipe (lane = query)
    {
      int a_ray, a_hp, a_jq;
      if constexpr (SP >= 64) {
        constexpr int CH = (WIDE ? TEAM : SP) / 64;  // WIDE: the team's 128 queries, as at SP = 128
        int idx = tw;
        const int chunk = idx % CH;
        idx /= CH;
        a_hp = idx & 1;
        a_ray = idx >> 1;
        a_jq = chunk * 64 + lane;
      } else {
        a_ray = tw;
        a_hp = lane >> 5;
        a_jq = lane & 31;
      }
      if constexpr (WIDE) {
        // half 0 = team A's keys, half 1 = team B's.  A team reads its OWN half at once (its q|k|v phase ended at the team sync
        // above) and the other half when the other team's waves have all arrived at the sync behind THEIR q|k|v phase: per tile
        // team A makes 3 team syncs (behind T1, T2, T3), team B 4 (one more in front of T1).
        const float* k0 = team == 0 ? k_lds : k_other;
        const float* vt0 = team == 0 ? vt_lds : vt_other;
        const float* k1 = team == 0 ? k_other : k_lds;
        const float* vt1 = team == 0 ? vt_other : vt_lds;
        const int other_t1 = team == 0 ? 4 * tile_it + 2 : 3 * tile_it + 1;
        auto wait = [&]() { wait_other_team(other_t1); };
        if (S == Sp)
          ray_attention_pp256<false>(q_lds, k0, vt0, k1, vt1, o_lds, a_hp, a_jq, lane, S, team, wait);
        else
          ray_attention_pp256<true>(q_lds, k0, vt0, k1, vt1, o_lds, a_hp, a_jq, lane, S, team, wait);
      } else {
        if (S == Sp)
          ray_attention_pp<SP, false>(q_lds, k_lds, vt_lds, o_lds, a_ray, a_hp, a_ray * Sp + a_jq, lane, S);
        else
          ray_attention_pp<SP, true>(q_lds, k_lds, vt_lds, o_lds, a_ray, a_hp, a_ray * Sp + a_jq, lane, S);
      }
    }
    segment_wait();  // the team's rows of the next tile
    PP_TSYNC();
    // ============================================================ phase 26 = T3: fc + residual + LayerNorm, density head
    {
      float ofc[8];
      {
        const float4* src = reinterpret_cast<const float4*>(o_lds + s_local * 16 + 8 * hl);
        const float4 t0 = src[0], t1 = src[1];
        ofc[0] = t0.x; ofc[1] = t0.y; ofc[2] = t0.z; ofc[3] = t0.w;
        ofc[4] = t1.x; ofc[5] = t1.y; ofc[6] = t1.z; ofc[7] = t1.w;
      }
      float yv[8];
      {
        f32x16 t1 = (f32x16)(0.0f);
#pragma unroll
        for (int t = 0; t < 8; ++t) step1(t1, tail + TAIL_FCO, t, lane, ofc[t]);
        float xs = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          yv[r] = t1[r] + av[r];
          xs += yv[r];
        }
        xs = pair_sum(xs);
        const float mean = xs * (1.0f / 16.0f);
        float var = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float dlt = yv[r] - mean;
          var += dlt * dlt;
        }
        var = pair_sum(var);
        const float rstd = 1.0f / sqrtf(var * (1.0f / 16.0f) + 1e-6f);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int o = (r & 3) + 8 * (r >> 2) + 4 * hl;
          yv[r] = (yv[r] - mean) * rstd * ln_lds[o] + ln_lds[16 + o];
        }
      }
      float sigma;
      {
        f32x16 t2 = (f32x16)(0.0f);
#pragma unroll
        for (int r = 0; r < 8; ++r) step1(t2, tail + TAIL_OA0, r, lane, yv[r]);
        step1(t2, tail + TAIL_OA0, 8, lane, hl ? 0.0f : 1.0f);
        f32x16 t3 = (f32x16)(0.0f);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float t = t2[r];
          t = D.raytrans_elu ? (t > 0.0f ? t : (expf(t) - 1.0f)) : fmaxf(t, 0.0f);
          step1(t3, tail + TAIL_OA2, r, lane, t);
        }
        step1(t3, tail + TAIL_OA2, 8, lane, hl ? 0.0f : 1.0f);
        sigma = fmaxf(t3[0], 0.0f);
      }
      if (D.density_maskfill && n_valid < 1.0f) sigma = 0.0f;
      if (hl == 0) {
        rs_lds[s_local * 4 + 3] = sigma;
        if (dbg_sigma && ray_ok && jp < S) dbg_sigma[gs] = sigma;
      }
    }
    // team B requests stage 0 of the next tile (ring slot 0: the views weights were last read four steps ago)
    if (team == 1 && has_next) stage_dma(0, 2);
    PP_TSYNC();
    // ============================================================ phase 27 = T4: compositing (one wavefront per ray)
    PP_TILE_RAYS(Rt);
    // WIDE: the tile's one ray is composited by team B's first wave, from both teams' halves of rs_all (contiguous: sample jp
    // of the ray is entry jp), once team A's waves have arrived at the sync behind their T3 (their densities are written).
    const int t4_rays = WIDE ? ((team == 1 && tw == 0) ? 1 : 0) : rays_per_team;
    const float* rs_ray = WIDE ? rs_all : rs_lds;
    if (WIDE && t4_rays) wait_other_team(3 * tile_it + 3);
    for (int rt = WIDE ? 0 : tw; rt < t4_rays; rt += 4) {
      const int rr = tile * rays_per_tile + team * rays_per_team + rt;
      if (rr >= R.n_rays || !out_rgb) continue;
      float rlen = 1.0f;
      if (!D.wo_render_interval) {
        const RayGeom gg = make_ray(Rt, rr);
        rlen = sqrtf(gg.rx * gg.rx + gg.ry * gg.ry + gg.rz * gg.rz);
      }
      float carry = 0.f, ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, ao = 0.f;
      for (int j0 = 0; j0 < S; j0 += 64) {
        const int jj = j0 + lane;
        const bool ok = jj < S;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        float dd = 0.f;
        if (ok) {
          c = reinterpret_cast<const float4*>(rs_ray)[rt * Sp + jj];
          dd = sample_depth(Rt, rr, jj);
          if (!D.wo_render_interval) {
            const float intv = (jj + 1 < S) ? (sample_depth(Rt, rr, jj + 1) - dd) : 1e10f;
            c.w = c.w * (intv * rlen);
          }
        }
        const float excl = carry + wave_exclusive_sum_dpp(c.w, lane);
        const float w = ok ? expf(-excl) * (1.0f - expf(-c.w)) : 0.0f;
        ar += w * c.x;
        ag += w * c.y;
        ab += w * c.z;
        ad += w * dd;
        ao += w;
        carry = readlane_f(excl + c.w, 63);
      }
      ar = wave_sum_dpp(ar);
      ag = wave_sum_dpp(ag);
      ab = wave_sum_dpp(ab);
      ad = wave_sum_dpp(ad);
      ao = wave_sum_dpp(ao);
      if (lane == 0) {
        const float bg = D.setbg_opaque ? (1.0f - ao) : 0.0f;
        out_rgb[(size_t)rr * 3 + 0] = ar + bg;
        out_rgb[(size_t)rr * 3 + 1] = ag + bg;
        out_rgb[(size_t)rr * 3 + 2] = ab + bg;
        out_depth[rr] = ad;
        out_opacity[rr] = ao;
      }
    }
    rows_in_lds = next_rows;
    ++tile_it;
    // Team A goes straight on to V_0 of its next tile (inputs from its own rows in LDS, no weights) and meets team B at the
    // end of that phase; team B's T4 ends at that same workgroup barrier, behind its wait for stage 0 of the next tile.
    if (team == 1) {
      segment_wait();
      PP_SYNC();
    } else {
      PP_NOSYNC();
    }
  }
  if (team == 0) __syncthreads();  // team A's matching last barrier
#undef PP_END_V
#undef PP_DMA
#undef PP_BEGIN_V
#undef PP_END_M
#undef PP_SYNC
#undef PP_TSYNC
#undef PP_NOSYNC
#undef PP_STAMPED
#undef PP_ENC_MFMA
#undef PP_MFMA_
#undef PP_PRIO_UP
#undef PP_PRIO_DOWN
#undef PP_MFMA
#undef PP_MFMA1
#undef PP_HDR_LDS
#undef PP_TILE_RAYS
#undef PP_SEG_SRC
#undef PP_SLOT_LDS
}
#endif  // MNERF_DECODER_PART == 0 (ping-pong form)


// ------------------------------------------------------------------ host side
// Segment schedules shared with the Python packers (matchnerf_amd/cond_nerf.py).
static void finish_schedule(DecSched* sch, int n, int film_steps, int enc_steps) {
  sch->n_seg = n;
#ifdef MNERF_TIMELINE
  sch->tl = nullptr;
  if (const char* e = getenv("MNERF_TIMELINE_PTR")) sch->tl = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
#ifdef MNERF_FUSED_DEBUG
  {
    auto envp = [](const char* k) -> unsigned long long {
      const char* e = getenv(k);
      return (e && *e) ? strtoull(e, nullptr, 0) : 0ull;
    };
    sch->dbg_flags = (unsigned)envp("MNERF_FDBG_FLAGS");
    sch->dbg_rows = (float*)envp("MNERF_FDBG_ROWS");
    sch->dbg_nv = (float*)envp("MNERF_FDBG_NV");
    sch->dbg_tile = (unsigned*)envp("MNERF_FDBG_TILE");
  }
#endif
  sch->stagger_sleeps = mnerf_tune().decoder_stagger;  // ~130k cycles ~ half a tile
  sch->stagger_mode = mnerf_tune().decoder_stagger_mode;
  sch->film_steps = film_steps;
  sch->enc_steps = enc_steps;
}

// split formats (bf16x3: PARTS = 3, fp16x2: PARTS = 2): stages (blocks, K16-steps per segment, header)
static int build_schedule_split(const mnerf_decoder* D, DecSched* sch, int parts) {
  const int tf = (D->cond_dim + 15) / 16, te = (3 * D->L_3D + 2 + 7) / 8;
  const int per = parts == 3 ? 2 : 4;  // K16-steps per segment of a 4-block stage (<= 33 KiB with the header)
  int n = 0;
  long long off = 0;
  auto add = [&](int nmb, int steps, bool hdr) -> bool {
    if (n >= MAX_SEGS - 1) return false;
    const int fl = (steps * nmb * parts + (hdr ? 1 : 0)) * 256;
    if (fl > SEG_CAP_FLOATS) return false;
    sch->seg_off[n] = (int)off;
    sch->seg_floats[n] = fl;
    sch->seg_steps[n] = steps;
    off += fl;
    ++n;
    return true;
  };
  auto add_chunks = [&](int t, bool hdr) -> bool {  // stage of 4 blocks cut into segments of `per` K16-steps
    for (int k = 0; k < t; k += per)
      if (!add(4, (t - k) >= per ? per : (t - k), hdr && k == 0)) return false;
    return true;
  };
  bool ok = add_chunks(tf, true) && add_chunks(te, true);
  for (int l = 1; l <= 4 && ok; ++l) ok = add_chunks(8, true);
  ok = ok && add_chunks(te, true) && add_chunks(8, false);  // l5e, l5h
  ok = ok && add(1, 8, true);                               // alpha
  ok = ok && add_chunks(8, true);                           // feature
  ok = ok && add(2, 4, true) && add(2, 5, false);           // views
  ok = ok && add(1, 4, true);                               // rgb
  if (!ok) return -1;
  const int fl = ((TAIL_FLOATS + 255) / 256) * 256;  // f32 tail: [w_qs;w_ks;w_vs | fc | out_alpha.0 | out_alpha.2]
  sch->seg_off[n] = (int)off;
  sch->seg_floats[n] = fl;
  sch->seg_steps[n] = 0;
  off += fl;
  ++n;
  finish_schedule(sch, n, tf, te);
  return (int)off;
}

// f32 format: a stage is cut into ceil(T/cap) segments, the first ones get floor(T/nseg) steps, the last one the
// rest; every segment is padded to a multiple of 256 floats.
static int build_schedule(const mnerf_decoder* D, DecSched* sch) {
  if (D->wstream_format == MNERF_WSTREAM_BF16X3) return build_schedule_split(D, sch, 3);
  if (D->wstream_format == MNERF_WSTREAM_F16X2 || D->wstream_format == MNERF_WSTREAM_F16X1) return build_schedule_split(D, sch, 2);
  const int fs = D->cond_stride / 2, es = 3 * D->L_3D + 2;
  // film, l0, l1..l4, l5-enc, l5-h, feature, views, rgb, alpha (+ the resident tail segment)
  const int T[12] = {fs, es, 65, 65, 65, 65, es, 64, 65, 66, 33, 65};
  const int M[12] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 2, 1, 1};
  int n = 0;
  long long off = 0;
  for (int st = 0; st < 12; ++st) {
    const int cap = SEG_CAP_FLOATS / (64 * M[st]);
    const int nseg = (T[st] + cap - 1) / cap;
    const int base = T[st] / nseg;
    for (int k = 0; k < nseg; ++k) {
      if (n >= MAX_SEGS) return -1;
      const int steps = (k == nseg - 1) ? (T[st] - base * (nseg - 1)) : base;
      const int fl = ((steps * 64 * M[st] + 255) / 256) * 256;
      if (fl > SEG_CAP_FLOATS) return -1;
      sch->seg_off[n] = (int)off;
      sch->seg_floats[n] = fl;
      sch->seg_steps[n] = steps;
      off += fl;
      ++n;
    }
  }
  {  // tail: [w_qs;w_ks;w_vs | fc | out_alpha.0 | out_alpha.2]
    if (n >= MAX_SEGS) return -1;
    const int fl = ((TAIL_FLOATS + 255) / 256) * 256;
    sch->seg_off[n] = (int)off;
    sch->seg_floats[n] = fl;
    sch->seg_steps[n] = 0;
    off += fl;
    ++n;
  }
  finish_schedule(sch, n, fs, es);
  return (int)off;
}

static bool known_format(int f) {
  return f == MNERF_WSTREAM_F32 || f == MNERF_WSTREAM_BF16X3 || f == MNERF_WSTREAM_F16X2 || f == MNERF_WSTREAM_F16X1;
}

#if MNERF_DECODER_PART == 0
extern "C" int64_t mnerf_decoder_wstream_floats(int32_t cond_dim, int32_t cond_stride, int32_t L_3D,
                                                int32_t wstream_format) {
  if (!known_format(wstream_format)) return -1;
  if (L_3D < 0 || L_3D > 16 || cond_dim < 1 || cond_stride < cond_dim) return -1;
  mnerf_decoder d = {};
  d.cond_dim = cond_dim;
  d.cond_stride = cond_stride;
  d.L_3D = L_3D;
  d.wstream_format = wstream_format;
  DecSched s;
  return build_schedule(&d, &s);
}

#endif  // MNERF_DECODER_PART == 0

#if MNERF_DECODER_PART == 0
// Does the ping-pong form (decoder_pp_kernel) take this decoder at Sp padded samples?  -> its FiLM K16-step count (2 .. 4), or 0.
static int pp_film_steps(const mnerf_decoder* dec, const DecSched& sch, int Sp) {
  const int pp_max_s = mnerf_tune().decoder_pp_max_s < 256 ? mnerf_tune().decoder_pp_max_s : 256;
  // FiLM stages of 2 K16-steps (<= 5 views) at every S <= 256 (round 5: the 256-sample instance, one ray per tile across both teams);
  // of 3 / 4 steps (6 .. 11 views) for S <= 64 (the instances that exist)
  const int fs = sch.film_steps;
  const bool f16 = dec->wstream_format == MNERF_WSTREAM_F16X2 || dec->wstream_format == MNERF_WSTREAM_F16X1;
  const bool ok = f16 && Sp > 0 && Sp <= pp_max_s && dec->L_3D == 10 &&
                  (fs == 2 || ((fs == 3 || fs == 4) && Sp == 64)) && dec->cond_stride <= 16 * fs && sch.enc_steps == 4 &&
                  sch.n_seg == 20 && mnerf_tune().decoder_pp;
  return ok ? fs : 0;
}
#endif

static int pick_padded_samples(int S) {
  if (S <= 32) return 32;
  if (S <= 64) return 64;
  if (S <= 128) return 128;
  if (S <= 256) return 256;
  return -1;
}

#ifdef MNERF_FUSED_DEBUG
#define MNERF_FUSED_DEBUG_ON 1
#else
#define MNERF_FUSED_DEBUG_ON 0
#endif
// dynamic LDS of the one-launch form (see CVF_LDS_BYTES); the debug build can ask for the co-resident footprint
static size_t mnerf_fused_lds_bytes(size_t natural) {
#ifdef MNERF_FUSED_DEBUG
  if (const char* e = getenv("MNERF_FDBG_LDS_KB")) {
    const size_t kb = (size_t)atoi(e);
    return kb == 0 ? natural : kb * 1024;
  }
#endif
  return CVF_LDS_BYTES ? (size_t)CVF_LDS_BYTES : natural;
}

// Shared by mnerf_decoder_chunk (rays rebuilt in-kernel, composited outputs) and mnerf_decoder_samples
// (caller-supplied sample coordinates / directions, per-sample outputs only).
static int launch_decoder(const char* who, const mnerf_decoder* dec, const mnerf_view* view0, const mnerf_rays* rays,
                          const float* cond, float* rgb, float* depth, float* opacity, float* rgb_s, float* sigma,
                          const float* ext_ndc, const float* ext_dir, const mnerf_scene* fused_scene, void* stream) {
  MNERF_REQUIRE(dec->wstream && dec->small_ && (cond || fused_scene), MNERF_E_NULL, "%s: NULL buffer", who);
  MNERF_REQUIRE(mnerf_aligned16(dec->wstream) && mnerf_aligned16(cond), MNERF_E_ALIGN,
                "%s: wstream / cond must be 16-byte aligned", who);
  MNERF_REQUIRE(dec->L_3D >= 0 && dec->L_3D <= 16, MNERF_E_RANGE, "%s: L_3D=%d", who, dec->L_3D);
  MNERF_REQUIRE(known_format(dec->wstream_format), MNERF_E_UNSUPPORTED, "%s: wstream_format=%d", who,
                dec->wstream_format);
  const int cs_max = dec->wstream_format == MNERF_WSTREAM_F32 ? MNERF_COND_STRIDE_MAX_F32 : MNERF_COND_STRIDE_MAX;
  MNERF_REQUIRE(dec->cond_stride % 8 == 0 && dec->cond_stride >= dec->cond_dim + 1 && dec->cond_stride <= cs_max,
                MNERF_E_RANGE, "%s: cond_stride=%d (cond_dim=%d) must be a multiple of 8 in (cond_dim, %d]", who,
                dec->cond_stride, dec->cond_dim, cs_max);
  MNERF_REQUIRE(dec->n_views >= 1 && dec->n_views * 4 < dec->cond_dim, MNERF_E_RANGE,
                "%s: n_views=%d inconsistent with cond_dim=%d", who, dec->n_views, dec->cond_dim);
  MNERF_REQUIRE(rays->n_rays >= 0 && rays->n_samples >= 1, MNERF_E_RANGE, "%s: n_rays=%d S=%d", who,
                rays->n_rays, rays->n_samples);
  const int Sp = pick_padded_samples(rays->n_samples);
  MNERF_REQUIRE(Sp > 0, MNERF_E_UNSUPPORTED, "%s: sample_intvs=%d > 256 is not supported by the fused kernel", who,
                rays->n_samples);
  const bool poses = rays->pose_table != nullptr;
  if (poses) {
    MNERF_REQUIRE(rays->rays_per_pose > 0 && rays->rays_per_pose % 64 == 0, MNERF_E_RANGE,
                  "%s: pose table needs rays_per_pose = a positive multiple of 64, got %d", who, rays->rays_per_pose);
    MNERF_REQUIRE(!rays->ray_idx && !rays->strat_u && !ext_ndc && !fused_scene, MNERF_E_UNSUPPORTED,
                  "%s: a pose table excludes ray_idx / strat_u / caller-supplied samples / the one-launch form", who);
  } else {
    MNERF_REQUIRE(rays->rays_per_pose == 0, MNERF_E_RANGE, "%s: rays_per_pose=%d without a pose table", who, rays->rays_per_pose);
  }
  DecSched sch;
  const int total = build_schedule(dec, &sch);
  MNERF_REQUIRE(total > 0, MNERF_E_RANGE, "%s: cannot schedule weight stream", who);
  MNERF_REQUIRE(dec->wstream_floats == total, MNERF_E_RANGE, "%s: wstream has %lld floats, schedule expects %d", who,
                (long long)dec->wstream_floats, total);
  if (rays->n_rays == 0) return MNERF_OK;
  hipStream_t st = (hipStream_t)stream;
  const int resident = mnerf_tune().decoder_grid;  // persistent: 2 workgroups per CU x 256 CUs
  static const mnerf_scene no_scene = {};
  const mnerf_scene* scn = fused_scene ? fused_scene : &no_scene;
#define MNERF_LAUNCH_DECODER(NW_, SP_, FMT_, CVF_)                                                   \
  do {                                                                                               \
    const int rpt = (NW_ * 32) / SP_;                                                                \
    const int tiles = (rays->n_rays + rpt - 1) / rpt;                                                \
    const int grid = tiles < resident ? tiles : resident;                                            \
    size_t lds = Smem<NW_, SP_>::TOTAL_FLOATS * sizeof(float);                                       \
    if (CVF_) lds = mnerf_fused_lds_bytes(lds);                                                      \
    static std::atomic<unsigned long long> attr_set{0};                                              \
    if (mnerf_once_per_device(attr_set) || MNERF_FUSED_DEBUG_ON)                                     \
      (void)hipFuncSetAttribute((const void*)decoder_kernel<NW_, SP_, FMT_, CVF_>,                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
    hipLaunchKernelGGL((decoder_kernel<NW_, SP_, FMT_, CVF_>), dim3(grid), dim3(NW_ * 64), lds, st,  \
                       *dec, sch, *view0, *rays, cond, rgb, depth, opacity, rgb_s, sigma, ext_ndc,   \
                       ext_dir, *scn);                                                               \
  } while (0)
#if MNERF_DECODER_PART == 0
  // ---- ping-pong form (decoder_pp_kernel): the shipped decoder shape on the split-fp16 stream, S <= 256 (round 3: 82.3 vs
  // 88.7 ms per 800x800 frame at 128 samples per ray; round 5: the 256-sample instance; the knob can only LOWER the limit)
  const int fs = fused_scene ? 0 : pp_film_steps(dec, sch, Sp);
  if (fs) {
    PPSched pps;
    const int first[PP_STAGES] = {0, 1, 2, 4, 6, 8, 11, 10, 13, 14, 16, 18};  // layer 5: activation half, then encoding half
    const int nseg[PP_STAGES] = {1, 1, 2, 2, 2, 2, 2, 1, 1, 2, 2, 1};
    for (int i = 0; i < PP_STAGES; ++i) {
      pps.seg_first[i] = first[i], pps.n_seg[i] = nseg[i];
      // the kernel's compile-time piece counts must be the schedule's
      MNERF_REQUIRE((sch.seg_floats[first[i]] >> 8) == pp_p0(i, fs) && (nseg[i] == 2 ? (sch.seg_floats[first[i] + 1] >> 8) : 0) == pp_p1(i),
                    MNERF_E_RANGE, "%s: weight schedule does not match the ping-pong kernel (stage %d)", who, i);
      MNERF_REQUIRE(first[i] == pp_seg_first(i), MNERF_E_RANGE, "%s: stage table of the ping-pong kernel (stage %d)", who, i);
    }
    for (int k = 0; k < 20; ++k)  // the kernel addresses the stream with compile-time offsets
      MNERF_REQUIRE(sch.seg_off[k] == pp_stream_off_floats(k, fs) && (sch.seg_floats[k] >> 8) == pp_stream_pieces(k, fs), MNERF_E_RANGE,
                    "%s: weight stream layout does not match the ping-pong kernel (segment %d)", who, k);
    const int rpt = 256 / Sp;
    const int tiles = (rays->n_rays + rpt - 1) / rpt;
    const int cus = mnerf_tune().decoder_pp_grid > 0 ? mnerf_tune().decoder_pp_grid : 1;  // persistent: one 8-wave workgroup per CU
    const int grid = tiles < cus ? tiles : cus;
#define MNERF_LAUNCH_PP_(SP_, FS_, POSES_)                                                                            \
  do {                                                                                                                \
    const size_t lds = SmemPP<SP_>::TOTAL_FLOATS * sizeof(float);                                                     \
    static std::atomic<unsigned long long> attr_set{0};                                                               \
    if (mnerf_once_per_device(attr_set))                                                                              \
      (void)hipFuncSetAttribute((const void*)decoder_pp_kernel<SP_, FS_, POSES_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((decoder_pp_kernel<SP_, FS_, POSES_>), dim3(grid), dim3(512), lds, st, *dec, sch, pps, *view0, *rays, cond, rgb, depth, \
                       opacity, rgb_s, sigma, ext_ndc, ext_dir);                                                      \
  } while (0)
#define MNERF_LAUNCH_PP(SP_, FS_) MNERF_LAUNCH_PP_(SP_, FS_, false)
    if (dec->wstream_format == MNERF_WSTREAM_F16X1) {  // reduced-precision fast mode: the shipped shape only
      MNERF_REQUIRE(fs == 2 && !poses && Sp <= 128, MNERF_E_UNSUPPORTED,
                    "%s: the one-product fp16 mode (MNERF_WSTREAM_F16X1) is built for <= 5 source views, sample_intvs <= 128, without a "
                    "pose table", who);
#define MNERF_LAUNCH_PP1(SP_)                                                                                          \
  do {                                                                                                                \
    const size_t lds = SmemPP<SP_>::TOTAL_FLOATS * sizeof(float);                                                     \
    static std::atomic<unsigned long long> attr_set{0};                                                               \
    if (mnerf_once_per_device(attr_set))                                                                              \
      (void)hipFuncSetAttribute((const void*)decoder_pp_kernel<SP_, 2, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((decoder_pp_kernel<SP_, 2, false, 1>), dim3(grid), dim3(512), lds, st, *dec, sch, pps, *view0, *rays, cond, rgb, depth, \
                       opacity, rgb_s, sigma, ext_ndc, ext_dir);                                                      \
  } while (0)
      if (Sp == 32)
        MNERF_LAUNCH_PP1(32);
      else if (Sp == 64)
        MNERF_LAUNCH_PP1(64);
      else
        MNERF_LAUNCH_PP1(128);
#undef MNERF_LAUNCH_PP1
    } else if (poses) {  // the instances that exist with a pose table: the shipped 3-view shape (<= 5 views) at S <= 128
      MNERF_REQUIRE(fs == 2 && (Sp == 32 || Sp == 64 || Sp == 128), MNERF_E_UNSUPPORTED,
                    "%s: pose tables are built for <= 5 source views and sample_intvs <= 128", who);
      if (Sp == 32)
        MNERF_LAUNCH_PP_(32, 2, true);
      else if (Sp == 64)
        MNERF_LAUNCH_PP_(64, 2, true);
      else  // round 5: configs/demo_own.yaml renders its video at the reference's default 128 samples per ray
        MNERF_LAUNCH_PP_(128, 2, true);
    } else if (fs == 3)
      MNERF_LAUNCH_PP(64, 3);
    else if (fs == 4)
      MNERF_LAUNCH_PP(64, 4);
    else if (Sp == 32)
      MNERF_LAUNCH_PP(32, 2);
    else if (Sp == 64)
      MNERF_LAUNCH_PP(64, 2);
    else if (Sp == 128)
      MNERF_LAUNCH_PP(128, 2);
    else  // 128 < S <= 256 (configs/test_video_own.yaml: sample_intvs 256)
      MNERF_LAUNCH_PP(256, 2);
#undef MNERF_LAUNCH_PP
#undef MNERF_LAUNCH_PP_
    return mnerf_check_launch(who);
  }
#endif
  MNERF_REQUIRE(dec->wstream_format != MNERF_WSTREAM_F16X1, MNERF_E_UNSUPPORTED,
                "%s: the one-product fp16 mode (MNERF_WSTREAM_F16X1) exists in the ping-pong decoder only (<= 5 source views, "
                "sample_intvs <= 128, MNERF_DECODER_PP on, not the one-launch form)", who);
  MNERF_REQUIRE(!poses, MNERF_E_UNSUPPORTED, "%s: a pose table needs the ping-pong decoder (split-fp16 stream, <= 5 source views, "
                "sample_intvs <= 128, MNERF_DECODER_PP on)", who);
#if MNERF_DECODER_PART == 1
  MNERF_REQUIRE(fused_scene, MNERF_E_NULL, "%s: the one-launch form needs the scene", who);
#define MNERF_LAUNCH_DECODER_FMT(NW_, SP_) MNERF_LAUNCH_DECODER(NW_, SP_, 2, 1)
#else
  MNERF_REQUIRE(!fused_scene, MNERF_E_UNSUPPORTED, "%s: the one-launch form lives in decoder_fused.o", who);
#define MNERF_LAUNCH_DECODER_FMT(NW_, SP_)                                   \
  do {                                                                       \
    if (dec->wstream_format == MNERF_WSTREAM_F16X2)                          \
      MNERF_LAUNCH_DECODER(NW_, SP_, 2, 0);                                  \
    else if (dec->wstream_format == MNERF_WSTREAM_BF16X3)                    \
      MNERF_LAUNCH_DECODER(NW_, SP_, 1, 0);                                  \
    else                                                                     \
      MNERF_LAUNCH_DECODER(NW_, SP_, 0, 0);                                  \
  } while (0)
#endif
  switch (Sp) {
    case 32: MNERF_LAUNCH_DECODER_FMT(4, 32); break;
    case 64: MNERF_LAUNCH_DECODER_FMT(4, 64); break;
    case 128: MNERF_LAUNCH_DECODER_FMT(4, 128); break;
    default:  // 128 < S <= 256 when the ping-pong form does not apply (other stream formats, 6+ views, MNERF_DECODER_PP_MAX_S < 256):
              // one 8-wave workgroup per CU, VALU ray attention
#if MNERF_DECODER_PART == 1
      MNERF_REQUIRE(false, MNERF_E_UNSUPPORTED, "%s: the one-launch form needs sample_intvs <= 128", who);
#else
      if (dec->wstream_format == MNERF_WSTREAM_F16X2)
        MNERF_LAUNCH_DECODER(8, 256, 2, 0);
      else if (dec->wstream_format == MNERF_WSTREAM_BF16X3)
        MNERF_LAUNCH_DECODER(8, 256, 1, 0);
      else
        MNERF_LAUNCH_DECODER(8, 256, 0, 0);
#endif
      break;
  }
#undef MNERF_LAUNCH_DECODER_FMT
#undef MNERF_LAUNCH_DECODER
  return mnerf_check_launch(who);
}

#if MNERF_DECODER_PART == 1
// The fused ray-chunk form (one launch, no workspace) exists for the shipped configuration class: split-fp16
// stream, S <= 128, at most 32 conditioning inputs (<= 5 views), cosine groups of at most 8 lanes (G >= 2) and
// walk scratch that fits one weight buffer.
bool mnerf_fused_render_applies(const mnerf_scene* sc, const mnerf_decoder* dec, const mnerf_rays* rays) {
  if (dec->wstream_format != MNERF_WSTREAM_F16X2 || rays->n_samples > 128) return false;
  if (rays->pose_table) return false;  // pose tables: two-launch form only
  if (dec->cond_stride > 32 || (dec->cond_dim + 15) / 16 > 2) return false;
  if (sc->n_views < 2 || sc->n_views != dec->n_views) return false;
  int sumG = 0;
  for (int s = 0; s < sc->n_scales; ++s) {
    if (sc->n_group[s] < 2) return false;
    sumG += sc->n_group[s];
  }
  if (sumG > 16) return false;
  return 16 * cv_slot_lds_floats(CVF_SEG, sc->n_views, sumG) <= SEG_CAP_FLOATS &&
         CVF_COND_OFF_FLOATS + 128 * dec->cond_stride <= SEG_CAP_FLOATS;
}

int mnerf_fused_render_launch(const mnerf_scene* sc, const mnerf_decoder* dec, const mnerf_rays* rays, float* rgb,
                              float* depth, float* opacity, void* stream) {
  float *rgb_s = nullptr, *sigma = nullptr;
#ifdef MNERF_FUSED_DEBUG
  if (const char* e = getenv("MNERF_FDBG_RGBS")) rgb_s = (float*)strtoull(e, nullptr, 0);
  if (const char* e = getenv("MNERF_FDBG_SIGMA")) sigma = (float*)strtoull(e, nullptr, 0);
#endif
  return launch_decoder("mnerf_render_chunk", dec, &sc->views[0], rays, nullptr, rgb, depth, opacity, rgb_s, sigma,
                        nullptr, nullptr, sc, stream);
}

#else  // MNERF_DECODER_PART == 0
// the pose-table instances of the decoder: the ping-pong form with a 2-step FiLM stage (<= 5 source views) at S <= 64
bool mnerf_decoder_takes_pose_table(const mnerf_decoder* dec, int n_samples) {
  if (!known_format(dec->wstream_format) || dec->wstream_format == MNERF_WSTREAM_F16X1 || dec->L_3D < 0 || dec->L_3D > 16) return false;
  DecSched sch;
  if (build_schedule(dec, &sch) <= 0) return false;
  const int Sp = pick_padded_samples(n_samples);
  return (Sp == 32 || Sp == 64 || Sp == 128) && pp_film_steps(dec, sch, Sp) == 2;
}

extern "C" int mnerf_decoder_chunk(const mnerf_decoder* dec, const mnerf_view* view0,
                                   const mnerf_rays* rays, const float* cond, float* rgb,
                                   float* depth, float* opacity, float* dbg_rgb_s,
                                   float* dbg_sigma, void* stream) {
  MNERF_REQUIRE(dec && view0 && rays, MNERF_E_NULL, "mnerf_decoder_chunk: NULL argument struct");
  MNERF_REQUIRE(rgb && depth && opacity && cond, MNERF_E_NULL, "mnerf_decoder_chunk: NULL buffer");
  MNERF_REQUIRE(rays->legacy_coord == 0 || rays->n_samples >= 2, MNERF_E_RANGE,
                "mnerf_decoder_chunk: legacy depth sampling needs S >= 2");
  return launch_decoder("mnerf_decoder_chunk", dec, view0, rays, cond, rgb, depth, opacity, dbg_rgb_s, dbg_sigma,
                        nullptr, nullptr, nullptr, stream);
}

extern "C" int mnerf_decoder_samples(const mnerf_decoder* dec, int32_t n_rays, int32_t n_samples,
                                     int32_t legacy_coord, const float* x_ndc, const float* dir,
                                     const float* cond, float* rgb_s, float* sigma, void* stream) {
  MNERF_REQUIRE(dec, MNERF_E_NULL, "mnerf_decoder_samples: dec is NULL");
  MNERF_REQUIRE(n_rays >= 0 && n_samples >= 1, MNERF_E_RANGE, "mnerf_decoder_samples: n_rays=%d S=%d", n_rays, n_samples);
  if (n_rays == 0) return MNERF_OK;  // empty chunk: nothing to read or write
  MNERF_REQUIRE(x_ndc && dir && rgb_s && sigma, MNERF_E_NULL, "mnerf_decoder_samples: NULL buffer");
  mnerf_rays rays = {};
  rays.n_rays = n_rays;
  rays.n_samples = n_samples;
  rays.height = rays.width = 2;
  rays.legacy_coord = legacy_coord ? 1 : 0;  // here it only selects the positional-encoding frequency factor (1 | pi)
  const mnerf_view none = {};
  MNERF_REQUIRE(cond, MNERF_E_NULL, "mnerf_decoder_samples: cond is NULL");
  return launch_decoder("mnerf_decoder_samples", dec, &none, &rays, cond, nullptr, nullptr, nullptr, rgb_s, sigma,
                        x_ndc, dir, nullptr, stream);
}
#endif  // MNERF_DECODER_PART
