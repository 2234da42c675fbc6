// Split-fp16 MFMA building blocks (shared by the fused decoder, decoder.hip, and the encoder block kernel,
// encoder_block.hip): fp32-grade matrix products on v_mfma_f32_32x32x16_f16 from two range-managed fp16 terms per
// operand, the LDS-DMA weight-segment copy, and the LDS operand types.
#pragma once
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
// LDS operands are addressed by LDS byte offset through explicit address_space(3) pointers made from integers
// (no generic->LDS pointer casts near the hot loops: they trip a gfx950 code-generation bug in hipcc 7.2)
typedef const u32x4 __attribute__((address_space(3)))* lds_u32x4_cptr;
typedef const v4f32 __attribute__((address_space(3)))* lds_v4f32_cptr;

// ---- weight-segment DMA: each wave copies 1 KiB pieces, LDS dest = uniform base + lane*16.
// The LDS-DMA is issued from inline asm on purpose: hipcc tracks a builtin global_load_lds as a
// pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the NEXT ds_read — which would
// drain the prefetch of segment i+1 before segment i has issued a single MFMA (seen in the
// ISA: the DMA was fully exposed 18x per tile).  An asm load is invisible to that bookkeeping
// (cdna_hip_programming.md 5.7); segment_wait() below is the one place that waits for it,
// right before the barrier that publishes the buffer.  M0 (LDS base) is written and restored
// inside the same statement.
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte_addr)
      : "memory");
}

// glds16 for FOUR consecutive pieces (see glds16_sv4 below: one M0 write, the instruction offset moves both addresses)
__device__ __forceinline__ void glds16x4(const float* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte_addr)
      : "memory");
}
// A contiguous segment of `pieces` 1-KiB pieces copied by the `nw` waves of a workgroup: runs of four dealt round-robin (one M0
// write each), the remainder piece by piece.  src_lane = segment + lane * 4 floats; wave / pieces wave-uniform.
__device__ __forceinline__ void glds_segment(const float* src_lane, unsigned lds_byte_addr, int pieces, int wave, int nw) {
  const int n_runs = pieces >> 2;
  for (int r = wave; r < n_runs; r += nw) glds16x4(src_lane + r * 1024, __builtin_amdgcn_readfirstlane(lds_byte_addr + (unsigned)r * 4096u));
  for (int p = 4 * n_runs + wave; p < pieces; p += nw) glds16(src_lane + p * 256, __builtin_amdgcn_readfirstlane(lds_byte_addr + (unsigned)p * 1024u));
}

// The same copy with a wave-uniform source base in a scalar register pair and the per-lane byte offset (lane * 16) in ONE
// vector register that never changes: no per-piece vector address arithmetic.  (s_nop 4: a base that came through
// v_readfirstlane is a VALU-written SGPR read by a memory instruction.)
__device__ __forceinline__ void glds16_s(const float* uniform_src, unsigned lane_off_bytes, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_off_bytes), "s"(uniform_src), "s"(lds_byte_addr)
      : "memory");
}

// The same copy with ONE wave-uniform base for a whole stream and the piece's byte offset added to the per-lane offset in the
// vector ALU (one v_add_u32 per request, offset in a scalar register or a literal): the ping-pong decoder's ~200 distinct
// `stream + constant` source pointers are loop-invariant 64-bit values that the compiler hoists out of the tile loop and parks
// in VGPR lanes (two v_readlane per request, ~120 spilled SGPRs); a 32-bit offset computed at the request is one scalar add.
__device__ __forceinline__ void glds16_sv(const float* uniform_base, unsigned lane_off_bytes, unsigned piece_off_bytes,
                                          unsigned lds_byte_addr) {
  unsigned keep, vo;
  asm volatile(
      "v_add_u32 %1, %4, %2\n\t"
      "s_nop 3\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "=&v"(vo)
      : "v"(lane_off_bytes), "s"(uniform_base), "s"(piece_off_bytes), "s"(lds_byte_addr)
      : "memory");
}

// FOUR consecutive 1-KiB pieces per M0 write: the instruction offset (0, 1024, 2048, 3072) moves the global AND the LDS address
// of an LDS-DMA load, so a contiguous 4-KiB run needs one address add, one M0 save / write / restore and one set of hazard nops.
// tools/exp/ubench/dma_issue.hip (8 waves per CU, requests next to a vector filler, cycles of issue per request):
// one piece per statement (glds16_s) 74-78, the same without its leading s_nop 4: 50-55, four per M0 write: 40-43.
__device__ __forceinline__ void glds16_sv4(const float* uniform_base, unsigned lane_off_bytes, unsigned piece_off_bytes,
                                           unsigned lds_byte_addr) {
  unsigned keep, vo;
  asm volatile(
      "v_add_u32 %1, %4, %2\n\t"
      "s_nop 3\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "=&v"(vo)
      : "v"(lane_off_bytes), "s"(uniform_base), "s"(piece_off_bytes), "s"(lds_byte_addr)
      : "memory");
}

// glds16_s for data read exactly once (the conditioning rows): non-temporal, so that the stream does not push the weight
// segments and the register-spill scratch out of L2 (ping-pong decoder, MI355X: FETCH_SIZE 462 -> 212 MB per 65 536-ray launch,
// L2 misses 40 M -> 17 M, 18.27 -> 17.93 ms per frame; -DMNERF_ROWS_TEMPORAL restores the plain form).
__device__ __forceinline__ void glds16_s_stream(const float* uniform_src, unsigned lane_off_bytes, unsigned lds_byte_addr) {
#ifndef MNERF_ROWS_TEMPORAL
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_off_bytes), "s"(uniform_src), "s"(lds_byte_addr)
      : "memory");
#else
  glds16_s(uniform_src, lane_off_bytes, lds_byte_addr);
#endif
}

__device__ __forceinline__ void segment_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- a barrier for a SUBSET of the workgroup's waves (gfx950 has one hardware barrier per workgroup and no named barriers):
// a monotonic counter in LDS.  Every wave of the group (n_waves of them) calls it the same number of times; `epoch` is the
// wave's own count of arrivals so far (kept in a scalar register by the caller, starts at 0, counter zeroed before first use).
//   arrive: all LDS traffic of this wave has completed (lgkmcnt(0): LDS executes a wave's operations in order, so its writes are
//           visible and its reads have their data), then lane 0 adds 1;
//   wait:   poll the counter (one ds_read_b32 + s_sleep per round: the SIMD's other wave keeps the issue slots) until it has
//           reached n_waves x calls.
// Outstanding VMEM / LDS-DMA traffic is NOT waited for (the ping-pong decoder keeps weight and row DMAs in flight across these
// syncs; data that arrives by DMA is published by the issuing wave's own segment_wait() before it arrives here).
// Everything is asm volatile with a memory clobber: the compiler moves no LDS access across it.
// The lane index straight from the hardware (v_mbcnt), behind a volatile asm: a value the compiler can neither hoist out of a loop nor
// keep alive across it.  The ping-pong decoder used `tid & 63` from the kernel's first lines at every weight request and every team
// sync; live across all 28 phases it was parked in scratch memory, and each vector phase began with a scratch reload + s_waitcnt
// vmcnt(0) (a few hundred exposed cycles, twelve times per tile) in front of its first request.
__device__ __forceinline__ unsigned hw_lane() {
  unsigned v;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(v));
  return v;
}
__device__ __forceinline__ void wave_group_sync(unsigned ctr_lds_byte_addr, int n_waves, int& epoch, int lane) {
  epoch += n_waves;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) {
    const unsigned one = 1u;
    asm volatile("ds_add_u32 %0, %1" ::"v"(ctr_lds_byte_addr), "v"(one) : "memory");
  }
  for (;;) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ctr_lds_byte_addr) : "memory");
    if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - (unsigned)epoch) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

// ---------------------------------------------------------------- split-fp16 matrix path ("f16x3", FMT = 2)
// Same chain on v_mfma_f32_32x32x16_f16 with TWO fp16 terms per operand and THREE products per MAC
// (hi.hi + hi.lo + lo.hi; the dropped lo.lo term is < 2^-22 of the product): half the matrix instructions of
// bf16x6 and about half its operand-split VALU work (gfx950 has v_cvt_pk_f16_f32; the residual is one
// v_pk_fma_f32 per pair).  fp16 has 11 significand bits but a 5-bit exponent, so both operands are range-managed
// with exact power-of-two scales (cond_nerf.py: pack_wstream_h): weights carry one scale 2^ew per tensor (host),
// activations one gain per SAMPLE and stage, taken from the running maximum of the sample's features (in-lane
// max + one cross-half shuffle) so that the largest operand lands in [2^14, 2^15).  The accumulator then holds
// 2^(ew+eg) (W x); scales are tracked as integer exponents per lane and multiplied back in exactly.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define H16_UNIT_BYTES 2048  // one (step, block): hi | lo fragments
#define H16_TARGET_EXP 15    // largest operand of a sample is scaled into [2^14, 2^15)

struct PartsH {
  f16x8 hi, lo;
};

// residual v * mult - half(hpk) in one VALU op: v_fma_mix_f32 reads the fp16 operand straight out of the packed
// register (no v_cvt_f32_f16), fp32 FMA, one rounding (the product is exact: mult is a power of two)
__device__ __forceinline__ float resid_lo(float v, float mult, unsigned hpk) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(mult), "v"(hpk));
  return r;
}
__device__ __forceinline__ float resid_hi(float v, float mult, unsigned hpk) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(mult), "v"(hpk));
  return r;
}

// v * mult (mult a power of two) -> fp16 hi (RNE) + fp16 lo of the exact fp32 residual: 6 VALU ops per pair of values
// (2 x v_mul_f32, v_cvt_pk_f16_f32, 2 x v_fma_mix_f32, v_cvt_pk_f16_f32; 5 with v_pk_mul_f32 before packed fp32 was banned from
// the library, build.py: NO_PACKED_F32).
// -DMNERF_SPLIT_MIX: the same values (bit for bit, measured) from FOUR "mix" FMAs per pair that write one fp16 half each,
//   hi.lo16 = f16(v0 * mult + 0)  v_fma_mixlo_f16      lo.lo16 = f16(v0 * mult - hi.lo)  v_fma_mixlo_f16
//   hi.hi16 = f16(v1 * mult + 0)  v_fma_mixhi_f16      lo.hi16 = f16(v1 * mult - hi.hi)  v_fma_mixhi_f16
// 1 000 fewer vector instructions per decoder tile - and SLOWER on MI355X: 18.99 vs 18.51 ms per frame (the half-writing forms
// are read-modify-write on their destination and evidently cost more than one issue slot).  Kept as a build-time experiment.
__device__ __forceinline__ PartsH split8h(const float (&v)[8], float mult) {
  u32x4 H, L;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#ifndef MNERF_SPLIT_MIX
    const f32x2 ab = {v[2 * i] * mult, v[2 * i + 1] * mult};
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, f16x2));  // v_cvt_pk_f16_f32
    const f32x2 r = {resid_lo(v[2 * i], mult, h), resid_hi(v[2 * i + 1], mult, h)};
    H[i] = h;
    L[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
#else
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(v[2 * i]), "v"(mult));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(v[2 * i + 1]), "v"(mult));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v[2 * i]), "v"(mult), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v[2 * i + 1]), "v"(mult), "v"(h));
    H[i] = h;
    L[i] = l;
#endif
  }
  PartsH p;
  p.hi = __builtin_bit_cast(f16x8, H);
  p.lo = __builtin_bit_cast(f16x8, L);
  return p;
}

__device__ __forceinline__ f32x16 mfma16h(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// Diagnostic build (-DMNERF_ONE_PRODUCT, tools/exp/one_product.sh; never shipped): the two cross products hi.lo and lo.hi are
// skipped, i.e. plain fp16 operands with fp32 accumulation — how much of the kernel's time is the 3x split tax?
#ifdef MNERF_ONE_PRODUCT
#define MFMA16H_CROSS(a_, b_, c_) (c_)
#else
#define MFMA16H_CROSS(a_, b_, c_) mfma16h(a_, b_, c_)
#endif

// exponent em with 2^em * m in [2^14, 2^15) (m > 0; clamped so that every scale stays a normal fp32 number)
__device__ __forceinline__ int gain_exp(float m) {
  int e = __builtin_amdgcn_frexp_expf(m);  // m = f 2^e, f in [0.5, 1); 0 for m = 0 / inf / nan
  // (register-unit magnitudes sit around 2^57: accumulator 2^29 x un-normalised FiLM 2^28; the clamp only keeps
  // 2^(15-e) a normal fp32 number for denormal / near-overflow inputs)
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return H16_TARGET_EXP - e;
}
__device__ __forceinline__ float pow2i(int e) { return ldexpf(1.0f, e); }

typedef const float __attribute__((address_space(3)))* lds_f32_cptr;
// stage header (1 KiB): floats [0,128) bias in accumulator order, [128] 2^-ew, [129] (float)ew
__device__ __forceinline__ int header_ew(unsigned frag_lds) {
  return __builtin_amdgcn_readfirstlane((int)((lds_f32_cptr)(size_t)frag_lds)[129]);
}

// accumulators <- bias * bmult (bmult = 2^(ew + operand-gain exponent): the accumulator's scale)
template <int NMB>
__device__ __forceinline__ void bias_init_h(f32x16 (&acc)[NMB], unsigned frag_lds, int hl, float bmult) {
  lds_v4f32_cptr p = (lds_v4f32_cptr)(size_t)frag_lds + hl * 16;
#pragma unroll
  for (int m = 0; m < NMB; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f32 t = p[m * 4 + q];
      acc[m][4 * q] = t.x * bmult;
      acc[m][4 * q + 1] = t.y * bmult;
      acc[m][4 * q + 2] = t.z * bmult;
      acc[m][4 * q + 3] = t.w * bmult;
    }
}

// NS K16-steps against NMB output blocks; v: the lane's 8*NS operand values, mult: their power-of-two gain.
// A fragments of unit (step, block) i+1 are read before the three MFMAs of unit i.
template <int NMB, int NS>
__device__ __forceinline__ void ksteps_h(f32x16 (&acc)[NMB], unsigned base_lds, int lane, const float (&v)[8 * NS],
                                         float mult) {
  lds_u32x4_cptr a = (lds_u32x4_cptr)(size_t)base_lds + lane;
  u32x4 ch = a[0], cl = a[64];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    float vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vv[j] = v[8 * u + j];
    const PartsH b = split8h(vv, mult);
#pragma unroll
    for (int m = 0; m < NMB; ++m) {
      const int i = u * NMB + m;
      const int nx = (i + 1 < NS * NMB) ? (i + 1) * 128 : i * 128;  // the last unit re-reads itself
      const u32x4 nh = a[nx], nl = a[nx + 64];
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(ch), "+v"(cl));  // ONE wait for both fragments, in front of the three dependent MFMAs (see ksteps_presplit2)
      const f16x8 ah = __builtin_bit_cast(f16x8, ch), al = __builtin_bit_cast(f16x8, cl);
      acc[m] = MFMA16H_CROSS(ah, b.lo, acc[m]);
      acc[m] = MFMA16H_CROSS(al, b.hi, acc[m]);
      acc[m] = mfma16h(ah, b.hi, acc[m]);
      __builtin_amdgcn_sched_barrier(0);
      ch = nh;
      cl = nl;
    }
  }
}

// NS K16-steps against NMB output blocks with operands that are ALREADY split (mlp.0's operands are the same for
// all eight hidden chunks: split once, 128 registers that the matrix instructions can read from the AGPR half)
template <int NMB, int NS>
__device__ __forceinline__ void ksteps_presplit(f32x16 (&acc)[NMB], unsigned base_lds, int lane, const PartsH* b) {
  lds_u32x4_cptr a = (lds_u32x4_cptr)(size_t)base_lds + lane;
  u32x4 ch = a[0], cl = a[64];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
#pragma unroll
    for (int m = 0; m < NMB; ++m) {
      const int i = u * NMB + m;
      const int nx = (i + 1 < NS * NMB) ? (i + 1) * 128 : i * 128;
      const u32x4 nh = a[nx], nl = a[nx + 64];
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(ch), "+v"(cl));
      const f16x8 ah = __builtin_bit_cast(f16x8, ch), al = __builtin_bit_cast(f16x8, cl);
      acc[m] = MFMA16H_CROSS(ah, b[u].lo, acc[m]);
      acc[m] = MFMA16H_CROSS(al, b[u].hi, acc[m]);
      acc[m] = mfma16h(ah, b[u].hi, acc[m]);
      __builtin_amdgcn_sched_barrier(0);
      ch = nh;
      cl = nl;
    }
  }
}

// The same products in the same order with the A fragments requested TWO units ahead of their matrix instructions and
// across a segment boundary (units [0, NS0 NMB) at base0, the rest at base1): with one unit of look-ahead (96 cycles of
// matrix work) every unit waited for its ds_read_b128 pair (measured: 4.4-5.3 k cycles for the 96 instructions of a layer
// against 3.1 k of issue time).
#ifndef MNERF_PP_DEPTH
#define MNERF_PP_DEPTH 2  // units of look-ahead of the fragment reads (each unit in flight holds 8 registers)
#endif
#ifndef MNERF_PP_GROUP2
#define MNERF_PP_GROUP2 0  // experiment: the fragments of TWO units requested together (4 ds_read_b128 every second unit): half as
#endif                     // many interruptions of the matrix-instruction stream, each twice as long
#if MNERF_PP_GROUP2
template <int NMB, int NS0, int NS1>
__device__ __forceinline__ void ksteps_presplit2(f32x16 (&acc)[NMB], unsigned base0_lds, unsigned base1_lds, int lane,
                                                 const PartsH* b) {
  constexpr int N0 = NS0 * NMB, N = (NS0 + NS1) * NMB, NP = (N + 1) / 2;
  lds_u32x4_cptr a0 = (lds_u32x4_cptr)(size_t)base0_lds + lane;
  lds_u32x4_cptr a1 = (lds_u32x4_cptr)(size_t)base1_lds + lane;
  u32x4 fh[2][2], fl[2][2];  // [pair buffer][unit of the pair]
  auto fetch = [&](int pr, int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = 2 * pr + q;
      if (i >= N) break;
      if (i < N0) {
        fh[buf][q] = a0[i * 128];
        fl[buf][q] = a0[i * 128 + 64];
      } else {
        fh[buf][q] = a1[(i - N0) * 128];
        fl[buf][q] = a1[(i - N0) * 128 + 64];
      }
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    if (pr + 1 < NP) fetch(pr + 1, (pr + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    const int bf = pr & 1;
    asm volatile("" : "+v"(fh[bf][0]), "+v"(fl[bf][0]), "+v"(fh[bf][1]), "+v"(fl[bf][1]));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = 2 * pr + q;
      if (i >= N) break;
      const int u = i / NMB, m = i % NMB;
      const f16x8 ah = __builtin_bit_cast(f16x8, fh[bf][q]), al = __builtin_bit_cast(f16x8, fl[bf][q]);
      acc[m] = MFMA16H_CROSS(ah, b[u].lo, acc[m]);
      acc[m] = MFMA16H_CROSS(al, b[u].hi, acc[m]);
      acc[m] = mfma16h(ah, b[u].hi, acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
#else
// `hook(i, lane_addr, lane_addr_base)` runs behind the three matrix instructions of unit i (i is a constant after unrolling): the
// place where a wave can issue something else — a weight request — at the lowest cost (tools/exp/ubench/mfma_issue.hip: ~50
// cycles per LDS-DMA request among matrix instructions against ~95 in a vector phase).  lane_addr = lane_addr_base + 16 lane is
// the vector register this function reads its fragments through: a request can use it as its per-lane offset (with the
// wave-uniform base lowered by lane_addr_base) instead of claiming a register of its own in the fullest phase of the kernel.
struct KstepsNoHook {
  __device__ __forceinline__ void operator()(int, unsigned, unsigned) const {}
};
// NP: products per MAC.  3 = hi.lo + lo.hi + hi.hi (fp32-grade, the parity path); 1 = hi.hi only — plain fp16 operands with
// fp32 accumulation, the reduced-precision fast mode (MNERF_WSTREAM_F16X1): the lo fragments are then never read and the
// operands' lo halves never used, so the compiler drops their LDS reads and their split arithmetic with them.
template <int NMB, int NS0, int NS1, int NP = 3, class Hook = KstepsNoHook>
__device__ __forceinline__ void ksteps_presplit2(f32x16 (&acc)[NMB], unsigned base0_lds, unsigned base1_lds, int lane,
                                                 const PartsH* b, Hook hook = Hook()) {
  constexpr int N0 = NS0 * NMB, N = (NS0 + NS1) * NMB, DEP = MNERF_PP_DEPTH, NB = DEP + 1;
  lds_u32x4_cptr a0 = (lds_u32x4_cptr)(size_t)base0_lds + lane;
  lds_u32x4_cptr a1 = (lds_u32x4_cptr)(size_t)base1_lds + lane;
  u32x4 fh[NB], fl[NB];
  auto fetch = [&](int i, int slot) {
    if (i < N0) {
      fh[slot] = a0[i * 128];
      if constexpr (NP == 3) fl[slot] = a0[i * 128 + 64];
    } else {
      fh[slot] = a1[(i - N0) * 128];
      if constexpr (NP == 3) fl[slot] = a1[(i - N0) * 128 + 64];
    }
  };
#pragma unroll
  for (int i = 0; i < DEP; ++i)
    if (i < N) fetch(i, i);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i + DEP < N) fetch(i + DEP, (i + DEP) % NB);
    __builtin_amdgcn_sched_barrier(0);
    const int u = i / NMB, m = i % NMB;
    // both fragments are waited for BEFORE the first of the three dependent matrix instructions: hipcc otherwise puts
    // the s_waitcnt of the lo fragment between the first and the second, and an extra issue slot between two MFMAs on the
    // same accumulator costs ~40 cycles (measured: 3.9 k cycles for the 96 instructions of a layer against 3.1 k)
    if constexpr (NP == 3) asm volatile("" : "+v"(fh[i % NB]), "+v"(fl[i % NB]));
    else asm volatile("" : "+v"(fh[i % NB]));
    const f16x8 ah = __builtin_bit_cast(f16x8, fh[i % NB]), al = __builtin_bit_cast(f16x8, NP == 3 ? fl[i % NB] : fh[i % NB]);
    if constexpr (NP == 3) {
      acc[m] = MFMA16H_CROSS(ah, b[u].lo, acc[m]);
      acc[m] = MFMA16H_CROSS(al, b[u].hi, acc[m]);
    }
    acc[m] = mfma16h(ah, b[u].hi, acc[m]);
    __builtin_amdgcn_sched_barrier(0);
    hook(i, (unsigned)(size_t)a0, base0_lds);
    __builtin_amdgcn_sched_barrier(0);
  }
}
#endif  // MNERF_PP_GROUP2

// The same again with the two output blocks of a pair (m, m+1) interleaved: a0 a1 a0 a1 a0 a1 instead of a0 a0 a0 a1 a1 a1,
// so that no matrix instruction has the accumulator of its predecessor (every accumulator still receives its three products
// in the same order: identical results).  Fragments of a pair are requested one pair (six matrix instructions) ahead.
template <int NMB, int NS0, int NS1>
__device__ __forceinline__ void ksteps_presplit2p(f32x16 (&acc)[NMB], unsigned base0_lds, unsigned base1_lds, int lane,
                                                  const PartsH* b) {
  static_assert(NMB % 2 == 0, "pairs of output blocks");
  constexpr int N0 = NS0 * NMB, N = (NS0 + NS1) * NMB, NP = N / 2;
  lds_u32x4_cptr a0 = (lds_u32x4_cptr)(size_t)base0_lds + lane;
  lds_u32x4_cptr a1 = (lds_u32x4_cptr)(size_t)base1_lds + lane;
  u32x4 fh[2][2], fl[2][2];  // [buffer][unit of the pair]
  auto fetch = [&](int pr, int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = 2 * pr + q;
      if (i < N0) {
        fh[buf][q] = a0[i * 128];
        fl[buf][q] = a0[i * 128 + 64];
      } else {
        fh[buf][q] = a1[(i - N0) * 128];
        fl[buf][q] = a1[(i - N0) * 128 + 64];
      }
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    if (pr + 1 < NP) fetch(pr + 1, (pr + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    const int u = (2 * pr) / NMB, m = (2 * pr) % NMB;
    const int bf = pr & 1;
    const f16x8 ah0 = __builtin_bit_cast(f16x8, fh[bf][0]), al0 = __builtin_bit_cast(f16x8, fl[bf][0]);
    const f16x8 ah1 = __builtin_bit_cast(f16x8, fh[bf][1]), al1 = __builtin_bit_cast(f16x8, fl[bf][1]);
    acc[m] = mfma16h(ah0, b[u].lo, acc[m]);
    acc[m + 1] = mfma16h(ah1, b[u].lo, acc[m + 1]);
    acc[m] = mfma16h(al0, b[u].hi, acc[m]);
    acc[m + 1] = mfma16h(al1, b[u].hi, acc[m + 1]);
    acc[m] = mfma16h(ah0, b[u].hi, acc[m]);
    acc[m + 1] = mfma16h(ah1, b[u].hi, acc[m + 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ksteps_presplit2 with the NEXT stage's weight pieces requested between the matrix instructions (the cheapest place to
// issue an LDS-DMA: ~60 cycles among MFMAs against 150-200 in a VALU phase): this wave takes pieces tw, tw+4, .. of the P0 /
// P1 one-KiB pieces of the next stage's two segments and spreads them evenly over the units of this stage.
template <int NMB, int NS0, int NS1, int P0, int P1>
__device__ __forceinline__ void ksteps_presplit2_dma(f32x16 (&acc)[NMB], unsigned base0_lds, unsigned base1_lds, int lane,
                                                     const PartsH* b, bool issue, int tw, const float* src0,
                                                     const float* src1, unsigned dst0_lds, unsigned dst1_lds) {
  constexpr int N0 = NS0 * NMB, N = (NS0 + NS1) * NMB;
  constexpr int K0 = (P0 + 3) / 4, K1 = (P1 + 3) / 4, K = K0 + K1;
  lds_u32x4_cptr a0 = (lds_u32x4_cptr)(size_t)base0_lds + lane;
  lds_u32x4_cptr a1 = (lds_u32x4_cptr)(size_t)base1_lds + lane;
  const unsigned voff = (unsigned)lane * 16u;
  u32x4 fh[3], fl[3];
  auto fetch = [&](int i, int slot) {
    if (i < N0) {
      fh[slot] = a0[i * 128];
      fl[slot] = a0[i * 128 + 64];
    } else {
      fh[slot] = a1[(i - N0) * 128];
      fl[slot] = a1[(i - N0) * 128 + 64];
    }
  };
  fetch(0, 0);
  if (N > 1) fetch(1, 1);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i + 2 < N) fetch(i + 2, (i + 2) % 3);
    __builtin_amdgcn_sched_barrier(0);
    const int u = i / NMB, m = i % NMB;
    const f16x8 ah = __builtin_bit_cast(f16x8, fh[i % 3]), al = __builtin_bit_cast(f16x8, fl[i % 3]);
    acc[m] = MFMA16H_CROSS(ah, b[u].lo, acc[m]);
    acc[m] = MFMA16H_CROSS(al, b[u].hi, acc[m]);
    acc[m] = mfma16h(ah, b[u].hi, acc[m]);
    __builtin_amdgcn_sched_barrier(0);
    // pieces k with floor(k N / K) == i go out behind unit i
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if ((k * N) / K == i && issue) {
        if (k < K0) {
          const int pce = tw + 4 * k;
          if (pce < P0) glds16_s(src0 + pce * 256, voff, __builtin_amdgcn_readfirstlane(dst0_lds + (unsigned)pce * 1024u));
        } else {
          const int pce = tw + 4 * (k - K0);
          if (pce < P1) glds16_s(src1 + pce * 256, voff, __builtin_amdgcn_readfirstlane(dst1_lds + (unsigned)pce * 1024u));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NMB>
__device__ __forceinline__ void kblock_h(f32x16 (&acc)[NMB], unsigned base_lds, int lane, const f32x16& h, float mult) {
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = h[r];
  ksteps_h<NMB, 2>(acc, base_lds, lane, v, mult);
}

// x of lane (n, 0) and of lane (n, 1) in every lane of the pair: v_permlane32_swap (gfx950) exchanges the upper half of its
// first operand with the lower half of its second - one VALU instruction instead of a ds_bpermute round trip through LDS
// (~100 cycles on the critical path of every per-sample gain).  (inline asm: hipcc 7.2 folds the two results of
// __builtin_amdgcn_permlane32_swap into one.)  max / sum of the pair from (lo, hi) are the same bits in both lanes.
__device__ __forceinline__ void pair_halves(float x, float& lo, float& hi) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1" : "+v"(a), "+v"(b));
  lo = __builtin_bit_cast(float, a);
  hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float pair_max(float x) {
  float lo, hi;
  pair_halves(x, lo, hi);
  return fmaxf(lo, hi);
}
__device__ __forceinline__ float pair_sum(float x) {
  float lo, hi;
  pair_halves(x, lo, hi);
  return lo + hi;
}

// largest |value| of a sample's features held in NB accumulator blocks of its two lanes
template <int NB>
__device__ __forceinline__ float sample_absmax(const f32x16 (&a)[NB]) {
  float mx = 0.0f;
#pragma unroll
  for (int m = 0; m < NB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(a[m][r]));
  return pair_max(mx);
}

