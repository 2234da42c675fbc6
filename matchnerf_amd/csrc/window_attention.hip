// K6 — GMFlow single-head (shifted-)window attention, flash style, on the matrix pipe (gfx950):
// exact-f32 MFMA kernel and, by default, a split-bf16 variant (second half of this file).
//
// Replaces (paths relative to /root/reference/models/gmflow):
//   transformer.py:8-16    single_head_full_attention   (num_splits == 1)
//   transformer.py:46-105  single_head_split_window_attention: roll by -window/2, split into
//                          K x K windows, softmax(QK^T/sqrt(C) + mask) V, merge, roll back
//   transformer.py:19-43   generate_shift_window_attn_mask: -100 between tokens whose rolled
//                          positions lie in different wrap regions
//   utils.py:7-54          split_feature / merge_splits (pure index permutations)
// The reference materialises the [B*K*K, Lw, Lw] score tensor (157 MB per layer at
// 512x640x3 views) plus three rolled/split copies of q,k,v.  Here the roll, the window split
// and the region mask are index arithmetic on the token id, and scores never leave registers.
//
// MFMA formulation (d = C = 128, one head).  A wave owns 32 queries (lane&31 = query n).
//   S^T[key, n]  = sum_d K[key,d] Q[n,d]      : A = K tile (LDS image XOR-swizzled on the DMA
//                  source side, read as conflict-free ds_read_b128), B = Q (64 VGPRs/lane:
//                  half-wave hl holds d in [64hl, 64hl+64))
//   O^T[d, n]   += sum_key V[key,d] P[n,key]  : A = V tile (natural row-major LDS image),
//                  B = P — and P is *already* in B-operand layout: the C/D layout of S^T puts
//                  key (r&3)+8(r>>2)+4*hl of the tile in register r of lane (n,hl), which is
//                  what K-step r of the PV product needs.  No transpose, no LDS round trip.
// Softmax statistics (running max / sum per query) are per-lane scalars; the two half-waves
// of a query exchange them with one cross-half shuffle per tile.
// K/V tiles (32 keys) are double-buffered in LDS by global_load_lds DMA: the copy of tile t+1
// is in flight while tile t feeds 128 MFMAs per wave; one barrier per tile.
#include <stdlib.h>

#include "wa_common.hpp"

// ---- LDS-DMA helpers (see decoder.hip: an asm global_load_lds is invisible to hipcc's
// wait-count bookkeeping, so the prefetch of tile t+1 is not drained in front of tile t's reads)
__device__ __forceinline__ void wa_glds16(const float* gsrc, unsigned lds_byte_addr) {
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

// four consecutive 1-KiB pieces per M0 write: the instruction offset moves the global AND the LDS address of an LDS-DMA load
// (split_f16.hpp: glds16_sv4; tools/exp/ubench/dma_issue.hip: 40-43 instead of 74-78 cycles of issue per request)
__device__ __forceinline__ void wa_glds16x4(const float* gsrc, unsigned lds_byte_addr) {
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

// One K/V tile = 32 keys x 128 channels of each matrix = 2 x 16 KiB, copied as 32 pieces of
// 1 KiB (one wave instruction = 2 token rows).  V keeps its natural row-major image.  K is
// swizzled on the SOURCE side: the 16-byte column group c4 of LDS row `key` holds channels
// 4*(c4 ^ (key & 31)) .., so that the S^T operand read (32 lanes = 32 keys, same channel group)
// hits 16 distinct 16-byte slots per ds_read_b128 lane group instead of one (row stride 512 B).
template <int NQW>
__device__ __forceinline__ void wa_stage_tile(const float* __restrict__ k, const float* __restrict__ v,
                                              size_t seq_base, const WinGeom& G, int wy, int wx, int kt,
                                              float* kbuf, float* vbuf, int wave, int lane) {
  const int row_in_piece = lane >> 5, c4 = lane & 31;
  const unsigned kb = (unsigned)(size_t)(__attribute__((address_space(3))) float*)kbuf;
  const unsigned vb = (unsigned)(size_t)(__attribute__((address_space(3))) float*)vbuf;
  for (int piece = wave; piece < 16; piece += NQW) {
    const int key = piece * 2 + row_in_piece;
    int li = kt * WA_KT + key;
    if (li >= G.Lw) li = G.Lw - 1;
    int reg_unused;
    const int tok = win_token(G, wy, wx, li, reg_unused);
    const size_t row = seq_base + (size_t)tok * WA_C;
    wa_glds16(k + row + ((c4 ^ (key & 31)) << 2), __builtin_amdgcn_readfirstlane(kb + (unsigned)piece * 1024u));
    wa_glds16(v + row + (c4 << 2), __builtin_amdgcn_readfirstlane(vb + (unsigned)piece * 1024u));
  }
}

template <int NQW>
__global__ __launch_bounds__(NQW * 64, 2) void window_attention_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    float* __restrict__ out, WinGeom G, int shifted, float scale) {
  // double-buffered K and V tiles: [2][32 keys][128] each = 64 KiB per workgroup
  extern __shared__ __attribute__((aligned(16))) float wa_smem[];
#define WA_KBUF(i) (wa_smem + (i) * (WA_KT * WA_C))
#define WA_VBUF(i) (wa_smem + (2 + (i)) * (WA_KT * WA_C))

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hl = lane >> 5;
  const int win = blockIdx.y, b = blockIdx.z;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const size_t seq_base = (size_t)b * G.h * G.w * WA_C;

  const int n_tiles = (G.Lw + WA_KT - 1) / WA_KT;
  wa_stage_tile<NQW>(k, v, seq_base, G, wy, wx, 0, WA_KBUF(0), WA_VBUF(0), wave, lane);

  // ---- this lane's query (half-wave hl holds channels [64 hl, 64 hl + 64))
  const int qi_raw = (blockIdx.x * NQW + wave) * 32 + n;
  const bool q_ok = qi_raw < G.Lw;
  int q_region;
  const int q_tok = win_token(G, wy, wx, q_ok ? qi_raw : (G.Lw - 1), q_region);
  float qreg[64];
  {
    const float4* src = reinterpret_cast<const float4*>(q + seq_base + (size_t)q_tok * WA_C + hl * 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 t = src[i];
      qreg[4 * i] = t.x;
      qreg[4 * i + 1] = t.y;
      qreg[4 * i + 2] = t.z;
      qreg[4 * i + 3] = t.w;
    }
  }
  f32x16 o[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) o[m] = (f32x16)(0.0f);
  float m_run = -3.0e38f, l_run = 0.0f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < n_tiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < n_tiles)  // DMA of the next tile overlaps this tile's 128 MFMAs
      wa_stage_tile<NQW>(k, v, seq_base, G, wy, wx, kt + 1, WA_KBUF(cur ^ 1), WA_VBUF(cur ^ 1), wave, lane);

    // ---- S^T = K Q^T  (64 K-steps over d; one swizzled ds_read_b128 feeds 4 steps)
    f32x16 s = (f32x16)(0.0f);
    {
      const float* krow = WA_KBUF(cur) + n * WA_C;
#pragma unroll
      for (int q4 = 0; q4 < 16; ++q4) {
        const float4 ka = *reinterpret_cast<const float4*>(krow + (((hl * 16 + q4) ^ n) << 2));
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qreg[4 * q4 + 0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qreg[4 * q4 + 1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qreg[4 * q4 + 2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qreg[4 * q4 + 3], s, 0, 0, 0);
      }
    }
    // ---- scale, masks, online softmax
    float tmax = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hl;
      const int li = kt * WA_KT + key;
      float sv = s[r] * scale;
      if (shifted) {
        int kreg;
        (void)win_token(G, wy, wx, li < G.Lw ? li : (G.Lw - 1), kreg);
        if (kreg != q_region) sv += -100.0f;
      }
      if (li >= G.Lw) sv = -3.0e38f;
      s[r] = sv;
      tmax = fmaxf(tmax, sv);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(s[r] - m_new);
      s[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[m][r] *= alpha;
    // ---- O^T += V^T P^T  (16 K-steps over the tile's keys, 4 M-blocks over d)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hl;
      const float* va = WA_VBUF(cur) + key * WA_C + n;
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[0], s[r], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32], s[r], o[1], 0, 0, 0);
      o[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[64], s[r], o[2], 0, 0, 0);
      o[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[96], s[r], o[3], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile landed (this wave's pieces)
    __syncthreads();                                   // ... all waves' pieces; current tile consumed
  }
  // ---- normalise and store: register quad (4g..4g+3) of block m = channels m*32+8g+4hl+{0..3}
  if (q_ok) {
    const float inv_l = 1.0f / l_run;
    float* dst = out + seq_base + (size_t)q_tok * WA_C;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 t = make_float4(o[m][4 * g4] * inv_l, o[m][4 * g4 + 1] * inv_l,
                                     o[m][4 * g4 + 2] * inv_l, o[m][4 * g4 + 3] * inv_l);
        *reinterpret_cast<float4*>(dst + m * 32 + 8 * g4 + 4 * hl) = t;
      }
  }
}

// ============================================================================ split-bf16 variant
// Same flash-style loop with the products on v_mfma_f32_32x32x16_bf16: every fp32 operand is split into
// three bf16 terms and a product is accumulated in fp32 from six terms (decoder.hip, "split-bf16 matrix
// path": fp32-grade results at 16/6 of the f32 matrix rate).  Q is split once per workgroup and kept in
// 96 VGPRs; K and V fragments are split as they are read from the fp32 LDS tiles; P is split straight
// out of the score accumulator, whose layout makes registers 8t..8t+7 the operands of K16-step t.
// Operand layout: lane (n, half) supplies k = 8*half + j of a 16-wide K-step.
//   S^T: K-step t covers channels 16t..16t+15          (A = K row of key n, B = Q of query n)
//   O^T: K-step t covers the keys held by score registers 8t..8t+7, i.e. key(t, half, j) =
//        ((8t+j)&3) + 8((8t+j)>>2) + 4*half            (A = V^T rows d = 32m + n, B = P)
typedef __bf16 wa_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wa_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wa_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wa_u32x4 __attribute__((ext_vector_type(4)));

struct WaParts {
  wa_bf16x8 hi, mid, lo;
};

__device__ __forceinline__ unsigned wa_pk_bf16(float a, float b) {
  const wa_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wa_bf16x2));
}

__device__ __forceinline__ WaParts wa_split8(const float (&v)[8]) {
  wa_u32x4 H, M, L;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned h = wa_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    const unsigned m = wa_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    H[i] = h;
    M[i] = m;
    L[i] = wa_pk_bf16(sa, sb);
  }
  WaParts p;
  p.hi = __builtin_bit_cast(wa_bf16x8, H);
  p.mid = __builtin_bit_cast(wa_bf16x8, M);
  p.lo = __builtin_bit_cast(wa_bf16x8, L);
  return p;
}

__device__ __forceinline__ f32x16 wa_mfma16(wa_bf16x8 a, wa_bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// acc += A . B with both operands given as three bf16 terms (six product terms, small ones first)
__device__ __forceinline__ f32x16 wa_mac6(const WaParts& a, const WaParts& b, f32x16 acc) {
  acc = wa_mfma16(a.hi, b.lo, acc);
  acc = wa_mfma16(a.lo, b.hi, acc);
  acc = wa_mfma16(a.mid, b.mid, acc);
  acc = wa_mfma16(a.hi, b.mid, acc);
  acc = wa_mfma16(a.mid, b.hi, acc);
  acc = wa_mfma16(a.hi, b.hi, acc);
  return acc;
}

template <int NQW>
__global__ __launch_bounds__(NQW * 64, 2) void window_attention_bf16_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    float* __restrict__ out, WinGeom G, int shifted, float scale) {
  extern __shared__ __attribute__((aligned(16))) float wa_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hl = lane >> 5;
  const int win = blockIdx.y, b = blockIdx.z;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const size_t seq_base = (size_t)b * G.h * G.w * WA_C;

  const int n_tiles = (G.Lw + WA_KT - 1) / WA_KT;
  wa_stage_tile<NQW>(k, v, seq_base, G, wy, wx, 0, WA_KBUF(0), WA_VBUF(0), wave, lane);

  // ---- this lane's query, split once: K-step t holds channels 16t + 8 hl + j
  const int qi_raw = (blockIdx.x * NQW + wave) * 32 + n;
  const bool q_ok = qi_raw < G.Lw;
  int q_region;
  const int q_tok = win_token(G, wy, wx, q_ok ? qi_raw : (G.Lw - 1), q_region);
  WaParts qp[8];
  {
    const float4* src = reinterpret_cast<const float4*>(q + seq_base + (size_t)q_tok * WA_C + hl * 8);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 a = src[4 * t], c = src[4 * t + 1];
      const float qv[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      qp[t] = wa_split8(qv);
    }
  }
  f32x16 o[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) o[m] = (f32x16)(0.0f);
  float m_run = -3.0e38f, l_run = 0.0f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < n_tiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < n_tiles)
      wa_stage_tile<NQW>(k, v, seq_base, G, wy, wx, kt + 1, WA_KBUF(cur ^ 1), WA_VBUF(cur ^ 1), wave, lane);
    // ---- S^T = K Q^T: 8 K16-steps over d, two accumulators so that consecutive MFMA groups are independent
    f32x16 s0 = (f32x16)(0.0f), s1 = (f32x16)(0.0f);
    {
      const float* krow = WA_KBUF(cur) + n * WA_C;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int c4 = 4 * t + 2 * hl;  // 16-byte column groups of channels 16t + 8hl .. + 7 (swizzled by the key)
        const float4 a = *reinterpret_cast<const float4*>(krow + ((c4 ^ n) << 2));
        const float4 c = *reinterpret_cast<const float4*>(krow + (((c4 + 1) ^ n) << 2));
        const float kv[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        const WaParts kp = wa_split8(kv);
        if (t & 1)
          s1 = wa_mac6(kp, qp[t], s1);
        else
          s0 = wa_mac6(kp, qp[t], s0);
      }
    }
    f32x16 s = s0 + s1;
    // ---- scale, masks, online softmax (as in the f32 kernel)
    float tmax = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hl;
      const int li = kt * WA_KT + key;
      float sv = s[r] * scale;
      if (shifted) {
        int kreg;
        (void)win_token(G, wy, wx, li < G.Lw ? li : (G.Lw - 1), kreg);
        if (kreg != q_region) sv += -100.0f;
      }
      if (li >= G.Lw) sv = -3.0e38f;
      s[r] = sv;
      tmax = fmaxf(tmax, sv);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(s[r] - m_new);
      s[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[m][r] *= alpha;
    // ---- O^T += V^T P^T: 2 K16-steps over the tile's keys (score registers 8t..8t+7), 4 M-blocks over d
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[8 * t + j];
      const WaParts pp = wa_split8(pv);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * t + j;
          const int key = (r & 3) + 8 * (r >> 2) + 4 * hl;
          vv[j] = WA_VBUF(cur)[key * WA_C + 32 * m + n];
        }
        o[m] = wa_mac6(wa_split8(vv), pp, o[m]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (q_ok) {
    const float inv_l = 1.0f / l_run;
    float* dst = out + seq_base + (size_t)q_tok * WA_C;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 t = make_float4(o[m][4 * g4] * inv_l, o[m][4 * g4 + 1] * inv_l,
                                     o[m][4 * g4 + 2] * inv_l, o[m][4 * g4 + 3] * inv_l);
        *reinterpret_cast<float4*>(dst + m * 32 + 8 * g4 + 4 * hl) = t;
      }
  }
}

// ============================================================================ split-fp16 variant
// The same loop on v_mfma_f32_32x32x16_f16 with two range-managed fp16 terms per operand and three products per MAC
// (split_f16.hpp; decoder.hip "split-fp16 matrix path"): half the matrix instructions of the split-bf16 variant and
// less than half of its operand-split VALU work.  NOT the default: the tile maxima have to be known before the first
// product of a tile (read the whole tile, reduce across the wave, then split), which serialises a loop that is
// latency-bound already — 229 us per call against 204 us for the split-bf16 variant at 64x80 tokens on MI355X.
// Gains (exact powers of two):
//   Q   one per query (lane), from the query's 128 channels;
//   K,V one per 32-key TILE (the wave's 64 lanes hold the whole tile exactly once: in-lane max + wave butterfly);
//   P   fixed 2^14 (probabilities are <= 1).
// The score accumulator is un-scaled per lane (2^-(eq + ek) folded into the softmax scale); the output accumulator
// carries the V gain of the tile it was last updated with, and the change of gain from tile to tile rides in the
// multiplication by the softmax correction factor that happens anyway.
template <int NQW>
__global__ __launch_bounds__(NQW * 64, 2) void window_attention_f16_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    float* __restrict__ out, WinGeom G, int shifted, float scale) {
  extern __shared__ __attribute__((aligned(16))) float wa_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hl = lane >> 5;
  const int win = blockIdx.y, b = blockIdx.z;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const size_t seq_base = (size_t)b * G.h * G.w * WA_C;

  const int n_tiles = (G.Lw + WA_KT - 1) / WA_KT;
  wa_stage_tile<NQW>(k, v, seq_base, G, wy, wx, 0, WA_KBUF(0), WA_VBUF(0), wave, lane);

  // ---- this lane's query, split once: K-step t holds channels 16t + 8 hl + j
  const int qi_raw = (blockIdx.x * NQW + wave) * 32 + n;
  const bool q_ok = qi_raw < G.Lw;
  int q_region;
  const int q_tok = win_token(G, wy, wx, q_ok ? qi_raw : (G.Lw - 1), q_region);
  PartsH qp[8];
  int eq;
  {
    const float4* src = reinterpret_cast<const float4*>(q + seq_base + (size_t)q_tok * WA_C + hl * 8);
    float qv[64];
    float qmax = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 a = src[4 * t], c = src[4 * t + 1];
      qv[8 * t + 0] = a.x; qv[8 * t + 1] = a.y; qv[8 * t + 2] = a.z; qv[8 * t + 3] = a.w;
      qv[8 * t + 4] = c.x; qv[8 * t + 5] = c.y; qv[8 * t + 6] = c.z; qv[8 * t + 7] = c.w;
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) qmax = fmaxf(qmax, fabsf(qv[i]));
    qmax = fmaxf(qmax, __shfl_xor(qmax, 32, 64));
    eq = gain_exp(qmax);
    const float mq = pow2i(eq);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = qv[8 * t + j];
      qp[t] = split8h(v8, mq);
    }
  }
  f32x16 o[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) o[m] = (f32x16)(0.0f);
  float m_run = -3.0e38f, l_run = 0.0f;
  int ev_run = 0;  // V gain exponent the output accumulator currently carries (irrelevant while o == 0)

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < n_tiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < n_tiles)
      wa_stage_tile<NQW>(k, v, seq_base, G, wy, wx, kt + 1, WA_KBUF(cur ^ 1), WA_VBUF(cur ^ 1), wave, lane);
    // ---- S^T = K Q^T: the wave's 64 lanes hold the 32 x 128 K tile exactly once (lane = key row, half = channel
    // half of every K16-step): tile maximum -> one gain
    f32x16 s0 = (f32x16)(0.0f), s1 = (f32x16)(0.0f);
    int ek;
    {
      const float* krow = WA_KBUF(cur) + n * WA_C;
      float kv[64];
      float kmax = 0.0f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int c4 = 4 * t + 2 * hl;  // 16-byte column groups of channels 16t + 8hl .. + 7 (swizzled by the key)
        const float4 a = *reinterpret_cast<const float4*>(krow + ((c4 ^ n) << 2));
        const float4 c = *reinterpret_cast<const float4*>(krow + (((c4 + 1) ^ n) << 2));
        kv[8 * t + 0] = a.x; kv[8 * t + 1] = a.y; kv[8 * t + 2] = a.z; kv[8 * t + 3] = a.w;
        kv[8 * t + 4] = c.x; kv[8 * t + 5] = c.y; kv[8 * t + 6] = c.z; kv[8 * t + 7] = c.w;
      }
#pragma unroll
      for (int i = 0; i < 64; ++i) kmax = fmaxf(kmax, fabsf(kv[i]));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) kmax = fmaxf(kmax, __shfl_xor(kmax, off, 64));
      ek = gain_exp(kmax);
      const float mk = pow2i(ek);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = kv[8 * t + j];
        const PartsH kp = split8h(v8, mk);
        if (t & 1) {
          s1 = mfma16h(kp.hi, qp[t].lo, s1);
          s1 = mfma16h(kp.lo, qp[t].hi, s1);
          s1 = mfma16h(kp.hi, qp[t].hi, s1);
        } else {
          s0 = mfma16h(kp.hi, qp[t].lo, s0);
          s0 = mfma16h(kp.lo, qp[t].hi, s0);
          s0 = mfma16h(kp.hi, qp[t].hi, s0);
        }
      }
    }
    f32x16 s = s0 + s1;
    const float sscale = scale * pow2i(-(ek + eq));
    // ---- scale, masks, online softmax (as in the f32 kernel)
    float tmax = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hl;
      const int li = kt * WA_KT + key;
      float sv = s[r] * sscale;
      if (shifted) {
        int kreg;
        (void)win_token(G, wy, wx, li < G.Lw ? li : (G.Lw - 1), kreg);
        if (kreg != q_region) sv += -100.0f;
      }
      if (li >= G.Lw) sv = -3.0e38f;
      s[r] = sv;
      tmax = fmaxf(tmax, sv);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(s[r] - m_new);
      s[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // ---- O^T += V^T P^T: the lanes' 2 x 4 x 8 V values are the whole tile once more -> tile maximum -> gain
    float vv[2][4][8];
    float vmax = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * t + j;
          const int key = (r & 3) + 8 * (r >> 2) + 4 * hl;
          const float x = WA_VBUF(cur)[key * WA_C + 32 * m + n];
          vv[t][m][j] = x;
          vmax = fmaxf(vmax, fabsf(x));
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    const int ev = gain_exp(vmax);
    const float mv = pow2i(ev);
    const float corr = alpha * pow2i(ev - ev_run);  // softmax correction and change of V gain in one multiplication
    ev_run = ev;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[m][r] *= corr;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[8 * t + j];
      const PartsH pp = split8h(pv, 16384.0f);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const PartsH vp = split8h(vv[t][m], mv);
        o[m] = mfma16h(vp.hi, pp.lo, o[m]);
        o[m] = mfma16h(vp.lo, pp.hi, o[m]);
        o[m] = mfma16h(vp.hi, pp.hi, o[m]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (q_ok) {
    const float inv_l = pow2i(-(ev_run + 14)) / l_run;
    float* dst = out + seq_base + (size_t)q_tok * WA_C;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 t = make_float4(o[m][4 * g4] * inv_l, o[m][4 * g4 + 1] * inv_l,
                                     o[m][4 * g4 + 2] * inv_l, o[m][4 * g4 + 3] * inv_l);
        *reinterpret_cast<float4*>(dst + m * 32 + 8 * g4 + 4 * hl) = t;
      }
  }
}

// ============================================================================ pre-split fp16 variant
// The split-fp16 arithmetic above with the K / V operand fragments produced ONCE per call instead of once per wave
// and tile: in the kernels above every one of the Lw/32 query waves of a window converts the window's whole K and V
// (Lw x 128 values each) to matrix operands again - 70 % of their VALU instructions.  wa_presplit_kernel (one wave per
// 32-key tile) writes, per (batch, window, tile), a 32 KiB image that IS the LDS image the main loop wants:
//   K part  [t = 0..7 ][term hi|lo][lane 0..63] x 16 B   A operand of K16-step t of S^T (lane = key n, half hl)
//   V part  [t = 0..1 ][m = 0..3][term hi|lo][lane]  x 16 B   A operand of step t, output block m of O^T
// plus a 32-byte side record: the two power-of-two gain exponents of the tile (ek, ev) and the wrap region of each of
// its keys (shift mask, one nibble per key - the kernels above recompute them with an integer division per score).  The main kernel copies an image with
// 32 x 1 KiB LDS-DMA pieces and reads every fragment with one conflict-free ds_read_b128; the tile gains come from a
// scalar load issued one tile ahead, so nothing in the loop waits for a reduction any more (what made the in-loop
// split-fp16 variant slower than split-bf16).  Same products, same accumulation order as window_attention_f16_kernel.

__global__ __launch_bounds__(64) void wa_presplit_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                         u32x4* __restrict__ img, int* __restrict__ gains, WinGeom G) {
  const int lane = threadIdx.x, n = lane & 31, hl = lane >> 5;
  const int kt = blockIdx.x, win = blockIdx.y, b = blockIdx.z;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const size_t seq_base = (size_t)b * G.h * G.w * WA_C;
  const size_t tile = ((size_t)b * gridDim.y + win) * gridDim.x + kt;
  u32x4* dst = img + tile * (WA_IMG_BYTES / 16) + lane;
  int reg_unused;
  int ek, ev;
  {  // ---- K: lane (key n, half hl) holds channels 16t + 8hl .. + 7 of every K16-step t
    int li = kt * WA_KT + n;
    if (li >= G.Lw) li = G.Lw - 1;
    const int tok = win_token(G, wy, wx, li, reg_unused);
    const float4* src = reinterpret_cast<const float4*>(k + seq_base + (size_t)tok * WA_C + hl * 8);
    float kv[64];
    float kmax = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 a = src[4 * t], c = src[4 * t + 1];
      kv[8 * t + 0] = a.x; kv[8 * t + 1] = a.y; kv[8 * t + 2] = a.z; kv[8 * t + 3] = a.w;
      kv[8 * t + 4] = c.x; kv[8 * t + 5] = c.y; kv[8 * t + 6] = c.z; kv[8 * t + 7] = c.w;
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) kmax = fmaxf(kmax, fabsf(kv[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kmax = fmaxf(kmax, __shfl_xor(kmax, off, 64));
    ek = gain_exp(kmax);
    const float mk = pow2i(ek);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = kv[8 * t + j];
      const PartsH kp = split8h(v8, mk);
      dst[(2 * t) * 64] = __builtin_bit_cast(u32x4, kp.hi);
      dst[(2 * t + 1) * 64] = __builtin_bit_cast(u32x4, kp.lo);
    }
  }
  {  // ---- V: lane (n, hl) holds V[key(t, hl, j)][32m + n], key(t, hl, j) = ((8t+j)&3) + 8((8t+j)>>2) + 4hl
    float vv[2][4][8];
    float vmax = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 8 * t + j;
        int li = kt * WA_KT + (r & 3) + 8 * (r >> 2) + 4 * hl;
        if (li >= G.Lw) li = G.Lw - 1;
        const int tok = win_token(G, wy, wx, li, reg_unused);
        const float* row = v + seq_base + (size_t)tok * WA_C + n;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float x = row[32 * m];
          vv[t][m][j] = x;
          vmax = fmaxf(vmax, fabsf(x));
        }
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    ev = gain_exp(vmax);
    const float mv = pow2i(ev);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const PartsH vp = split8h(vv[t][m], mv);
        dst[(WA_IMG_VOFF / 64 + 2 * (4 * t + m)) * 64] = __builtin_bit_cast(u32x4, vp.hi);
        dst[(WA_IMG_VOFF / 64 + 2 * (4 * t + m) + 1) * 64] = __builtin_bit_cast(u32x4, vp.lo);
      }
  }
  int* rec = gains + WA_REC_INTS * tile;
  wa_store_regions(G, wy, wx, kt, lane, rec);
  if (lane == 0) {
    rec[0] = ek;
    rec[1] = ev;
  }
}

// x of lane (n, 0) and of lane (n, 1) in every lane of the pair: v_permlane32_swap (gfx950) exchanges the upper half of
// its first operand with the lower half of its second - one VALU instruction instead of a ds_bpermute round trip.
// (inline asm: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane32_swap into one)
__device__ __forceinline__ void wa_halves(float x, float& lo, float& hi) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1" : "+v"(a), "+v"(b));
  lo = __builtin_bit_cast(float, a);
  hi = __builtin_bit_cast(float, b);
}

template <int NQW>
__device__ __forceinline__ void wa_stage_image(const u32x4* __restrict__ img_tile, unsigned lds_base, int wave, int lane) {
  for (int piece = wave; piece < WA_IMG_BYTES / 1024; piece += NQW)
    wa_glds16(reinterpret_cast<const float*>(img_tile + piece * 64 + lane),
              __builtin_amdgcn_readfirstlane(lds_base + (unsigned)piece * 1024u));
}

// STATS (training forward): the row statistics of the online softmax - running maximum and sum of exponentials, log2 domain, the
// very quantities the backward's first pass would recompute - are written to [2][n_batch * h * w] floats behind `tl` (the debug
// pointer of the timeline build, otherwise unused: the inference instance keeps its signature and its code).
template <int NQW, bool STATS = false>
__global__ __launch_bounds__(NQW * 64, 2) void window_attention_pre_kernel(
    const float* __restrict__ q, const u32x4* __restrict__ img, const int* __restrict__ gains,
    float* __restrict__ out, WinGeom G, int shifted, float scale, int n_batch, int xcd_map, unsigned long long* tl) {
  extern __shared__ __attribute__((aligned(16))) float wa_smem[];  // two images: 64 KiB
  const unsigned smem0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)wa_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hl = lane >> 5;
  // 1-D grid, XCD-aware: workgroups go to the 8 XCDs round-robin by their linear id, so id = 8 * slot + xcd; all query
  // blocks of one window get the same xcd - the window's K / V images (1.3 MB at 1280 tokens) are then fetched into
  // ONE L2 and hit there by its other query blocks, instead of missing in all eight
  const int n_win = G.splits * G.splits;
  const int n_qb = (G.Lw + NQW * 32 - 1) / (NQW * 32);
  const int slot = xcd_map ? blockIdx.x >> 3 : blockIdx.x, wl = slot / n_qb;
  const int qblock = slot - wl * n_qb, gwin = xcd_map ? wl * 8 + (blockIdx.x & 7) : wl;
  if (gwin >= n_win * n_batch) return;
  const int b = gwin / n_win, win = gwin - b * n_win;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const size_t seq_base = (size_t)b * G.h * G.w * WA_C;

  const int n_tiles = (G.Lw + WA_KT - 1) / WA_KT;
  const size_t tile0 = (size_t)gwin * n_tiles;
  const u32x4* img_win = img + tile0 * (WA_IMG_BYTES / 16);
  const int4* rec_win = reinterpret_cast<const int4*>(gains + WA_REC_INTS * tile0);
  wa_stage_image<NQW>(img_win, smem0, wave, lane);
  int4 ra = rec_win[0], rb = rec_win[1];  // (ek, ev, regions 0-7, 8-15), (regions 16-23, 24-31, -, -)

  // ---- this lane's query, split once: K-step t holds channels 16t + 8 hl + j
  const int qi_raw = (qblock * NQW + wave) * 32 + n;
  const bool q_ok = qi_raw < G.Lw;
  int q_region;
  const int q_tok = win_token(G, wy, wx, q_ok ? qi_raw : (G.Lw - 1), q_region);
  PartsH qp[8];
  int eq;
  {
    const float4* src = reinterpret_cast<const float4*>(q + seq_base + (size_t)q_tok * WA_C + hl * 8);
    float qv[64];
    float qmax = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 a = src[4 * t], c = src[4 * t + 1];
      qv[8 * t + 0] = a.x; qv[8 * t + 1] = a.y; qv[8 * t + 2] = a.z; qv[8 * t + 3] = a.w;
      qv[8 * t + 4] = c.x; qv[8 * t + 5] = c.y; qv[8 * t + 6] = c.z; qv[8 * t + 7] = c.w;
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) qmax = fmaxf(qmax, fabsf(qv[i]));
    qmax = fmaxf(qmax, __shfl_xor(qmax, 32, 64));
    eq = gain_exp(qmax);
    const float mq = pow2i(eq);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = qv[8 * t + j];
      qp[t] = split8h(v8, mq);
    }
  }
  f32x16 o[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) o[m] = (f32x16)(0.0f);
  float m_run = -3.0e38f, l_run = 0.0f;
  int ev_run = 0;  // V gain exponent the output accumulator currently carries (irrelevant while o == 0)

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#ifdef MNERF_TIMELINE  // debug build (tools/exp/build_timeline.sh): cycles per loop phase, summed over the tiles
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tmark = __builtin_amdgcn_s_memtime();
#define WA_STAMP(i)                                          \
  do {                                                       \
    const unsigned long long now = __builtin_amdgcn_s_memtime(); \
    tph[i] += now - tmark;                                   \
    tmark = now;                                             \
  } while (0)
#else
#define WA_STAMP(i)
#endif
  // DMA pieces of the next image are issued BETWEEN the rows of matrix instructions (8 slots per tile): issued in one
  // block at the top of the tile they cost 880 cycles of pure issue (110 each) with the matrix pipe idle
  constexpr int PPS = (WA_IMG_BYTES / 1024) / NQW / 8;  // pieces per slot and wave
  const float l2e = 1.44269504088896f;
  const float mask_l2 = -100.0f * l2e;  // the reference's -100 across wrap regions, in the log2 domain
  const bool ragged = (G.Lw & (WA_KT - 1)) != 0;
  for (int kt = 0; kt < n_tiles; ++kt) {
    const int cur = kt & 1;
    // the last tile prefetches itself into the idle buffer (no branch around the pieces)
    const int ktn = kt + 1 < n_tiles ? kt + 1 : kt;
    const u32x4* img_next = img_win + (size_t)ktn * (WA_IMG_BYTES / 16) + lane;
    const unsigned lds_next = smem0 + (unsigned)(cur ^ 1) * WA_IMG_BYTES;
    // (round 6) a wave copies a CONTIGUOUS range of the next image, 32 / NQW pieces, in runs of four per M0 write; the runs are
    // spread over the tile's eight request slots (NQW = 4: slots 0 and 4)
#define WA_SLOT(slot)                                                                        \
  do {                                                                                       \
    constexpr int PW_ = 8 * PPS, EVERY_ = 8 / (PW_ / 4);                                     \
    if ((slot) % EVERY_ == 0) {                                                              \
      const int piece = wave * PW_ + 4 * ((slot) / EVERY_);                                  \
      wa_glds16x4(reinterpret_cast<const float*>(img_next + piece * 64),                     \
                  __builtin_amdgcn_readfirstlane(lds_next + (unsigned)piece * 1024u));       \
    }                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
    WA_STAMP(0);
    const int ek = ra.x, ev = ra.y;
    const unsigned regw[4] = {(unsigned)ra.z, (unsigned)ra.w, (unsigned)rb.x, (unsigned)rb.y};
    lds_u32x4_cptr frag = (lds_u32x4_cptr)(size_t)(smem0 + (unsigned)cur * WA_IMG_BYTES) + lane;
    // ---- S^T = K Q^T: 8 K16-steps x 3 products on FOUR accumulators, term-major inside a group of four steps:
    // consecutive matrix instructions never wait for each other's result
    f32x16 sa[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) sa[a] = (f32x16)(0.0f);
    // fragment reads staged so that at most eight are live (lo of a group dies after its first row):
    //   lo.hi g0 | hi.lo g0 | hi.hi g0 | lo.hi g1 | hi.lo g1 | hi.hi g1, reads of group 1 two rows ahead of their use
    f16x8 khi[4], klo[4], khi1[4], klo1[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      klo[a] = __builtin_bit_cast(f16x8, frag[(2 * a + 1) * 64]);
      khi[a] = __builtin_bit_cast(f16x8, frag[(2 * a) * 64]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) sa[a] = mfma16h(klo[a], qp[a].hi, sa[a]);
#pragma unroll
    for (int a = 0; a < 4; ++a) klo1[a] = __builtin_bit_cast(f16x8, frag[(2 * (4 + a) + 1) * 64]);
    WA_SLOT(0);
#pragma unroll
    for (int a = 0; a < 4; ++a) sa[a] = mfma16h(khi[a], qp[a].lo, sa[a]);
    WA_SLOT(1);
#pragma unroll
    for (int a = 0; a < 4; ++a) sa[a] = mfma16h(khi[a], qp[a].hi, sa[a]);
#pragma unroll
    for (int a = 0; a < 4; ++a) khi1[a] = __builtin_bit_cast(f16x8, frag[(2 * (4 + a)) * 64]);
    WA_SLOT(2);
#pragma unroll
    for (int a = 0; a < 4; ++a) sa[a] = mfma16h(klo1[a], qp[4 + a].hi, sa[a]);
    WA_SLOT(3);
#pragma unroll
    for (int a = 0; a < 4; ++a) sa[a] = mfma16h(khi1[a], qp[4 + a].lo, sa[a]);
    WA_SLOT(4);
#pragma unroll
    for (int a = 0; a < 4; ++a) sa[a] = mfma16h(khi1[a], qp[4 + a].hi, sa[a]);
    WA_SLOT(5);
    f32x16 s = (sa[0] + sa[1]) + (sa[2] + sa[3]);
#ifdef MNERF_TIMELINE
    asm volatile("" ::"v"(s[0]));
#endif
    WA_STAMP(1);  // scores
    // V fragments of the first key half and the next tile's record are requested here: the softmax below has no LDS
    // or scalar-memory traffic of its own, so both latencies disappear under it (the record load at the top of the
    // tile was waited for by the first fragment read: ~400 exposed cycles per tile)
    f16x8 vhi[4], vlo[4], vhi1[4], vlo1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      vlo[m] = __builtin_bit_cast(f16x8, frag[(WA_IMG_VOFF / 64 + 2 * m + 1) * 64]);
      vhi[m] = __builtin_bit_cast(f16x8, frag[(WA_IMG_VOFF / 64 + 2 * m) * 64]);
    }
    const int4 ra_next = rec_win[2 * ktn], rb_next = rec_win[2 * ktn + 1];
    __builtin_amdgcn_sched_barrier(0);
    // ---- scale, masks, online softmax in the log2 domain (log2 e folded into the score scale: one v_exp_f32 per key)
    const float sscale = scale * l2e * pow2i(-(ek + eq));
    float tmax = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sv = s[r] * sscale;
      if (shifted) {  // key (r, hl) = nibble (r & 3) + 4 hl of word r >> 2
        const int kreg = (int)((regw[r >> 2] >> (4 * (r & 3) + 16 * hl)) & 15u);
        if (kreg != q_region) sv += mask_l2;
      }
      s[r] = sv;
    }
    if (ragged && kt == n_tiles - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * WA_KT + (r & 3) + 8 * (r >> 2) + 4 * hl >= G.Lw) s[r] = -3.0e38f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    {
      float lo, hi;
      wa_halves(tmax, lo, hi);
      tmax = fmaxf(lo, hi);
    }
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] - m_new);
      s[r] = p;
      psum += p;
    }
    {
      float lo, hi;
      wa_halves(psum, lo, hi);
      psum = lo + hi;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#ifdef MNERF_TIMELINE
    asm volatile("" ::"v"(l_run));
#endif
    WA_STAMP(2);  // softmax
    // ---- O^T += V^T P^T; the softmax correction and the change of V gain ride in one multiplication
    const float corr = alpha * pow2i(ev - ev_run);
    ev_run = ev;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[m][r] *= corr;
    {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[j];
      const PartsH pp = split8h(pv, 16384.0f);
      // term-major over the four output blocks: consecutive matrix instructions are independent
#pragma unroll
      for (int m = 0; m < 4; ++m) o[m] = mfma16h(vlo[m], pp.hi, o[m]);
#pragma unroll
      for (int m = 0; m < 4; ++m) vlo1[m] = __builtin_bit_cast(f16x8, frag[(WA_IMG_VOFF / 64 + 2 * (4 + m) + 1) * 64]);
      WA_SLOT(6);
#pragma unroll
      for (int m = 0; m < 4; ++m) o[m] = mfma16h(vhi[m], pp.lo, o[m]);
      WA_SLOT(7);
#pragma unroll
      for (int m = 0; m < 4; ++m) o[m] = mfma16h(vhi[m], pp.hi, o[m]);
#pragma unroll
      for (int m = 0; m < 4; ++m) vhi1[m] = __builtin_bit_cast(f16x8, frag[(WA_IMG_VOFF / 64 + 2 * (4 + m)) * 64]);
    }
    {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[8 + j];
      const PartsH pp = split8h(pv, 16384.0f);
#pragma unroll
      for (int m = 0; m < 4; ++m) o[m] = mfma16h(vlo1[m], pp.hi, o[m]);
#pragma unroll
      for (int m = 0; m < 4; ++m) o[m] = mfma16h(vhi1[m], pp.lo, o[m]);
#pragma unroll
      for (int m = 0; m < 4; ++m) o[m] = mfma16h(vhi1[m], pp.hi, o[m]);
    }
#ifdef MNERF_TIMELINE
    asm volatile("" ::"v"(o[3][0]));
#endif
    WA_STAMP(3);  // output update
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WA_STAMP(4);  // own DMA pieces
    __syncthreads();
    WA_STAMP(5);  // barrier
    ra = ra_next;
    rb = rb_next;
  }
#undef WA_SLOT
#ifdef MNERF_TIMELINE
  if (tl && lane == 0 && blockIdx.x < 64)
    for (int i = 0; i < 6; ++i) tl[(blockIdx.x * NQW + wave) * 6 + i] = tph[i];
#endif
  if (q_ok) {
    const float inv_l = pow2i(-(ev_run + 14)) / l_run;
    float* dst = out + seq_base + (size_t)q_tok * WA_C;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 t = make_float4(o[m][4 * g4] * inv_l, o[m][4 * g4 + 1] * inv_l,
                                     o[m][4 * g4 + 2] * inv_l, o[m][4 * g4 + 3] * inv_l);
        *reinterpret_cast<float4*>(dst + m * 32 + 8 * g4 + 4 * hl) = t;
      }
#ifndef MNERF_TIMELINE
    if constexpr (STATS) {
      if (hl == 0) {  // both halves of a query's lane pair hold the same statistics
        float* st = reinterpret_cast<float*>(tl);
        const size_t tok = (size_t)b * G.h * G.w + q_tok, n_tok = (size_t)n_batch * G.h * G.w;
        st[tok] = m_run;
        st[n_tok + tok] = l_run;
      }
    }
#endif
  }
}

extern "C" size_t mnerf_window_attention_workspace_bytes(int32_t batch, int32_t h, int32_t w, int32_t num_splits) {
  return wa_workspace_bytes(batch, h, w, num_splits);
}

// the attention kernel over prepared images (filled by wa_presplit_kernel or by the q|k|v kernel, qkv.hip)
static int wa_launch_main(const char* who, const float* q, float* out, const WinGeom& G, int do_shift, int32_t batch,
                          int32_t num_splits, void* workspace, hipStream_t st, float* row_stats = nullptr) {
  const int n_tiles = (G.Lw + WA_KT - 1) / WA_KT;
  const int n_win = num_splits * num_splits;
  const u32x4* img = reinterpret_cast<const u32x4*>(workspace);
  const int* gains = reinterpret_cast<const int*>(reinterpret_cast<const char*>(workspace) + (size_t)batch * n_win * n_tiles * WA_IMG_BYTES);
  const float scale = 1.0f / sqrtf((float)WA_C);
  const size_t lds = 2 * WA_IMG_BYTES;
  static std::atomic<unsigned long long> attr{0};
  if (mnerf_once_per_device(attr)) {
    (void)hipFuncSetAttribute((const void*)window_attention_pre_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)window_attention_pre_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)window_attention_pre_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)window_attention_pre_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  unsigned long long* tl = nullptr;
#ifdef MNERF_TIMELINE
  if (const char* e = getenv("MNERF_WA_TIMELINE_PTR")) tl = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  const long long wgs4 = (long long)((G.Lw + 127) / 128) * n_win * batch;
  const long long win_groups = ((long long)n_win * batch + 7) / 8;  // windows per XCD
  const int xcd = mnerf_tune().wa_xcd;
  const bool four = wgs4 >= mnerf_tune().wa_min4;
  const int n_qb = four ? (G.Lw + 127) / 128 : (G.Lw + 63) / 64;
  const dim3 grid((unsigned)(8 * win_groups * n_qb));
#ifndef MNERF_TIMELINE
  if (row_stats) {
    unsigned long long* sp = reinterpret_cast<unsigned long long*>(row_stats);
    if (four)
      hipLaunchKernelGGL((window_attention_pre_kernel<4, true>), grid, dim3(256), lds, st, q, img, gains, out, G, do_shift, scale, batch, xcd, sp);
    else
      hipLaunchKernelGGL((window_attention_pre_kernel<2, true>), grid, dim3(128), lds, st, q, img, gains, out, G, do_shift, scale, batch, xcd, sp);
    return mnerf_check_launch(who);
  }
#else
  MNERF_REQUIRE(!row_stats, MNERF_E_UNSUPPORTED, "%s: no row statistics in the timeline build", who);
#endif
  if (four)
    hipLaunchKernelGGL(window_attention_pre_kernel<4>, grid, dim3(256), lds, st, q, img, gains, out, G, do_shift, scale, batch, xcd, tl);
  else
    hipLaunchKernelGGL(window_attention_pre_kernel<2>, grid, dim3(128), lds, st, q, img, gains, out, G, do_shift, scale, batch, xcd, tl);
  return mnerf_check_launch(who);
}

static int wa_check_workspace(const char* who, const void* workspace, size_t workspace_bytes, int32_t batch, int32_t h,
                              int32_t w, int32_t num_splits) {
  const size_t need = wa_workspace_bytes(batch, h, w, num_splits);
  MNERF_REQUIRE(workspace && mnerf_aligned16(workspace), MNERF_E_ALIGN, "%s: workspace NULL or not 16-byte aligned", who);
  MNERF_REQUIRE(workspace_bytes >= need, MNERF_E_RANGE, "%s: workspace %zu bytes < %zu", who, workspace_bytes, need);
  return MNERF_OK;
}

static int wa_presplit_impl(const char* who, const float* q, const float* k, const float* v, float* out, float* row_stats,
                            int32_t batch, int32_t h, int32_t w, int32_t num_splits, int32_t shifted, void* workspace,
                            size_t workspace_bytes, void* stream) {
  MNERF_REQUIRE(q && k && v && out, MNERF_E_NULL, "%s: NULL buffer", who);
  MNERF_REQUIRE(mnerf_aligned16(q) && mnerf_aligned16(k) && mnerf_aligned16(v) && mnerf_aligned16(out), MNERF_E_ALIGN,
                "%s: buffers must be 16-byte aligned", who);
  WinGeom G;
  int do_shift;
  if (const int rc = wa_geometry(who, batch, h, w, num_splits, shifted, G, do_shift)) return rc;
  if (batch == 0) return MNERF_OK;
  if (const int rc = wa_check_workspace(who, workspace, workspace_bytes, batch, h, w, num_splits)) return rc;
  const int n_tiles = (G.Lw + WA_KT - 1) / WA_KT;
  const int n_win = num_splits * num_splits;
  u32x4* img = reinterpret_cast<u32x4*>(workspace);
  int* gains = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + (size_t)batch * n_win * n_tiles * WA_IMG_BYTES);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wa_presplit_kernel, dim3(n_tiles, n_win, batch), dim3(64), 0, st, k, v, img, gains, G);
  return wa_launch_main(who, q, out, G, do_shift, batch, num_splits, workspace, st, row_stats);
}

extern "C" int mnerf_window_attention_presplit(const float* q, const float* k, const float* v, float* out,
                                               int32_t batch, int32_t h, int32_t w, int32_t num_splits, int32_t shifted,
                                               void* workspace, size_t workspace_bytes, void* stream) {
  return wa_presplit_impl("mnerf_window_attention_presplit", q, k, v, out, nullptr, batch, h, w, num_splits, shifted, workspace,
                          workspace_bytes, stream);
}

// the training forward: the same launch, and the softmax's row statistics for mnerf_window_attention_backward_stats
extern "C" int mnerf_window_attention_presplit_stats(const float* q, const float* k, const float* v, float* out, float* row_stats,
                                                     int32_t batch, int32_t h, int32_t w, int32_t num_splits, int32_t shifted,
                                                     void* workspace, size_t workspace_bytes, void* stream) {
  const char* who = "mnerf_window_attention_presplit_stats";
  MNERF_REQUIRE(row_stats, MNERF_E_NULL, "%s: row_stats is NULL", who);
  return wa_presplit_impl(who, q, k, v, out, row_stats, batch, h, w, num_splits, shifted, workspace, workspace_bytes, stream);
}

extern "C" int mnerf_window_attention_images(const float* q, float* out, int32_t batch, int32_t h, int32_t w,
                                             int32_t num_splits, int32_t shifted, const void* workspace,
                                             size_t workspace_bytes, void* stream) {
  const char* who = "mnerf_window_attention_images";
  MNERF_REQUIRE(q && out, MNERF_E_NULL, "%s: NULL buffer", who);
  MNERF_REQUIRE(mnerf_aligned16(q) && mnerf_aligned16(out), MNERF_E_ALIGN, "%s: buffers must be 16-byte aligned", who);
  WinGeom G;
  int do_shift;
  if (const int rc = wa_geometry(who, batch, h, w, num_splits, shifted, G, do_shift)) return rc;
  if (batch == 0) return MNERF_OK;
  if (const int rc = wa_check_workspace(who, workspace, workspace_bytes, batch, h, w, num_splits)) return rc;
  return wa_launch_main(who, q, out, G, do_shift, batch, num_splits, const_cast<void*>(workspace), (hipStream_t)stream);
}

extern "C" int mnerf_window_attention(const float* q, const float* k, const float* v, float* out,
                                      int32_t batch, int32_t h, int32_t w, int32_t num_splits,
                                      int32_t shifted, int32_t math, void* stream) {
  MNERF_REQUIRE(q && k && v && out, MNERF_E_NULL, "mnerf_window_attention: NULL buffer");
  MNERF_REQUIRE(mnerf_aligned16(q) && mnerf_aligned16(k) && mnerf_aligned16(v) && mnerf_aligned16(out),
                MNERF_E_ALIGN, "mnerf_window_attention: buffers must be 16-byte aligned");
  MNERF_REQUIRE(batch >= 0 && h >= 1 && w >= 1 && num_splits >= 1, MNERF_E_RANGE,
                "mnerf_window_attention: batch=%d h=%d w=%d splits=%d", batch, h, w, num_splits);
  MNERF_REQUIRE(h % num_splits == 0 && w % num_splits == 0, MNERF_E_RANGE,
                "mnerf_window_attention: %dx%d not divisible into %d splits", h, w, num_splits);
  MNERF_REQUIRE(batch <= 65535 && num_splits * num_splits <= 65535, MNERF_E_RANGE,
                "mnerf_window_attention: grid too large");
  MNERF_REQUIRE(math == MNERF_WA_SPLIT_BF16 || math == MNERF_WA_EXACT_F32 || math == MNERF_WA_SPLIT_F16, MNERF_E_UNSUPPORTED,
                "mnerf_window_attention: math=%d", math);
  if (batch == 0) return MNERF_OK;
  WinGeom G;
  G.h = h;
  G.w = w;
  G.splits = num_splits;
  G.wh = h / num_splits;
  G.ww = w / num_splits;
  const int do_shift = (shifted && num_splits > 1) ? 1 : 0;
  G.sh = do_shift ? G.wh / 2 : 0;
  G.sw = do_shift ? G.ww / 2 : 0;
  G.Lw = G.wh * G.ww;
  const float scale = 1.0f / sqrtf((float)WA_C);
  hipStream_t st = (hipStream_t)stream;
  const long long wgs4 = (long long)((G.Lw + 127) / 128) * num_splits * num_splits * batch;
  const size_t lds = 4 * WA_KT * WA_C * sizeof(float);  // 64 KiB: K and V tiles, double buffered
  const bool split = math == MNERF_WA_SPLIT_BF16;  // split-bf16 products (the host's default), split-fp16 products or the exact-f32 MFMA
  const bool split_h = math == MNERF_WA_SPLIT_F16;
  static std::atomic<unsigned long long> attr_f32{0}, attr_split{0}, attr_split_h{0};
  if (split_h && mnerf_once_per_device(attr_split_h)) {
    (void)hipFuncSetAttribute((const void*)window_attention_f16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)window_attention_f16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (!split && !split_h && mnerf_once_per_device(attr_f32)) {
    (void)hipFuncSetAttribute((const void*)window_attention_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)window_attention_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (split && mnerf_once_per_device(attr_split)) {
    (void)hipFuncSetAttribute((const void*)window_attention_bf16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)window_attention_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int min4 = mnerf_tune().wa_min4;
  if (wgs4 >= min4) {
    dim3 grid((G.Lw + 127) / 128, num_splits * num_splits, batch);
    if (split_h)
      hipLaunchKernelGGL(window_attention_f16_kernel<4>, grid, dim3(256), lds, st, q, k, v, out, G, do_shift, scale);
    else if (split)
      hipLaunchKernelGGL(window_attention_bf16_kernel<4>, grid, dim3(256), lds, st, q, k, v, out, G, do_shift, scale);
    else
      hipLaunchKernelGGL(window_attention_kernel<4>, grid, dim3(256), lds, st, q, k, v, out, G, do_shift, scale);
  } else {
    dim3 grid((G.Lw + 63) / 64, num_splits * num_splits, batch);
    if (split_h)
      hipLaunchKernelGGL(window_attention_f16_kernel<2>, grid, dim3(128), lds, st, q, k, v, out, G, do_shift, scale);
    else if (split)
      hipLaunchKernelGGL(window_attention_bf16_kernel<2>, grid, dim3(128), lds, st, q, k, v, out, G, do_shift, scale);
    else
      hipLaunchKernelGGL(window_attention_kernel<2>, grid, dim3(128), lds, st, q, k, v, out, G, do_shift, scale);
  }
  return mnerf_check_launch("mnerf_window_attention");
}
