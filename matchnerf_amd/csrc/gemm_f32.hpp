// Exact-fp32 MFMA GEMM with strided operands and the Linear-layer wrappers built on it (shared by decoder_backward.hip and
// encoder_backward.hip; kernels are templates / static so that two translation units may include this header).
#pragma once
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ exact-fp32 GEMM  C[I,J] (+)= sum_k A(i,k) B(k,j) (+ bias[j])
struct GemmArgs {
  const float* a;
  long long sa_i, sa_k;
  const float* b;
  long long sb_k, sb_j;
  float* c;
  long long sc_i;
  const float* bias;
  int I, J, K;
  int mode;  // 0: C = ..   1: C += .. (one workgroup per tile)   2: atomicAdd (split-K)
};

template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float As[16][64 + 4], Bs[16][64 + 4];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wi = w >> 1, wj = w & 1;
  const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
  const int chunks = (g.K + 15) / 16, per = (chunks + gridDim.z - 1) / gridDim.z;
  const int kb = blockIdx.z * per * 16;
  const int ke = min(g.K, kb + per * 16);
  f32x16 acc = (f32x16)(0.0f);
  for (int k0 = kb; k0 < ke; k0 += 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = A_KCONT ? (t & 15) : ((t >> 6) + 4 * r);
      const int ii = A_KCONT ? ((t >> 4) + 16 * r) : (t & 63);
      const int i = i0 + ii, k = k0 + kk;
      As[kk][ii] = (i < g.I && k < ke) ? g.a[(long long)i * g.sa_i + (long long)k * g.sa_k] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = B_KCONT ? (t & 15) : ((t >> 6) + 4 * r);
      const int jj = B_KCONT ? ((t >> 4) + 16 * r) : (t & 63);
      const int j = j0 + jj, k = k0 + kk;
      Bs[kk][jj] = (j < g.J && k < ke) ? g.b[(long long)k * g.sb_k + (long long)j * g.sb_j] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      const float av = As[2 * k2 + (l >> 5)][wi * 32 + (l & 31)];
      const float bv = Bs[2 * k2 + (l >> 5)][wj * 32 + (l & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int j = j0 + wj * 32 + (l & 31);
  if (j >= g.J) return;
  const float bj = (g.bias && blockIdx.z == 0) ? g.bias[j] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wi * 32 + 8 * (r >> 2) + (l >> 5) * 4 + (r & 3);
    if (i >= g.I) continue;
    float* p = g.c + (long long)i * g.sc_i + j;
    const float v = acc[r] + bj;
    if (g.mode == 0) *p = v;
    else if (g.mode == 1) *p += v;
    else atomicAdd(p, v);
  }
}

// The same product with a 128 x 128 tile per workgroup (round 4: the transformer layers' backward runs ~550 GFLOP per training
// step through this file, and the 64 x 64 kernel above sustains ~34 TFLOP/s): a wave owns 64 x 64 = 2 x 2 blocks, so one
// K-step is four matrix instructions for four LDS reads (two per instruction above); the operands of chunk c + 1 are fetched
// into registers while chunk c is multiplied and written to the OTHER LDS buffer afterwards: one barrier per 16-wide chunk.
template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(256) void gemm_f32_tile128_kernel(GemmArgs g) {
  __shared__ float As[2][16][128 + 4], Bs[2][16][128 + 4];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wi = w >> 1, wj = w & 1;
  const int m = l & 31, kk2 = l >> 5;
  const int i0 = blockIdx.x * 128, j0 = blockIdx.y * 128;
  const int chunks = (g.K + 15) / 16, per = (chunks + gridDim.z - 1) / gridDim.z;
  const int kb = blockIdx.z * per * 16;
  const int ke = min(g.K, kb + per * 16);
  float ra[8], rb[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int kk = A_KCONT ? (t & 15) : ((t >> 7) + 2 * r);
      const int ii = A_KCONT ? ((t >> 4) + 16 * r) : (t & 127);
      const int i = i0 + ii, k = k0 + kk;
      ra[r] = (i < g.I && k < ke) ? g.a[(long long)i * g.sa_i + (long long)k * g.sa_k] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int kk = B_KCONT ? (t & 15) : ((t >> 7) + 2 * r);
      const int jj = B_KCONT ? ((t >> 4) + 16 * r) : (t & 127);
      const int j = j0 + jj, k = k0 + kk;
      rb[r] = (j < g.J && k < ke) ? g.b[(long long)k * g.sb_k + (long long)j * g.sb_j] : 0.0f;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      As[buf][A_KCONT ? (t & 15) : ((t >> 7) + 2 * r)][A_KCONT ? ((t >> 4) + 16 * r) : (t & 127)] = ra[r];
      Bs[buf][B_KCONT ? (t & 15) : ((t >> 7) + 2 * r)][B_KCONT ? ((t >> 4) + 16 * r) : (t & 127)] = rb[r];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) acc[bi][bj] = (f32x16)(0.0f);
  if (kb < ke) {
    fetch(kb);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = kb; k0 < ke; k0 += 16, buf ^= 1) {
    const bool more = k0 + 16 < ke;
    if (more) fetch(k0 + 16);  // in flight during the matrix instructions below
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      const float a0 = As[buf][2 * k2 + kk2][wi * 64 + m], a1 = As[buf][2 * k2 + kk2][wi * 64 + 32 + m];
      const float b0 = Bs[buf][2 * k2 + kk2][wj * 64 + m], b1 = Bs[buf][2 * k2 + kk2][wj * 64 + 32 + m];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);  // (the other buffer was last read before the previous barrier)
    __syncthreads();
  }
#pragma unroll
  for (int bj = 0; bj < 2; ++bj) {
    const int j = j0 + wj * 64 + bj * 32 + m;
    if (j >= g.J) continue;
    const float bias_j = (g.bias && blockIdx.z == 0) ? g.bias[j] : 0.0f;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * 64 + bi * 32 + 8 * (r >> 2) + kk2 * 4 + (r & 3);
        if (i >= g.I) continue;
        float* p = g.c + (long long)i * g.sc_i + j;
        const float v = acc[bi][bj][r] + bias_j;
        if (g.mode == 0) *p = v;
        else if (g.mode == 1) *p += v;
        else atomicAdd(p, v);
      }
  }
}

// ------------------------------------------------------------------ the same product on the 16-bit matrix pipe: split-bf16
// Every fp32 operand element is three bf16 terms (hi + mid + lo = the fp32 value exactly for |x| >= 2^-100: 3 x 8 significand bits;
// below that the third term sinks into the subnormals and the value keeps >= 16 bits: tests/test_split_bf16_cpu.py), a product is
// accumulated in fp32 from the six term products that are not below 2^-24 of it (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi;
// the scheme of the decoder's "bf16x6" matrix path, decoder.hip) on v_mfma_f32_32x32x16_bf16: 16 K-elements per 32-cycle
// instruction against 2 for the exact-f32 one, i.e. 8/6 x 2 = 2.7 x its rate, with fp32's exponent range (no scaling, unlike the
// split-fp16 form) and fp32-grade results (tests/test_gemm_gpu.py: error against float64 at the level of an fp32 FMA chain).
// The operands are split ONCE per tile load — each element then serves 128 rows / columns of the tile — and wait in LDS as
// [term][k half][row] 16-byte fragments (lane (n, half) of a 16-bit matrix instruction supplies k = 8 half .. 8 half + 7 of row n).
typedef __bf16 gemm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gemm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gemm_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gemm_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned gemm_pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even)
  const gemm_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gemm_bf16x2));
}
__device__ __forceinline__ void gemm_split8(const float (&v)[8], gemm_u32x4& H, gemm_u32x4& M, gemm_u32x4& L) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned h = gemm_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    const unsigned m = gemm_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    H[i] = h;
    M[i] = m;
    L[i] = gemm_pk_bf16(sa, sb);
  }
}

// TI x TJ tile per workgroup (128 or 64 each), the four waves as 2 x 2, a wave owns (TI/2) x (TJ/2) = BI x BJ blocks of 32 x 32:
// one 16-wide K chunk is 6 BI BJ matrix instructions for 3 (BI + BJ) ds_read_b128 (128 x 128: 24 for 12).  Fetch items: (row,
// k half) of A (2 TI of them) and of B (2 TJ), dealt to the 256 threads in order; an item is 8 consecutive k of one row.
// a_vec / b_vec (host): the K-contiguous operand may be fetched as two float4 (row stride and base multiples of 4 floats).
// The 64-tiles are for products with few 128-tiles (N x 128 x 128: 240 of them = one 4-wave workgroup per CU, which cannot hide
// the latency of its own operand fetches; 960 64-tiles can: 25 -> 14 us).
template <bool A_KCONT, bool B_KCONT, int TI, int TJ>
__global__ __launch_bounds__(256) void gemm_b6_kernel(GemmArgs g, int a_vec, int b_vec) {
  constexpr int BI = TI / 64, BJ = TJ / 64, ITEMS = 2 * TI + 2 * TJ, NIT = ITEMS / 256;
  static_assert(ITEMS % 256 == 0, "fetch items are dealt in rounds of 256");
  __shared__ gemm_u32x4 S[2][3][2][TI + TJ];  // [buffer][term][k half][A rows | B rows]: 48 KiB at 128 x 128
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wi = w >> 1, wj = w & 1;
  const int m = l & 31, kk2 = l >> 5;
  const int i0 = blockIdx.x * TI, j0 = blockIdx.y * TJ;
  const int chunks = (g.K + 15) / 16, per = (chunks + gridDim.z - 1) / gridDim.z;
  const int kb = blockIdx.z * per * 16;
  const int ke = min(g.K, kb + per * 16);
  float r[NIT][8];
  auto fetch1 = [&](float (&v)[8], const float* base, long long s_row, long long s_k, int row, int n_rows, int k, bool kcont, int vec) {
    const float* p = base + (long long)row * s_row + (long long)k * s_k;
    if (row < n_rows && k + 8 <= ke) {
      if (kcont && vec) {
        const float4 u = *reinterpret_cast<const float4*>(p), x = *reinterpret_cast<const float4*>(p + 4);
        v[0] = u.x, v[1] = u.y, v[2] = u.z, v[3] = u.w, v[4] = x.x, v[5] = x.y, v[6] = x.z, v[7] = x.w;
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[q * s_k];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = (row < n_rows && k + q < ke) ? p[q * s_k] : 0.0f;
    }
  };
  // item it: A for it < 2 TI (row it % TI, k half it / TI), B after that; slot = its row in S
  auto fetch = [&](int k0) {
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
      const int it = t + 256 * n;
      if (it < 2 * TI)
        fetch1(r[n], g.a, g.sa_i, g.sa_k, i0 + it % TI, g.I, k0 + 8 * (it / TI), A_KCONT, a_vec);
      else
        fetch1(r[n], g.b, g.sb_j, g.sb_k, j0 + (it - 2 * TI) % TJ, g.J, k0 + 8 * ((it - 2 * TI) / TJ), B_KCONT, b_vec);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
      const int it = t + 256 * n;
      const int kh = it < 2 * TI ? it / TI : (it - 2 * TI) / TJ;
      const int slot = it < 2 * TI ? it % TI : TI + (it - 2 * TI) % TJ;
      gemm_u32x4 H, M, L;
      gemm_split8(r[n], H, M, L);
      S[buf][0][kh][slot] = H, S[buf][1][kh][slot] = M, S[buf][2][kh][slot] = L;
    }
  };
  f32x16 acc[BI][BJ];
#pragma unroll
  for (int bi = 0; bi < BI; ++bi)
#pragma unroll
    for (int bj = 0; bj < BJ; ++bj) acc[bi][bj] = (f32x16)(0.0f);
  if (kb < ke) {
    fetch(kb);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = kb; k0 < ke; k0 += 16, buf ^= 1) {
    const bool more = k0 + 16 < ke;
    if (more) fetch(k0 + 16);  // in flight during the matrix instructions below
    gemm_bf16x8 a[BI][3], b[BJ][3];
#pragma unroll
    for (int term = 0; term < 3; ++term) {
#pragma unroll
      for (int q = 0; q < BI; ++q) a[q][term] = __builtin_bit_cast(gemm_bf16x8, S[buf][term][kk2][wi * (TI / 2) + q * 32 + m]);
#pragma unroll
      for (int q = 0; q < BJ; ++q) b[q][term] = __builtin_bit_cast(gemm_bf16x8, S[buf][term][kk2][TI + wj * (TJ / 2) + q * 32 + m]);
    }
    // smallest terms first; the accumulators take turns, so (with more than one) no instruction has its predecessor's accumulator
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int bi = 0; bi < BI; ++bi)
#pragma unroll
        for (int bj = 0; bj < BJ; ++bj)
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[bi][TA[p]], b[bj][TB[p]], acc[bi][bj], 0, 0, 0);
    if (more) stash(buf ^ 1);  // (the other buffer was last read before the previous barrier)
    __syncthreads();
  }
#pragma unroll
  for (int bj = 0; bj < BJ; ++bj) {
    const int j = j0 + wj * (TJ / 2) + bj * 32 + m;
    if (j >= g.J) continue;
    const float bias_j = (g.bias && blockIdx.z == 0) ? g.bias[j] : 0.0f;
#pragma unroll
    for (int bi = 0; bi < BI; ++bi)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int i = i0 + wi * (TI / 2) + bi * 32 + 8 * (rr >> 2) + kk2 * 4 + (rr & 3);
        if (i >= g.I) continue;
        float* p = g.c + (long long)i * g.sc_i + j;
        const float v = acc[bi][bj][rr] + bias_j;
        if (g.mode == 0) *p = v;
        else if (g.mode == 1) *p += v;
        else atomicAdd(p, v);
      }
  }
}

// ------------------------------------------------------------------ the same product with THREE products per MAC: split-fp16
// Round 6 (the verdict's "3-product twin of gemm_b6 with per-row gains").  fp16 has 11 significand bits per term, so two terms carry
// 22 bits and three products (lo.hi, hi.lo, hi.hi) give fp32-grade results - half the matrix instructions and two thirds of the LDS
// traffic of the six-product split-bf16 form - but only 5 exponent bits: every A ROW and every B COLUMN of the tile gets its own
// power-of-two gain (largest magnitude over the workgroup's K range -> [2^14, 2^15)), found by a pre-pass of the workgroup over its
// own operands (a second read of TI x K + K x TJ values from L2; no extra launch, nothing handed over between kernels) and taken
// out again at the store.  An element more than 22 binades below its row's largest magnitude is lost, whatever its own
// magnitude: per-row / per-column scales of any spread are exact (tests/test_gemm_gpu.py), a row that holds 1e-9 next to 1e+3 is
// not - the backward passes' operands (activations, gradients of one token / one channel) are not of that kind, and the
// six-product form stays selectable (MNERF_GEMM_MATH=bf16x6).
typedef _Float16 gemm_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gemm_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int gemm_gain_exp(float m) {  // e with m 2^e in [2^14, 2^15); 15 for m = 0
  int e = __builtin_amdgcn_frexp_expf(m);
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return 15 - e;
}
__device__ __forceinline__ void gemm_split8h(const float (&v)[8], float gain, gemm_u32x4& H, gemm_u32x4& L) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const gemm_f32x2 ab = {v[2 * i] * gain, v[2 * i + 1] * gain};
    const gemm_f16x2 h = __builtin_convertvector(ab, gemm_f16x2);
    const gemm_f32x2 r = ab - __builtin_convertvector(h, gemm_f32x2);
    H[i] = __builtin_bit_cast(unsigned, h);
    L[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, gemm_f16x2));
  }
}

template <bool A_KCONT, bool B_KCONT, int TI, int TJ>
__global__ __launch_bounds__(256) void gemm_h3_kernel(GemmArgs g, int a_vec, int b_vec) {
  constexpr int BI = TI / 64, BJ = TJ / 64, ITEMS = 2 * TI + 2 * TJ, NIT = ITEMS / 256;
  static_assert(ITEMS % 256 == 0, "fetch items are dealt in rounds of 256");
  __shared__ gemm_u32x4 S[2][2][2][TI + TJ];  // [buffer][hi | lo][k half][A rows | B rows]: 32 KiB at 128 x 128
  __shared__ int rmax[TI + TJ], rexp[TI + TJ];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wi = w >> 1, wj = w & 1;
  const int m = l & 31, kk2 = l >> 5;
  const int i0 = blockIdx.x * TI, j0 = blockIdx.y * TJ;
  const int chunks = (g.K + 15) / 16, per = (chunks + gridDim.z - 1) / gridDim.z;
  const int kb = blockIdx.z * per * 16;
  const int ke = min(g.K, kb + per * 16);
  float r[NIT][8];
  auto fetch1 = [&](float (&v)[8], const float* base, long long s_row, long long s_k, int row, int n_rows, int k, bool kcont, int vec) {
    const float* p = base + (long long)row * s_row + (long long)k * s_k;
    if (row < n_rows && k + 8 <= ke) {
      if (kcont && vec) {
        const float4 u = *reinterpret_cast<const float4*>(p), x = *reinterpret_cast<const float4*>(p + 4);
        v[0] = u.x, v[1] = u.y, v[2] = u.z, v[3] = u.w, v[4] = x.x, v[5] = x.y, v[6] = x.z, v[7] = x.w;
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[q * s_k];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = (row < n_rows && k + q < ke) ? p[q * s_k] : 0.0f;
    }
  };
  auto fetch = [&](int k0) {
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
      const int it = t + 256 * n;
      if (it < 2 * TI)
        fetch1(r[n], g.a, g.sa_i, g.sa_k, i0 + it % TI, g.I, k0 + 8 * (it / TI), A_KCONT, a_vec);
      else
        fetch1(r[n], g.b, g.sb_j, g.sb_k, j0 + (it - 2 * TI) % TJ, g.J, k0 + 8 * ((it - 2 * TI) / TJ), B_KCONT, b_vec);
    }
  };
  auto slot_of = [&](int n) {
    const int it = t + 256 * n;
    return it < 2 * TI ? it % TI : TI + (it - 2 * TI) % TJ;
  };
  // ---- pre-pass: largest magnitude of every A row / B column of the tile over this workgroup's K range
  for (int q = t; q < TI + TJ; q += 256) rmax[q] = 0;
  float mx[NIT];
#pragma unroll
  for (int n = 0; n < NIT; ++n) mx[n] = 0.0f;
  for (int k0 = kb; k0 < ke; k0 += 16) {
    fetch(k0);
#pragma unroll
    for (int n = 0; n < NIT; ++n)
#pragma unroll
      for (int q = 0; q < 8; ++q) mx[n] = fmaxf(mx[n], fabsf(r[n][q]));
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NIT; ++n) atomicMax(&rmax[slot_of(n)], __float_as_int(mx[n]));  // (non-negative floats order as integers)
  __syncthreads();
  for (int q = t; q < TI + TJ; q += 256) rexp[q] = gemm_gain_exp(__int_as_float(rmax[q]));
  float gain[NIT];
#pragma unroll
  for (int n = 0; n < NIT; ++n) gain[n] = ldexpf(1.0f, gemm_gain_exp(__int_as_float(rmax[slot_of(n)])));
  auto stash = [&](int buf) {
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
      const int it = t + 256 * n;
      const int kh = it < 2 * TI ? it / TI : (it - 2 * TI) / TJ;
      gemm_u32x4 H, L;
      gemm_split8h(r[n], gain[n], H, L);
      S[buf][0][kh][slot_of(n)] = H, S[buf][1][kh][slot_of(n)] = L;
    }
  };
  f32x16 acc[BI][BJ];
#pragma unroll
  for (int bi = 0; bi < BI; ++bi)
#pragma unroll
    for (int bj = 0; bj < BJ; ++bj) acc[bi][bj] = (f32x16)(0.0f);
  if (kb < ke) {
    fetch(kb);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = kb; k0 < ke; k0 += 16, buf ^= 1) {
    const bool more = k0 + 16 < ke;
    if (more) fetch(k0 + 16);  // in flight during the matrix instructions below
    gemm_f16x8 a[BI][2], b[BJ][2];
#pragma unroll
    for (int term = 0; term < 2; ++term) {
#pragma unroll
      for (int q = 0; q < BI; ++q) a[q][term] = __builtin_bit_cast(gemm_f16x8, S[buf][term][kk2][wi * (TI / 2) + q * 32 + m]);
#pragma unroll
      for (int q = 0; q < BJ; ++q) b[q][term] = __builtin_bit_cast(gemm_f16x8, S[buf][term][kk2][TI + wj * (TJ / 2) + q * 32 + m]);
    }
    constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};  // lo.hi, hi.lo, hi.hi
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int bi = 0; bi < BI; ++bi)
#pragma unroll
        for (int bj = 0; bj < BJ; ++bj)
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[bi][TA[p]], b[bj][TB[p]], acc[bi][bj], 0, 0, 0);
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int bj = 0; bj < BJ; ++bj) {
    const int js = wj * (TJ / 2) + bj * 32 + m;
    const int j = j0 + js;
    if (j >= g.J) continue;
    const int ej = rexp[TI + js];
    const float bias_j = (g.bias && blockIdx.z == 0) ? g.bias[j] : 0.0f;
#pragma unroll
    for (int bi = 0; bi < BI; ++bi)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int is = wi * (TI / 2) + bi * 32 + 8 * (rr >> 2) + kk2 * 4 + (rr & 3);
        const int i = i0 + is;
        if (i >= g.I) continue;
        float* p = g.c + (long long)i * g.sc_i + j;
        const float v = ldexpf(acc[bi][bj][rr], -(rexp[is] + ej)) + bias_j;
        if (g.mode == 0) *p = v;
        else if (g.mode == 1) *p += v;
        else atomicAdd(p, v);
      }
  }
}

#ifndef MNERF_GEMM_TILE128
#define MNERF_GEMM_TILE128 1  // 0: every product through the 64 x 64 kernel (round 3)
#endif
#ifndef MNERF_GEMM_DEFAULT_MATH
#define MNERF_GEMM_DEFAULT_MATH 1
#endif
// MNERF_GEMM_MATH (environment, read once): "bf16x6" (default) = split-bf16, "f16x3" = split-fp16 with row gains (round 6: half the
// matrix instructions, but its pre-pass reads the operands a second time and these products are bound by operand fetch and split,
// not by the matrix pipe: 35.6 against 34.5 ms per training iteration - kept selectable and tested, not the default), both on the
// 16-bit matrix pipe for products with I, J >= 128; "f32" = the exact-f32 matrix instruction everywhere
static int gemm_math_b6() {  // 0: exact f32, 1: split-bf16 (six products), 2: split-fp16 with row gains (three products)
  static const int v = [] {
    const char* e = getenv("MNERF_GEMM_MATH");
    if (e && !strcmp(e, "f32")) return 0;
    if (e && !strcmp(e, "bf16x6")) return 1;
    if (e && !strcmp(e, "f16x3")) return 2;
    return MNERF_GEMM_DEFAULT_MATH;
  }();
  return v;
}
static void gemm_with(hipStream_t st, const float* a, long long sa_i, long long sa_k, const float* b, long long sb_k, long long sb_j,
                      float* c, long long sc_i, const float* bias, int I, int J, int K, int mode, int math_b6) {
  GemmArgs g{a, sa_i, sa_k, b, sb_k, sb_j, c, sc_i, bias, I, J, K, mode};
  const bool ak = sa_k == 1, bk = sb_k == 1;
  if (math_b6 && I >= 128 && J >= 128) {
    // split-bf16: 128 x 128 tiles where they fill the chip twice, 64 x 64 ones below that; a reduction over the rows (mode 2)
    // is split into parts of at least 256 until ~1 000 workgroups exist
    const long long t128 = (long long)((I + 127) / 128) * ((J + 127) / 128);
    const int T = t128 >= 512 ? 128 : 64;
    const int ti = (I + T - 1) / T, tj = (J + T - 1) / T;
    int split = 1;
    if (mode == 2) {
      // measured (N = 30 720 rows, [N,128]^T [N,128]): parts of >= 256 rows 29.4 us, >= 128 36.4, >= 512 31.2; 128-tiles 40.6
      split = 1024 / (ti * tj);
      const int max_split = (K + 255) / 256;
      if (split > max_split) split = max_split;
      if (split < 1) split = 1;
    }
    const dim3 grid(ti, tj, split);
    const int a_vec = ak && sa_i % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0;
    const int b_vec = bk && sb_j % 4 == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0;
#define GEMM_B6_LAUNCH(AK_, BK_)                                                                                              \
  do {                                                                                                                        \
    if (math_b6 == 2) {                                                                                                       \
      if (T == 128) hipLaunchKernelGGL((gemm_h3_kernel<AK_, BK_, 128, 128>), grid, dim3(256), 0, st, g, a_vec, b_vec);         \
      else hipLaunchKernelGGL((gemm_h3_kernel<AK_, BK_, 64, 64>), grid, dim3(256), 0, st, g, a_vec, b_vec);                     \
    } else if (T == 128) hipLaunchKernelGGL((gemm_b6_kernel<AK_, BK_, 128, 128>), grid, dim3(256), 0, st, g, a_vec, b_vec);    \
    else hipLaunchKernelGGL((gemm_b6_kernel<AK_, BK_, 64, 64>), grid, dim3(256), 0, st, g, a_vec, b_vec);                       \
  } while (0)
    if (ak && bk) GEMM_B6_LAUNCH(true, true);
    else if (ak) GEMM_B6_LAUNCH(true, false);
    else if (bk) GEMM_B6_LAUNCH(false, true);
    else GEMM_B6_LAUNCH(false, false);
#undef GEMM_B6_LAUNCH
    return;
  }
  if (MNERF_GEMM_TILE128 && I >= 128 && J >= 128) {  // both tile dimensions at least half used on average
    const int ti = (I + 127) / 128, tj = (J + 127) / 128;
    int split = 1;
    if (mode == 2) {
      split = 512 / (ti * tj);
      const int max_split = (K + 511) / 512;
      if (split > max_split) split = max_split;
      if (split < 1) split = 1;
    }
    const dim3 grid(ti, tj, split);
    if (ak && bk) hipLaunchKernelGGL((gemm_f32_tile128_kernel<true, true>), grid, dim3(256), 0, st, g);
    else if (ak) hipLaunchKernelGGL((gemm_f32_tile128_kernel<true, false>), grid, dim3(256), 0, st, g);
    else if (bk) hipLaunchKernelGGL((gemm_f32_tile128_kernel<false, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_tile128_kernel<false, false>), grid, dim3(256), 0, st, g);
    return;
  }
  const int ti = (I + 63) / 64, tj = (J + 63) / 64;
  int split = 1;
  if (mode == 2) {  // the reduction runs over the samples: enough splits to fill the chip, at least 256 samples each
    split = 1024 / (ti * tj);  // (512 measured slower: 3.1 vs 2.9 ms per pass at 65 536 samples)
    const int max_split = (K + 255) / 256;
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
  }
  const dim3 grid(ti, tj, split);
  if (ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, g);
  else if (ak) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, g);
  else if (bk) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, g);
}
static void gemm(hipStream_t st, const float* a, long long sa_i, long long sa_k, const float* b, long long sb_k, long long sb_j,
                 float* c, long long sc_i, const float* bias, int I, int J, int K, int mode) {
  gemm_with(st, a, sa_i, sa_k, b, sb_k, sb_j, c, sc_i, bias, I, J, K, mode, gemm_math_b6());
}
// Y[N,M] = X[N,K] W[M,K]^T + bias          (torch Linear; ldx / ldw / ldy = row strides)
static void linear_fwd(hipStream_t st, const float* x, long long ldx, const float* w, long long ldw, const float* bias, float* y,
                       long long ldy, int N, int M, int K, bool add = false) {
  gemm(st, x, ldx, 1, w, 1, ldw, y, ldy, bias, N, M, K, add ? 1 : 0);
}
// dX[N,K] (+)= dY[N,M] W[M,K]
static void linear_bwd_data(hipStream_t st, const float* dy, long long lddy, const float* w, long long ldw, float* dx, long long lddx,
                            int N, int M, int K, bool add) {
  gemm(st, dy, lddy, 1, w, ldw, 1, dx, lddx, nullptr, N, K, M, add ? 1 : 0);
}
// dW[M,K] += dY[N,M]^T X[N,K]
static void linear_bwd_weight(hipStream_t st, const float* dy, long long lddy, const float* x, long long ldx, float* dw, long long lddw,
                              int N, int M, int K) {
  if (dw) gemm(st, dy, 1, lddy, x, ldx, 1, dw, lddw, nullptr, M, K, N, 2);
}

// ------------------------------------------------------------------ elementwise pieces
// out[j] += sum_n x[n, j]   (bias gradients, LayerNorm gain / shift gradients); M <= 128
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long ldx, int N, int M, float* __restrict__ out) {
  __shared__ float part[256];
  const int rows_per_pass = 256 / M;
  const int j = threadIdx.x % M, lane_row = threadIdx.x / M;
  float s = 0.0f;
  if (lane_row < rows_per_pass)
    for (long long n = (long long)blockIdx.x * rows_per_pass + lane_row; n < N; n += (long long)gridDim.x * rows_per_pass) s += x[n * ldx + j];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < M) {
    float tot = 0.0f;
    for (int r = 0; r < rows_per_pass; ++r) tot += part[r * M + threadIdx.x];
    atomicAdd(out + threadIdx.x, tot);
  }
}
// The same for M % 4 == 0, 16-byte aligned rows (round 6): a thread owns FOUR columns (one 16-byte load per row instead of four
// 4-byte ones), four rows in flight per thread - the first version walked N / 512 rows per thread with one
// dependent load each: 23.5 us for [65 536, 128], 43.7 us for [131 072, 128] (1.5 TB/s).
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ x, long long ldx, int N, int M, float* __restrict__ out) {
  __shared__ float4 part[256];
  const int m4 = M >> 2, rows_per_pass = 256 / m4;
  const int j4 = threadIdx.x % m4, lane_row = threadIdx.x / m4;
  float4 s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane_row < rows_per_pass) {
    const long long step = (long long)gridDim.x * rows_per_pass;
    long long n = (long long)blockIdx.x * rows_per_pass + lane_row;
    for (; n + 3 * step < N; n += 4 * step) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = *reinterpret_cast<const float4*>(x + (n + u * step) * ldx + 4 * j4);
        s[u].x += v.x, s[u].y += v.y, s[u].z += v.z, s[u].w += v.w;
      }
    }
    for (; n < N; n += step) {
      const float4 v = *reinterpret_cast<const float4*>(x + n * ldx + 4 * j4);
      s[0].x += v.x, s[0].y += v.y, s[0].z += v.z, s[0].w += v.w;
    }
  }
  part[threadIdx.x] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                                  (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
  __syncthreads();
  if (threadIdx.x < m4) {
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < rows_per_pass; ++r) {
      const float4 v = part[r * m4 + threadIdx.x];
      tot.x += v.x, tot.y += v.y, tot.z += v.z, tot.w += v.w;
    }
    atomicAdd(out + 4 * threadIdx.x + 0, tot.x), atomicAdd(out + 4 * threadIdx.x + 1, tot.y);
    atomicAdd(out + 4 * threadIdx.x + 2, tot.z), atomicAdd(out + 4 * threadIdx.x + 3, tot.w);
  }
}
static void colsum(hipStream_t st, const float* x, long long ldx, int N, int M, float* out) {
  if (!out) return;
  if (M % 4 == 0 && M >= 4 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int rows_per_pass = 256 / (M / 4);
    // (every workgroup ends with M atomics on the same M addresses: 2 048 workgroups made the kernel SLOWER than the first version,
    // 73 against 44 us; one workgroup per CU)
    long long blocks = ((long long)N + rows_per_pass * 16 - 1) / (rows_per_pass * 16);  // ~16 rows per thread
    blocks = blocks < 64 ? 64 : (blocks > 256 ? 256 : blocks);
    hipLaunchKernelGGL(colsum4_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx, N, M, out);
    return;
  }
  hipLaunchKernelGGL(colsum_kernel<0>, dim3(256), dim3(256), 0, st, x, ldx, N, M, out);
}

