// Micro-test for the split-bf16 ("bf16x6") matrix path on gfx950:
//  1. operand layout of v_mfma_f32_32x32x16_bf16 (A row = lane%32, k = 8*(lane/32)+j; B col = lane%32, same k;
//     C/D as the f32 32x32 layout) — checked numerically against a double-precision host product;
//  2. accuracy of 3-way bf16 splitting with 6 product terms vs the exact-f32 MFMA;
//  3. issue rate of a layer-shaped loop (8 K16-steps x 4 M-blocks x 6 MFMAs, A parts from LDS, B split from
//     accumulator-layout registers) with two workgroups per CU.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, r);
}

struct Parts { bf16x8 hi, mid, lo; };

__device__ __forceinline__ Parts split8(const float* v) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    h[i] = pk_bf16(a, b);
    const float ra = a - __uint_as_float(h[i] << 16), rb = b - __uint_as_float(h[i] & 0xffff0000u);
    m[i] = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m[i] << 16), sb = rb - __uint_as_float(m[i] & 0xffff0000u);
    l[i] = pk_bf16(sa, sb);
  }
  Parts p;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 H = {h[0], h[1], h[2], h[3]}, M = {m[0], m[1], m[2], m[3]}, L = {l[0], l[1], l[2], l[3]};
  p.hi = __builtin_bit_cast(bf16x8, H);
  p.mid = __builtin_bit_cast(bf16x8, M);
  p.lo = __builtin_bit_cast(bf16x8, L);
  return p;
}

__device__ __forceinline__ f32x16 mm(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// weights in LDS: [step t (8)][block m (4)][part p (3)][lane (64)][8 bf16] = 16 B per lane
#define LAYER_BYTES (8 * 4 * 3 * 64 * 16)
#define LAYER_PAD (6 * 1024)

template <int MODE>
__device__ __forceinline__ void layer(f32x16 (&acc)[4], const f32x16 (&h)[4], const char* wl, int lane) {
  // A parts of (step, block) i = 4t+m are double-buffered: the three ds_read_b128 of i+1 are issued before
  // the six MFMAs of i (MODE 0), or of the block PAIR i..i+1 before the twelve MFMAs of the pair (MODE 1).
  const bf16x8* a = reinterpret_cast<const bf16x8*>(wl + lane * 16);
  if (MODE == 0) {
    bf16x8 ch = a[0], cm = a[64], cl = a[128];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = h[t >> 1][(t & 1) * 8 + j];
      const Parts b = split8(v);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int nx = (t * 4 + m + 1) * 3 * 64;  // last one reads one fragment past the layer (padding)
        const bf16x8 nh = a[nx], nm = a[nx + 64], nl = a[nx + 128];
        __builtin_amdgcn_sched_barrier(0);
        acc[m] = mm(ch, b.lo, acc[m]);
        acc[m] = mm(cl, b.hi, acc[m]);
        acc[m] = mm(cm, b.mid, acc[m]);
        acc[m] = mm(ch, b.mid, acc[m]);
        acc[m] = mm(cm, b.hi, acc[m]);
        acc[m] = mm(ch, b.hi, acc[m]);
        __builtin_amdgcn_sched_barrier(0);
        ch = nh; cm = nm; cl = nl;
      }
    }
  } else if (MODE == 2 || MODE == 3) {
    // split of step t+1 interleaved between the MFMAs of (t, block 0): MFMA, <=8 VALU, MFMA, ...
    bf16x8 ch = a[0], cm = a[64], cl = a[128];
    float v0[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v0[j] = h[0][j];
    Parts b = split8(v0);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      Parts bn = b;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int nx = (t * 4 + m + 1) * 3 * 64;
        const bf16x8 nh = a[nx], nm = a[nx + 64], nl = a[nx + 128];
        __builtin_amdgcn_sched_barrier(0);
        acc[m] = mm(ch, b.lo, acc[m]);
        acc[m] = mm(cl, b.hi, acc[m]);
        acc[m] = mm(cm, b.mid, acc[m]);
        acc[m] = mm(ch, b.mid, acc[m]);
        acc[m] = mm(cm, b.hi, acc[m]);
        acc[m] = mm(ch, b.hi, acc[m]);
        if (m == 0 && t < 7) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = h[(t + 1) >> 1][((t + 1) & 1) * 8 + j];
          bn = split8(v);
          {  // pin the split here (IR-level sinking would otherwise move it next to its first use)
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            u32x4_ H = __builtin_bit_cast(u32x4_, bn.hi), M = __builtin_bit_cast(u32x4_, bn.mid), L = __builtin_bit_cast(u32x4_, bn.lo);
            asm volatile("" : "+v"(H), "+v"(M), "+v"(L));
            bn.hi = __builtin_bit_cast(bf16x8, H); bn.mid = __builtin_bit_cast(bf16x8, M); bn.lo = __builtin_bit_cast(bf16x8, L);
          }
          if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
              __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // then up to 8 VALU
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        ch = nh; cm = nm; cl = nl;
      }
      b = bn;
    }
  } else {
    bf16x8 ch0 = a[0], cm0 = a[64], cl0 = a[128], ch1 = a[192], cm1 = a[256], cl1 = a[320];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = h[t >> 1][(t & 1) * 8 + j];
      const Parts b = split8(v);
#pragma unroll
      for (int mp = 0; mp < 4; mp += 2) {
        const int nx = (t * 4 + mp + 2) * 3 * 64;
        const bf16x8 nh0 = a[nx], nm0 = a[nx + 64], nl0 = a[nx + 128], nh1 = a[nx + 192], nm1 = a[nx + 256], nl1 = a[nx + 320];
        __builtin_amdgcn_sched_barrier(0);
        acc[mp] = mm(ch0, b.lo, acc[mp]);
        acc[mp + 1] = mm(ch1, b.lo, acc[mp + 1]);
        acc[mp] = mm(cl0, b.hi, acc[mp]);
        acc[mp + 1] = mm(cl1, b.hi, acc[mp + 1]);
        acc[mp] = mm(cm0, b.mid, acc[mp]);
        acc[mp + 1] = mm(cm1, b.mid, acc[mp + 1]);
        acc[mp] = mm(ch0, b.mid, acc[mp]);
        acc[mp + 1] = mm(ch1, b.mid, acc[mp + 1]);
        acc[mp] = mm(cm0, b.hi, acc[mp]);
        acc[mp + 1] = mm(cm1, b.hi, acc[mp + 1]);
        acc[mp] = mm(ch0, b.hi, acc[mp]);
        acc[mp + 1] = mm(ch1, b.hi, acc[mp + 1]);
        __builtin_amdgcn_sched_barrier(0);
        ch0 = nh0; cm0 = nm0; cl0 = nl0; ch1 = nh1; cm1 = nm1; cl1 = nl1;
      }
    }
  }
}

// numerics: one wave, Y^T[128 x 32] = W[128x128] . H^T[128 x 32]; H given in accumulator layout
__global__ void check_kernel(const char* wpk, const float* hin, float* yout) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < LAYER_BYTES / 16; i += 64) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(wpk)[i];
  __syncthreads();
  f32x16 h[4], acc[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      h[m][r] = hin[f * 32 + (lane & 31)];
      acc[m][r] = 0.f;
    }
  layer<0>(acc, h, lds, lane);
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) {
      const int o = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      yout[o * 32 + (lane & 31)] = acc[m][r];
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void rate_kernel(const char* wpk, float* sink, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < LAYER_BYTES / 16; i += 256) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(wpk)[i];
  __syncthreads();
  f32x16 h[4], acc[4], film[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) { h[m][r] = 0.01f * (lane + r + m); film[m][r] = 1.0f + 0.001f * r; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.f);
    layer<MODE>(acc, h, lds, lane);
    for (int m = 0; m < 4; ++m)
      for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f) * 1e-3f + 0.01f;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += h[m][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// Two independent 32-sample groups per wave, ONE wave per SIMD (512 registers): the A fragments of a (step, block)
// unit are read once and feed 12 MFMAs (6 per group, interleaved => no two consecutive MFMAs depend on each other);
// the operand split of one group issues under the MFMAs of the other.
__global__ __launch_bounds__(256, 1) void rate2_kernel(const char* wpk, float* sink, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < LAYER_BYTES / 16; i += 256) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(wpk)[i];
  __syncthreads();
  f32x16 h0[4], h1[4], a0[4], a1[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) { h0[m][r] = 0.01f * (lane + r + m); h1[m][r] = 0.02f * (lane + r + 2 * m); }
  const bf16x8* a = reinterpret_cast<const bf16x8*>(lds + lane * 16);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    for (int m = 0; m < 4; ++m) { a0[m] = (f32x16)(0.f); a1[m] = (f32x16)(0.f); }
    bf16x8 ch = a[0], cm = a[64], cl = a[128];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v0[8], v1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v0[j] = h0[t >> 1][(t & 1) * 8 + j]; v1[j] = h1[t >> 1][(t & 1) * 8 + j]; }
      const Parts b0 = split8(v0), b1 = split8(v1);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int nx = (t * 4 + m + 1) * 3 * 64;
        const bf16x8 nh = a[nx], nm = a[nx + 64], nl = a[nx + 128];
        __builtin_amdgcn_sched_barrier(0);
        a0[m] = mm(ch, b0.lo, a0[m]);  a1[m] = mm(ch, b1.lo, a1[m]);
        a0[m] = mm(cl, b0.hi, a0[m]);  a1[m] = mm(cl, b1.hi, a1[m]);
        a0[m] = mm(cm, b0.mid, a0[m]); a1[m] = mm(cm, b1.mid, a1[m]);
        a0[m] = mm(ch, b0.mid, a0[m]); a1[m] = mm(ch, b1.mid, a1[m]);
        a0[m] = mm(cm, b0.hi, a0[m]);  a1[m] = mm(cm, b1.hi, a1[m]);
        a0[m] = mm(ch, b0.hi, a0[m]);  a1[m] = mm(ch, b1.hi, a1[m]);
        __builtin_amdgcn_sched_barrier(0);
        ch = nh; cm = nm; cl = nl;
      }
    }
    for (int m = 0; m < 4; ++m)
      for (int r = 0; r < 16; ++r) {
        h0[m][r] = fmaxf(a0[m][r] * 1.001f, 0.0f) * 1e-3f + 0.01f;
        h1[m][r] = fmaxf(a1[m][r] * 1.002f, 0.0f) * 1e-3f + 0.02f;
      }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += h0[m][r] + h1[m][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// Isolation variants of the layer loop (two workgroups per CU): WITH_LDS = A fragments re-read from LDS per unit
// (else held in registers), WITH_SPLIT = operands split per step (else split once).
template <int WITH_LDS, int WITH_SPLIT>
__global__ __launch_bounds__(256, 2) void iso_kernel(const char* wpk, float* sink, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < LAYER_BYTES / 16; i += 256) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(wpk)[i];
  __syncthreads();
  f32x16 h[4], acc[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) h[m][r] = 0.01f * (lane + r + m);
  const bf16x8* a = reinterpret_cast<const bf16x8*>(lds + lane * 16);
  bf16x8 ch = a[0], cm = a[64], cl = a[128];
  float v0[8];
  for (int j = 0; j < 8; ++j) v0[j] = h[0][j];
  Parts b = split8(v0);
  for (int it = 0; it < iters; ++it) {
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.f);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (WITH_SPLIT) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = h[t >> 1][(t & 1) * 8 + j];
        b = split8(v);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        bf16x8 nh = ch, nm = cm, nl = cl;
        if (WITH_LDS) {
          const int nx = (t * 4 + m + 1) * 3 * 64;
          nh = a[nx]; nm = a[nx + 64]; nl = a[nx + 128];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[m] = mm(ch, b.lo, acc[m]);
        acc[m] = mm(cl, b.hi, acc[m]);
        acc[m] = mm(cm, b.mid, acc[m]);
        acc[m] = mm(ch, b.mid, acc[m]);
        acc[m] = mm(cm, b.hi, acc[m]);
        acc[m] = mm(ch, b.hi, acc[m]);
        __builtin_amdgcn_sched_barrier(0);
        ch = nh; cm = nm; cl = nl;
      }
    }
    for (int m = 0; m < 4; ++m)
      for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * 1.001f, 0.0f) * 1e-3f + 0.01f;
  }
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += h[m][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// Same loop as iso_kernel<1,0> but with the accumulators pinned in AGPRs (inline-asm MFMA with "+a" operands):
// does the accumulator register class matter for the issue rate?
__device__ __forceinline__ void mm_agpr(f32x16& acc, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__global__ __launch_bounds__(256, 2) void agpr_kernel(const char* wpk, float* sink, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < LAYER_BYTES / 16; i += 256) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(wpk)[i];
  __syncthreads();
  f32x16 h[4], acc[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) h[m][r] = 0.01f * (lane + r + m);
  const bf16x8* a = reinterpret_cast<const bf16x8*>(lds + lane * 16);
  bf16x8 ch = a[0], cm = a[64], cl = a[128];
  float v0[8];
  for (int j = 0; j < 8; ++j) v0[j] = h[0][j];
  const Parts b = split8(v0);
  for (int it = 0; it < iters; ++it) {
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.f);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int nx = (t * 4 + m + 1) * 3 * 64;
        const bf16x8 nh = a[nx], nm = a[nx + 64], nl = a[nx + 128];
        __builtin_amdgcn_sched_barrier(0);
        mm_agpr(acc[m], ch, b.lo);
        mm_agpr(acc[m], cl, b.hi);
        mm_agpr(acc[m], cm, b.mid);
        mm_agpr(acc[m], ch, b.mid);
        mm_agpr(acc[m], cm, b.hi);
        mm_agpr(acc[m], ch, b.hi);
        __builtin_amdgcn_sched_barrier(0);
        ch = nh; cm = nm; cl = nl;
      }
    }
    for (int m = 0; m < 4; ++m)
      for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * 1.001f, 0.0f) * 1e-3f + 0.01f;
  }
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += h[m][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// iso_kernel<1,0> with the MFMAs of two output blocks interleaved (no two consecutive MFMAs on one accumulator)
__global__ __launch_bounds__(256, 2) void iso2_kernel(const char* wpk, float* sink, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < LAYER_BYTES / 16; i += 256) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(wpk)[i];
  __syncthreads();
  f32x16 h[4], acc[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) h[m][r] = 0.01f * (lane + r + m);
  const bf16x8* a = reinterpret_cast<const bf16x8*>(lds + lane * 16);
  bf16x8 ch0 = a[0], cm0 = a[64], cl0 = a[128], ch1 = a[192], cm1 = a[256], cl1 = a[320];
  float v0[8];
  for (int j = 0; j < 8; ++j) v0[j] = h[0][j];
  const Parts b = split8(v0);
  for (int it = 0; it < iters; ++it) {
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.f);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int mp = 0; mp < 4; mp += 2) {
        const int nx = (t * 4 + mp + 2) * 3 * 64;
        const bf16x8 nh0 = a[nx], nm0 = a[nx + 64], nl0 = a[nx + 128], nh1 = a[nx + 192], nm1 = a[nx + 256], nl1 = a[nx + 320];
        __builtin_amdgcn_sched_barrier(0);
        acc[mp] = mm(ch0, b.lo, acc[mp]);   acc[mp + 1] = mm(ch1, b.lo, acc[mp + 1]);
        acc[mp] = mm(cl0, b.hi, acc[mp]);   acc[mp + 1] = mm(cl1, b.hi, acc[mp + 1]);
        acc[mp] = mm(cm0, b.mid, acc[mp]);  acc[mp + 1] = mm(cm1, b.mid, acc[mp + 1]);
        acc[mp] = mm(ch0, b.mid, acc[mp]);  acc[mp + 1] = mm(ch1, b.mid, acc[mp + 1]);
        acc[mp] = mm(cm0, b.hi, acc[mp]);   acc[mp + 1] = mm(cm1, b.hi, acc[mp + 1]);
        acc[mp] = mm(ch0, b.hi, acc[mp]);   acc[mp + 1] = mm(ch1, b.hi, acc[mp + 1]);
        __builtin_amdgcn_sched_barrier(0);
        ch0 = nh0; cm0 = nm0; cl0 = nl0; ch1 = nh1; cm1 = nm1; cl1 = nl1;
      }
    }
    for (int m = 0; m < 4; ++m)
      for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * 1.001f, 0.0f) * 1e-3f + 0.01f;
  }
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += h[m][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// Pure issue-rate ceiling: register-resident operands, loop-carried accumulators (nothing to hoist), no LDS, no VALU.
__global__ __launch_bounds__(256, 2) void pure_kernel(const char* wpk, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  const bf16x8* a = reinterpret_cast<const bf16x8*>(wpk) + lane;
  bf16x8 A[6];
  for (int i = 0; i < 6; ++i) A[i] = a[i * 64];
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48; ++u) acc[u & 3] = mm(A[u % 6], A[(u + 1) % 6], acc[u & 3]);
  }
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void pure16_kernel(const char* wpk, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  const bf16x8* a = reinterpret_cast<const bf16x8*>(wpk) + lane;
  bf16x8 A[6];
  for (int i = 0; i < 6; ++i) A[i] = a[i * 64];
  f32x4v acc[8];
  for (int m = 0; m < 8; ++m) acc[m] = (f32x4v)(0.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[u % 6], A[(u + 1) % 6], acc[u & 7], 0, 0, 0);
  }
  float s = 0.f;
  for (int m = 0; m < 8; ++m) for (int r = 0; r < 4; ++r) s += acc[m][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

static unsigned short bf16_rn(float f) {
  unsigned u; memcpy(&u, &f, 4);
  unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(r >> 16);
}
static float bf16_f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  const int K = 128, M = 128;
  std::vector<float> W(M * K), H(K * 32);
  srand(1);
  for (auto& w : W) w = ((rand() / (float)RAND_MAX) - 0.5f) * 0.4f;
  for (auto& h : H) h = ((rand() / (float)RAND_MAX)) * 3.0f * ((rand() & 7) ? 1.f : 0.f);
  // pack: step t, block m, part p, lane l, j: row = 32m + l%32 ; k-feature = regorder(8t+j) + 4*(l/32)
  std::vector<unsigned short> pk(LAYER_BYTES / 2);
  for (int t = 0; t < 8; ++t)
    for (int m = 0; m < 4; ++m)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int reg = 8 * t + j, blk = reg >> 4, r = reg & 15;
          const int f = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
          const float w = W[(32 * m + (l & 31)) * K + f];
          const unsigned short hi = bf16_rn(w);
          const float r1 = w - bf16_f(hi);
          const unsigned short mid = bf16_rn(r1);
          const float r2 = r1 - bf16_f(mid);
          const unsigned short lo = bf16_rn(r2);
          const size_t base = ((size_t)((t * 4 + m) * 3) * 64 + l) * 8 + j;
          pk[base] = hi; pk[base + 64 * 8] = mid; pk[base + 2 * 64 * 8] = lo;
        }
  char* d_w; float *d_h, *d_y, *d_sink; long long* d_c;
  hipMalloc(&d_w, LAYER_BYTES); hipMalloc(&d_h, K * 32 * 4); hipMalloc(&d_y, M * 32 * 4);
  hipMalloc(&d_sink, 512 * 256 * 4); hipMalloc(&d_c, 8);
  hipMemcpy(d_w, pk.data(), LAYER_BYTES, hipMemcpyHostToDevice);
  hipMemcpy(d_h, H.data(), K * 32 * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)check_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LAYER_BYTES + LAYER_PAD);
  hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), LAYER_BYTES + LAYER_PAD, 0, d_w, d_h, d_y);
  std::vector<float> Y(M * 32);
  hipMemcpy(Y.data(), d_y, M * 32 * 4, hipMemcpyDeviceToHost);
  double max_err = 0, max_err32 = 0, max_ref = 0;
  for (int o = 0; o < M; ++o)
    for (int n = 0; n < 32; ++n) {
      double ref = 0; float f32 = 0.f;
      for (int k = 0; k < K; ++k) { ref += (double)W[o * K + k] * (double)H[k * 32 + n]; f32 = fmaf(W[o * K + k], H[k * 32 + n], f32); }
      max_err = fmax(max_err, fabs(Y[o * 32 + n] - ref));
      max_err32 = fmax(max_err32, fabs((double)f32 - ref));
      max_ref = fmax(max_ref, fabs(ref));
    }
  printf("numerics: max|bf16x6 - exact| = %.3e   max|f32 fma chain - exact| = %.3e   max|ref| = %.3f\n", max_err, max_err32, max_ref);

  const int lds_rate = 76 * 1024;  // >= LAYER_BYTES + LAYER_PAD  // same LDS footprint as the decoder => 2 workgroups per CU
  hipFuncSetAttribute((const void*)rate_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
  hipFuncSetAttribute((const void*)rate_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
  hipFuncSetAttribute((const void*)rate_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
  hipFuncSetAttribute((const void*)rate_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
  for (int mode = 0; mode < 4; ++mode)
    for (int grid : {256, 512}) {
      const int iters = 2000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
        else if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
        else if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
        else hipLaunchKernelGGL(rate_kernel<3>, dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
      const double mfma = (double)iters * 192;  // per wave
      const double useful = 2.0 * 128 * 128 * 32 * (double)iters * grid * 4;  // f32-equivalent flops
      printf("mode %d grid %d: %.3f ms, %.1f memtime ticks per MFMA (wave 0), f32-equivalent %.1f TFLOP/s\n", mode, grid, ms,
             (double)c / mfma, useful / (ms * 1e-3) / 1e12);
    }
  for (int v = 1; v < 4; v += 2) {  // (variants without the LDS reads are loop-invariant and get hoisted: not run)
    const int iters = 2000, grid = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    hipFuncSetAttribute((const void*)iso_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
    hipFuncSetAttribute((const void*)iso_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
    hipFuncSetAttribute((const void*)iso_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
    hipFuncSetAttribute((const void*)iso_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL((iso_kernel<0, 0>), dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
      if (v == 1) hipLaunchKernelGGL((iso_kernel<1, 0>), dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
      if (v == 2) hipLaunchKernelGGL((iso_kernel<0, 1>), dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
      if (v == 3) hipLaunchKernelGGL((iso_kernel<1, 1>), dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double useful = 2.0 * 128 * 128 * 32 * (double)iters * grid * 4;
    printf("isolation: LDS reads %d, per-step split %d: %.3f ms, f32-equivalent %.1f TFLOP/s (%.0f %% of 2.5 PF/6)\n", v & 1, v >> 1, ms,
           useful / (ms * 1e-3) / 1e12, 100.0 * useful / (ms * 1e-3) / 1e12 / (2500.0 / 6));
  }
  for (int grid : {256, 512}) {
    const int iters = 8000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(pure_kernel, dim3(grid), dim3(256), 0, 0, d_w, d_sink, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = 2.0 * 32 * 32 * 16 * 48.0 * iters * grid * 4;
    printf("pure MFMA stream (registers only), grid %d: %.3f ms, %.0f bf16 TFLOP/s = %.0f %% of 2.5 PF\n", grid, ms, flops / (ms * 1e-3) / 1e12,
           100.0 * flops / (ms * 1e-3) / 1e12 / 2500.0);
  }
  for (int grid : {256, 512, 1024}) {
    const int iters = 16000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(pure16_kernel, dim3(grid), dim3(256), 0, 0, d_w, d_sink, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = 2.0 * 16 * 16 * 32 * 48.0 * iters * grid * 4;
    printf("pure 16x16x32 bf16 MFMA stream, grid %d: %.3f ms, %.0f bf16 TFLOP/s = %.0f %% of 2.5 PF\n", grid, ms, flops / (ms * 1e-3) / 1e12,
           100.0 * flops / (ms * 1e-3) / 1e12 / 2500.0);
  }
  for (int grid : {256, 512}) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)iso2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(iso2_kernel, dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double useful = 2.0 * 128 * 128 * 32 * (double)iters * grid * 4;
    printf("two blocks interleaved, no per-step split, grid %d: %.3f ms, f32-equivalent %.1f TFLOP/s (%.0f %% of 2.5 PF/6)\n", grid, ms,
           useful / (ms * 1e-3) / 1e12, 100.0 * useful / (ms * 1e-3) / 1e12 / (2500.0 / 6));
  }
  {
    const int iters = 2000, grid = 512;
    hipFuncSetAttribute((const void*)agpr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_rate);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(agpr_kernel, dim3(grid), dim3(256), lds_rate, 0, d_w, d_sink, iters, d_c);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double useful = 2.0 * 128 * 128 * 32 * (double)iters * grid * 4;
    printf("accumulators in AGPRs (LDS reads, no per-step split): %.3f ms, f32-equivalent %.1f TFLOP/s (%.0f %% of 2.5 PF/6)\n", ms,
           useful / (ms * 1e-3) / 1e12, 100.0 * useful / (ms * 1e-3) / 1e12 / (2500.0 / 6));
  }
  {
    const int lds2 = 100 * 1024;  // > 80 KiB: one workgroup per CU
    hipFuncSetAttribute((const void*)rate2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(rate2_kernel, dim3(grid), dim3(256), lds2, 0, d_w, d_sink, iters, d_c);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double useful = 2.0 * 128 * 128 * 64 * (double)iters * grid * 4;
    printf("two groups per wave, 1 wave/SIMD, grid %d: %.3f ms, f32-equivalent %.1f TFLOP/s\n", grid, ms, useful / (ms * 1e-3) / 1e12);
  }
  return 0;
}
