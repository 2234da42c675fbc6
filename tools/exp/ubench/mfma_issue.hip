// Microbenchmark for DESIGN.md section 9: how close to back-to-back can a wave issue v_mfma_f32_32x32x16_f16 (32 cycles each)
//   mode 0: one wave per SIMD, matrix phase alone: 96 MFMAs (4 accumulators x 8 K-steps x 3 products), A fragments from LDS two
//           units ahead (the decoder's ksteps_presplit2), B from registers
//   mode 1: two waves per SIMD in ping-pong: team A runs the matrix phase while team B runs NV vector instructions, s_barrier, swap
//   mode 2: one wave per SIMD, the matrix phase of one sample group and the vector phase of the other in ONE instruction stream,
//           interleaved with sched_group_barrier (1 MFMA : FILL vector instructions)
//   mode 3: like 0 but with the vector phase after the matrix phase in the same wave (serial: the no-overlap reference)
// prints cycles per slot (s_memtime) averaged over the workgroup's wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef NV
#define NV 448  // vector instructions of a vector phase (the decoder's: ~400-500)
#endif
#ifndef FILL
#define FILL 4
#endif
#define REPS 192

__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void glds16_s(const float* uniform_src, unsigned lane_off_bytes, unsigned lds_byte_addr);
struct Frag {
  f16x8 hi[8], lo[8];
};

// 96 MFMAs: unit i = (K-step u, block m); fragments two units ahead
template <int NDMA>
__device__ __forceinline__ void matrix_phase_dma(f32x16 (&acc)[4], const u32x4* a_lds, int lane, const Frag& b, const float* src,
                                                 unsigned dst_lds) {
  u32x4 fh[3], fl[3];
  const u32x4* a = a_lds + lane;
  const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
  for (int i = 0; i < 2; ++i) fh[i] = a[i * 128], fl[i] = a[i * 128 + 64];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    if (i + 2 < 32) fh[(i + 2) % 3] = a[(i + 2) * 128], fl[(i + 2) % 3] = a[(i + 2) * 128 + 64];
    __builtin_amdgcn_sched_barrier(0);
    const int u = i / 4, m = i % 4;
    asm volatile("" : "+v"(fh[i % 3]), "+v"(fl[i % 3]));
    const f16x8 ah = __builtin_bit_cast(f16x8, fh[i % 3]), al = __builtin_bit_cast(f16x8, fl[i % 3]);
    acc[m] = mfma(ah, b.lo[u], acc[m]);
    acc[m] = mfma(al, b.hi[u], acc[m]);
    acc[m] = mfma(ah, b.hi[u], acc[m]);
    __builtin_amdgcn_sched_barrier(0);
    if (i % (32 / NDMA) == 1 && i / (32 / NDMA) < NDMA) {
      const int k = i / (32 / NDMA);
      glds16_s(src + k * 256, voff, __builtin_amdgcn_readfirstlane(dst_lds + (unsigned)k * 1024u));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void matrix_phase(f32x16 (&acc)[4], const u32x4* a_lds, int lane, const Frag& b) {
  u32x4 fh[3], fl[3];
  const u32x4* a = a_lds + lane;
#pragma unroll
  for (int i = 0; i < 2; ++i) fh[i] = a[i * 128], fl[i] = a[i * 128 + 64];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    if (i + 2 < 32) fh[(i + 2) % 3] = a[(i + 2) * 128], fl[(i + 2) % 3] = a[(i + 2) * 128 + 64];
    __builtin_amdgcn_sched_barrier(0);
    const int u = i / 4, m = i % 4;
    asm volatile("" : "+v"(fh[i % 3]), "+v"(fl[i % 3]));
    const f16x8 ah = __builtin_bit_cast(f16x8, fh[i % 3]), al = __builtin_bit_cast(f16x8, fl[i % 3]);
    acc[m] = mfma(ah, b.lo[u], acc[m]);
    acc[m] = mfma(al, b.hi[u], acc[m]);
    acc[m] = mfma(ah, b.hi[u], acc[m]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

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

__device__ __forceinline__ void glds4_s(const float* uniform_src, unsigned lane_off_bytes, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_off_bytes), "s"(uniform_src), "s"(lds_byte_addr)
      : "memory");
}

// NV vector instructions on 64 registers (independent chains of fma: issue-bound, like FiLM / gain / split)
__device__ __forceinline__ void vector_phase(float (&x)[64], float p, float q) {
#pragma unroll
  for (int r = 0; r < NV / 64; ++r)
#pragma unroll
    for (int i = 0; i < 64; ++i) x[i] = __builtin_fmaf(x[i], p, q);
}

__device__ __forceinline__ void vector_phase_part(float (&x)[64], float p, float q, int k) {
#pragma unroll
  for (int i = 0; i < NV / 16; ++i) {
    const int j = (k * (NV / 16) + i) % 64;
    x[j] = __builtin_fmaf(x[j], p, q);
  }
}

template <int MODE>
__global__ __launch_bounds__((MODE == 1 || MODE >= 4) ? 512 : 256) void bench(const float* in, float* out, unsigned long long* cyc, const float* wsrc) {
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32 * 128 + 64; i += blockDim.x) lds[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  Frag b;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    b.hi[u] = __builtin_bit_cast(f16x8, lds[u * 64 + lane]);
    b.lo[u] = b.hi[u];
  }
  f32x16 acc[4], acc2[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[m][k] = in[k], acc2[m][k] = in[k + 1];
  float x[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) x[i] = in[i + lane];
  const float p = in[100], q = in[101];
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    for (int r = 0; r < REPS; ++r) {
      matrix_phase(acc, lds, lane, b);
      __builtin_amdgcn_s_barrier();
    }
  } else if (MODE == 3) {
    for (int r = 0; r < REPS; ++r) {
      matrix_phase(acc, lds, lane, b);
      __builtin_amdgcn_sched_barrier(0);
      vector_phase(x, p, q);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  } else if (MODE == 1) {
    const int team = wave >> 2;
    for (int r = 0; r < REPS; ++r) {
      if ((r & 1) == team) {
        matrix_phase(acc, lds, lane, b);
      } else {
        vector_phase(x, p, q);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (MODE >= 20) {
    // ping-pong with the decoder's other ingredients in the vector phase:
    //   20: + 8 LDS-DMA pieces (1 KiB each) per wave at the start of the vector phase, waited for at its end
    //   21: + 16 ds_read_b128 of headers spread over the vector phase
    //   22: both
    //   23: DMA, but waited for only at the end of the NEXT matrix phase of the issuing wave (i.e. two phases later)
    const int team = wave >> 2, tw = wave & 3;
    unsigned long long tm = 0, tv = 0;
    const unsigned dma_lds = (32 * 128 + 64) * 16;  // 32 KiB target area behind the fragments
    for (int r = 0; r < REPS; ++r) {
      const unsigned long long a0 = __builtin_amdgcn_s_memtime();
      if ((r & 1) == team) {
        if (MODE == 28 || MODE == 29)
          matrix_phase_dma<(MODE == 28 ? 8 : 16)>(acc, lds, lane, b, wsrc + __builtin_amdgcn_readfirstlane((((r >> 1) & 7) * 32 + tw * 8) * 256),
                                  __builtin_amdgcn_readfirstlane(dma_lds + (unsigned)(tw * 8) * 1024u));
        else if (MODE != 27) matrix_phase(acc, lds, lane, b);
        if (MODE == 28 || MODE == 29) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 23) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tm += __builtin_amdgcn_s_memtime() - a0;
      } else {
        u32x4 stg[8];
        if (MODE == 24) {  // the same 8 KiB as 32 dword-wide requests
#pragma unroll
          for (int k = 0; k < 32; ++k) {
            unsigned vo = (unsigned)lane * 4u;
            asm volatile("" : "+v"(vo));
            glds4_s(wsrc + __builtin_amdgcn_readfirstlane((((r >> 1) & 7) * 32 + tw * 8) * 256 + k * 64), vo,
                    __builtin_amdgcn_readfirstlane(dma_lds + (unsigned)(tw * 8) * 1024u + (unsigned)k * 256u));
          }
        }
        if (MODE == 25) {  // through registers: 8 global_load_dwordx4 now, 8 ds_write_b128 at the end of the phase
          const u32x4* g = (const u32x4*)(wsrc + __builtin_amdgcn_readfirstlane((((r >> 1) & 7) * 32 + tw * 8) * 256)) + lane;
#pragma unroll
          for (int k = 0; k < 8; ++k) stg[k] = __builtin_nontemporal_load(g + k * 64);
        }
        if (MODE == 20 || MODE == 22 || MODE == 23) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            unsigned vo = (unsigned)lane * 16u;
            asm volatile("" : "+v"(vo));
            glds16_s(wsrc + __builtin_amdgcn_readfirstlane((((r >> 1) & 7) * 32 + tw * 8 + k) * 256), vo,
                     __builtin_amdgcn_readfirstlane(dma_lds + (unsigned)(tw * 8 + k) * 1024u));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 21 || MODE == 22) {
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const u32x4 hv = (lds + lane)[k * 64 + 7];
            x[k] += __builtin_bit_cast(float, hv.x);
            vector_phase_part(x, p, q, k);
          }
        } else {
          vector_phase(x, p, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 25) {
          u32x4* d = (u32x4*)((char*)lds + dma_lds) + tw * 8 * 64 + lane;
#pragma unroll
          for (int k = 0; k < 8; ++k) d[k * 64] = stg[k];
        }
        if (MODE == 20 || MODE == 22 || MODE == 24) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tv += __builtin_amdgcn_s_memtime() - a0;
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (lane == 0) cyc[256 * 8 + blockIdx.x * 8 + wave] = tm, cyc[2 * 256 * 8 + blockIdx.x * 8 + wave] = tv;
  } else if (MODE >= 4) {
    // ping-pong with the loop body unrolled MODE x 2 times: the decoder's tile loop is 74 KB of straight-line code
    const int team = wave >> 2;
    for (int r = 0; r < REPS / (2 * MODE); ++r) {
#pragma unroll
      for (int s = 0; s < 2 * MODE; ++s) {
        if ((s & 1) == team) {
          matrix_phase(acc, lds, lane, b);
        } else {
          vector_phase(x, p, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    // one stream: group 0's matrix phase with group 1's vector phase
    for (int r = 0; r < REPS; ++r) {
      u32x4 fh[3], fl[3];
      const u32x4* a = lds + lane;
#pragma unroll
      for (int i = 0; i < 2; ++i) fh[i] = a[i * 128], fl[i] = a[i * 128 + 64];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        if (i + 2 < 32) fh[(i + 2) % 3] = a[(i + 2) * 128], fl[(i + 2) % 3] = a[(i + 2) * 128 + 64];
        const int u = i / 4, m = i % 4;
        const f16x8 ah = __builtin_bit_cast(f16x8, fh[i % 3]), al = __builtin_bit_cast(f16x8, fl[i % 3]);
        acc[m] = mfma(ah, b.lo[u], acc[m]);
        acc[m] = mfma(al, b.hi[u], acc[m]);
        acc[m] = mfma(ah, b.hi[u], acc[m]);
#pragma unroll
        for (int k = 0; k < NV / 32; ++k) {
          const int j = (i * (NV / 32) + k) % 64;
          x[j] = __builtin_fmaf(x[j], p, q);
        }
        // pattern per unit: 2 LDS reads, then 3 x (1 MFMA, FILL vector instructions), the rest of the unit's vector share after
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x002, NV / 32 - 3 * FILL > 0 ? NV / 32 - 3 * FILL : 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[m][k] + acc2[m][k];
#pragma unroll
  for (int i = 0; i < 64; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const float* in, float* out, unsigned long long* cyc, const float* wsrc) {
  const int threads = (MODE == 1 || MODE >= 4) ? 512 : 256;
  const size_t lds = (32 * 128 + 64) * 16 + 32 * 1024;
  hipFuncSetAttribute((const void*)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(threads), lds, 0, in, out, cyc, wsrc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(3 * 256 * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  int n = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < threads / 64; ++w) sum += (double)h[b * 8 + w], ++n;
  printf("%-58s %8.0f cycles per slot (96 MFMA = 3072)", name, sum / n / REPS);
  if (MODE >= 20) {
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 8; ++w) m += (double)h[256 * 8 + b * 8 + w], v += (double)h[2 * 256 * 8 + b * 8 + w];
    printf("   matrix phase %6.0f, vector phase %6.0f", m / n / (REPS / 2), v / n / (REPS / 2));
  }
  printf("\n");
}

int main() {
  float *in, *out;
  unsigned long long* cyc;
  hipMalloc(&in, 4096);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 3 * 256 * 8 * 8);
  float* wsrc;
  hipMalloc(&wsrc, 16 * 32 * 1024);
  hipMemset(wsrc, 0, 16 * 32 * 1024);
  std::vector<float> h(1024, 0.001f);
  h[100] = 0.999f;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  printf("NV=%d FILL=%d\n", NV, FILL);
  run<0>("0: matrix phase alone, 1 wave/SIMD", in, out, cyc, wsrc);
  run<3>("3: matrix then vector phase, serial, 1 wave/SIMD", in, out, cyc, wsrc);
  run<1>("1: ping-pong, 2 waves/SIMD (slot = M of A || V of B)", in, out, cyc, wsrc);
  run<2>("2: one stream, MFMA : vector interleaved, 1 wave/SIMD", in, out, cyc, wsrc);
  run<4>("4: ping-pong, body unrolled x8 (~30 KB of code)", in, out, cyc, wsrc);
  run<8>("8: ping-pong, body unrolled x16 (~60 KB)", in, out, cyc, wsrc);
  run<12>("12: ping-pong, body unrolled x24 (~90 KB)", in, out, cyc, wsrc);
  run<16>("16: ping-pong, body unrolled x32 (~120 KB)", in, out, cyc, wsrc);
  run<20>("20: ping-pong + 8 KiB LDS-DMA per wave and vector phase", in, out, cyc, wsrc);
  run<21>("21: ping-pong + 16 ds_read_b128 in the vector phase", in, out, cyc, wsrc);
  run<22>("22: ping-pong + both", in, out, cyc, wsrc);
  run<23>("23: ping-pong + DMA waited for one phase later", in, out, cyc, wsrc);
  run<26>("26: ping-pong, plain (phase stamps)", in, out, cyc, wsrc);
  run<27>("27: vector phases only (the matrix team idles at the barrier)", in, out, cyc, wsrc);
  run<28>("28: ping-pong, 8 LDS-DMA pieces per wave spread over the MATRIX phase", in, out, cyc, wsrc);
  run<29>("29: ping-pong, 16 pieces per wave spread over the matrix phase (all of a stage)", in, out, cyc, wsrc);
  run<24>("24: ping-pong + the 8 KiB as 32 dword LDS-DMA requests", in, out, cyc, wsrc);
  run<25>("25: ping-pong + 8 KiB through registers (8 loads, 8 ds_write)", in, out, cyc, wsrc);
  return 0;
}
