// Micro-benchmark (round 3): what does the decoder's matrix phase cost per v_mfma_f32_32x32x16_f16?
// One 8-wave workgroup per CU (512 threads, <= 256 VGPRs: two waves per SIMD as the ping-pong decoder).  Waves 0-3 (or all
// eight) run `iters` repetitions of a 96-instruction layer: 32 units x 3 dependent MFMAs on 4 accumulators, A fragments
// from LDS (2 x ds_read_b128 per unit, two units of look-ahead) or from registers.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_stream mfma_stream.hip && ./mfma_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef const u4 __attribute__((address_space(3)))* lds_u4;

template <int MODE>  // 0: fragments in registers, 1: LDS reads, unit-major (a0 a0 a0 a1 a1 a1), 2: LDS reads, block pairs interleaved
__global__ __launch_bounds__(512, 2) void stream_kernel(int iters, int active_waves, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = 0x3c003c00u + (i & 7);
  __syncthreads();
  f16v acc[4];
  for (int m = 0; m < 4; ++m) acc[m] = (f16v)(0.0f);
  h8 bh[8], bl[8];
  for (int u = 0; u < 8; ++u)
    for (int j = 0; j < 8; ++j) {
      bh[u][j] = (_Float16)(0.001f * (lane + u + j));
      bl[u][j] = (_Float16)(0.00001f * (lane + j));
    }
  const unsigned base = __builtin_amdgcn_groupstaticsize();
  lds_u4 a = (lds_u4)(size_t)base + lane;
  unsigned long long t0 = 0, t1 = 0;
  if (wave < active_waves) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      if constexpr (MODE == 0) {
        u4 f0 = a[0], f1 = a[64];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int u = i / 4, m = i % 4;
          const h8 ah = __builtin_bit_cast(h8, f0), al = __builtin_bit_cast(h8, f1);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[u], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[u], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[u], acc[m], 0, 0, 0);
        }
      } else if constexpr (MODE == 1) {
        u4 fh[3], fl[3];
        fh[0] = a[0]; fl[0] = a[64]; fh[1] = a[128]; fl[1] = a[192];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          if (i + 2 < 32) { fh[(i + 2) % 3] = a[(i + 2) * 128]; fl[(i + 2) % 3] = a[(i + 2) * 128 + 64]; }
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("" : "+v"(fh[i % 3]), "+v"(fl[i % 3]));
          const int u = i / 4, m = i % 4;
          const h8 ah = __builtin_bit_cast(h8, fh[i % 3]), al = __builtin_bit_cast(h8, fl[i % 3]);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[u], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[u], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[u], acc[m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        u4 fh[2][2], fl[2][2];
        fh[0][0] = a[0]; fl[0][0] = a[64]; fh[0][1] = a[128]; fl[0][1] = a[192];
#pragma unroll
        for (int pr = 0; pr < 16; ++pr) {
          if (pr + 1 < 16) {
            fh[(pr + 1) & 1][0] = a[(2 * pr + 2) * 128]; fl[(pr + 1) & 1][0] = a[(2 * pr + 2) * 128 + 64];
            fh[(pr + 1) & 1][1] = a[(2 * pr + 3) * 128]; fl[(pr + 1) & 1][1] = a[(2 * pr + 3) * 128 + 64];
          }
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("" : "+v"(fh[pr & 1][0]), "+v"(fl[pr & 1][0]), "+v"(fh[pr & 1][1]), "+v"(fl[pr & 1][1]));
          const int u = (2 * pr) / 4, m = (2 * pr) % 4;
          const h8 ah0 = __builtin_bit_cast(h8, fh[pr & 1][0]), al0 = __builtin_bit_cast(h8, fl[pr & 1][0]);
          const h8 ah1 = __builtin_bit_cast(h8, fh[pr & 1][1]), al1 = __builtin_bit_cast(h8, fl[pr & 1][1]);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[u], acc[m], 0, 0, 0);
          acc[m + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl[u], acc[m + 1], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[u], acc[m], 0, 0, 0);
          acc[m + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh[u], acc[m + 1], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[u], acc[m], 0, 0, 0);
          acc[m + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh[u], acc[m + 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  float r = 0.f;
  for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][7];
  if (r == 12345.678f) sink[0] = r;
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

int main() {
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, 64);
  const int iters = 2000;
  const char* names[3] = {"fragments in registers", "LDS fragments, unit-major", "LDS fragments, block pairs interleaved"};
  for (int mode = 0; mode < 3; ++mode)
    for (int active = 4; active <= 8; active += 4) {
      hipMemset(out, 0, 64);
      void (*k)(int, int, unsigned long long*, float*) = mode == 0 ? stream_kernel<0> : (mode == 1 ? stream_kernel<1> : stream_kernel<2>);
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, iters, active, out, sink);
      hipDeviceSynchronize();
      unsigned long long h[8];
      hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
      printf("%-42s %d waves per CU active: %.1f cycles per MFMA (wave 0), %.1f (wave %d)\n", names[mode], active,
             (double)h[0] / (iters * 96.0), (double)h[active - 1] / (iters * 96.0), active - 1);
    }
  return 0;
}
