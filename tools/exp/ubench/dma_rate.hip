// How fast can a CU pull an L2-resident weight stream into LDS?  (the ping-pong decoder's weight requests)
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), BURST pieces of 1 KiB per wave, then s_waitcnt vmcnt(0); repeated
//   mode 1: global_load_dwordx4 into registers + ds_write_b128 (same pieces)
//   mode 2: global_load_dwordx4 into registers only (no LDS write)
// waves per CU: 4 or 8 (one workgroup per CU); grid 256 (all CUs) or 32; the source is a 512 KiB buffer every CU reads (L2 hits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define REPS 128

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

template <int MODE, int BURST>
__global__ __launch_bounds__(512) void rate(const float* wsrc, float* out, unsigned long long* cyc) {
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REPS; ++r) {
    // 512 KiB source = 512 pieces; this wave's pieces of round r
    const int base = __builtin_amdgcn_readfirstlane(((r * nw + wave) * BURST) & 511);
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < BURST; ++k) {
        unsigned vo = (unsigned)lane * 16u;
        asm volatile("" : "+v"(vo));
        glds16_s(wsrc + ((base + k) & 511) * 256, vo, __builtin_amdgcn_readfirstlane((unsigned)((wave * BURST + k) & 127) * 1024u));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[BURST];
#pragma unroll
      for (int k = 0; k < BURST; ++k) v[k] = ((const u32x4*)(wsrc + ((base + k) & 511) * 256))[lane];
      if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < BURST; ++k) lds[((wave * BURST + k) & 127) * 64 + lane] = v[k];
      } else {
#pragma unroll
        for (int k = 0; k < BURST; ++k) acc.x ^= v[k].x, acc.y ^= v[k].y, acc.z ^= v[k].z, acc.w ^= v[k].w;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  out[blockIdx.x * blockDim.x + tid] = __builtin_bit_cast(float, acc.x ^ lds[tid].x);
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int BURST>
static void run(const char* name, int grid, int waves, const float* wsrc, float* out, unsigned long long* cyc) {
  const size_t lds = 128 * 1024;
  (void)hipFuncSetAttribute((const void*)rate<MODE, BURST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((rate<MODE, BURST>), dim3(grid), dim3(waves * 64), lds, 0, wsrc, out, cyc);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  int n = 0;
  for (int b = 0; b < grid; ++b)
    for (int w = 0; w < waves; ++w) sum += (double)h[b * 8 + w], ++n;
  const double cycles = sum / n;
  printf("%-44s grid %3d, %d waves, bursts of %2d: %6.1f B/clk per CU, %5.0f cycles per piece and wave\n", name, grid, waves, BURST,
         (double)REPS * waves * BURST * 1024 / cycles, cycles / REPS / BURST);
}

int main() {
  float *wsrc, *out;
  unsigned long long* cyc;
  (void)hipMalloc(&wsrc, 512 * 1024);
  (void)hipMemset(wsrc, 0, 512 * 1024);
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&cyc, 256 * 8 * 8);
  for (int grid : {256, 32}) {
    for (int waves : {4, 8}) {
      run<0, 8>("LDS-DMA (global_load_lds_dwordx4)", grid, waves, wsrc, out, cyc);
      run<0, 2>("LDS-DMA (global_load_lds_dwordx4)", grid, waves, wsrc, out, cyc);
      run<1, 8>("registers + ds_write_b128", grid, waves, wsrc, out, cyc);
      run<2, 8>("registers only", grid, waves, wsrc, out, cyc);
    }
  }
  run<0, 16>("LDS-DMA (global_load_lds_dwordx4)", 256, 8, wsrc, out, cyc);
  run<1, 16>("registers + ds_write_b128", 256, 8, wsrc, out, cyc);
  return 0;
}
