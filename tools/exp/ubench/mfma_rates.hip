// Microbenchmark for the round-6 cost volume (DESIGN.md section 4 K1+K2 "matrix form"): issue rate of the four fp16 matrix
// instructions a texel-window interpolation could use on gfx950, and how a wave's fp32 vector work overlaps ANOTHER wave's
// matrix work on the same SIMD.
//   v_mfma_f32_32x32x8_f16   (CDNA3 shape, K = 8)     v_mfma_f32_32x32x16_f16 (gfx950 shape, K = 16)
//   v_mfma_f32_16x16x16_f16  (CDNA3 shape, K = 16)    v_mfma_f32_16x16x32_f16 (gfx950 shape, K = 32)
// mode A: W waves per SIMD, each issues REPS x 8 independent MFMAs back to back -> cycles per MFMA and SIMD
// mode B: 2 waves per SIMD, both run the cost volume's mix: NM MFMAs (two accumulator sets) + NV fp32 FMAs on 64 registers that
//         depend on nothing the MFMAs write, per slot -> cycles per slot against NM*cycles and NV*4
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/ubench/mfma_rates.hip -o tools/exp/ubench/mfma_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REPS 256

template <int KIND>
struct Op;
template <>
struct Op<0> {  // 32x32x8 f16
  typedef f16x4 AB;
  typedef f32x16 C;
  static __device__ __forceinline__ C go(AB a, AB b, C c) { return __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0); }
  static constexpr const char* name = "v_mfma_f32_32x32x8_f16 ";
  static constexpr double macs = 32.0 * 32 * 8;
};
template <>
struct Op<1> {  // 32x32x16 f16
  typedef f16x8 AB;
  typedef f32x16 C;
  static __device__ __forceinline__ C go(AB a, AB b, C c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static constexpr const char* name = "v_mfma_f32_32x32x16_f16";
  static constexpr double macs = 32.0 * 32 * 16;
};
template <>
struct Op<2> {  // 16x16x16 f16
  typedef f16x4 AB;
  typedef f32x4 C;
  static __device__ __forceinline__ C go(AB a, AB b, C c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
  static constexpr const char* name = "v_mfma_f32_16x16x16_f16";
  static constexpr double macs = 16.0 * 16 * 16;
};
template <>
struct Op<3> {  // 16x16x32 f16
  typedef f16x8 AB;
  typedef f32x4 C;
  static __device__ __forceinline__ C go(AB a, AB b, C c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static constexpr const char* name = "v_mfma_f32_16x16x32_f16";
  static constexpr double macs = 16.0 * 16 * 32;
};

template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(float* out, unsigned long long* cyc) {
  typedef Op<KIND> O;
  typename O::AB a, b;
  for (int i = 0; i < (int)(sizeof(a) / 2); ++i) a[i] = (_Float16)(0.001f * (threadIdx.x + i)), b[i] = (_Float16)(0.002f * (i + 1));
  typename O::C acc[8];
  for (int m = 0; m < 8; ++m)
    for (int i = 0; i < (int)(sizeof(acc[0]) / 4); ++i) acc[m][i] = 0.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REPS; ++r) {
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = O::go(a, b, acc[m]);
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  float s = 0;
  for (int m = 0; m < 8; ++m) s += acc[m][0];
  asm volatile("" : "+v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// mode B: the cost volume's slot: NM MFMAs (32x32x8, 3 products on 2 accumulator sets) and NV independent fp32 FMAs
template <int KIND, int NM, int NV>
__global__ __launch_bounds__(512) void mix_kernel(float* out, unsigned long long* cyc, int slots) {
  typedef Op<KIND> O;
  typename O::AB a, b;
  for (int i = 0; i < (int)(sizeof(a) / 2); ++i) a[i] = (_Float16)(0.001f * (threadIdx.x + i)), b[i] = (_Float16)(0.002f * (i + 1));
  typename O::C acc[4];
  for (int m = 0; m < 4; ++m)
    for (int i = 0; i < (int)(sizeof(acc[0]) / 4); ++i) acc[m][i] = 0.0f;
  float x[64];
  for (int i = 0; i < 64; ++i) x[i] = 0.01f * (threadIdx.x + i);
  const float p = 0.999f, q = 0.001f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < slots; ++s) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 3] = O::go(a, b, acc[m & 3]);
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i & 63] = __builtin_fmaf(x[i & 63], p, q);
  }
  float s2 = 0;
  for (int m = 0; m < 4; ++m) s2 += acc[m][0];
  for (int i = 0; i < 64; ++i) s2 += x[i];
  asm volatile("" : "+v"(s2));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s2;
}

template <int KIND>
static void run_rate(float* out, unsigned long long* cyc, int waves_per_simd) {
  const int threads = 64 * 4 * waves_per_simd;  // one workgroup fills every SIMD of its CU with `waves_per_simd` waves
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  // s_memtime ticks at 100 MHz on gfx9-class parts; convert with the measured ratio below
  printf("%s  %d wave(s)/SIMD: %.2f memtime ticks per MFMA and SIMD\n", Op<KIND>::name, waves_per_simd,
         s / 256 / (REPS * 8.0 * waves_per_simd));
}

template <int KIND, int NM, int NV>
static void run_mix(float* out, unsigned long long* cyc, int waves_per_simd) {
  const int threads = 64 * 4 * waves_per_simd, slots = 64;
  hipLaunchKernelGGL((mix_kernel<KIND, NM, NV>), dim3(256), dim3(threads), 0, 0, out, cyc, slots);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%s  mix NM=%d NV=%d, %d wave(s)/SIMD: %.1f ticks per slot and wave, %.1f per slot and SIMD\n", Op<KIND>::name, NM, NV,
         waves_per_simd, s / 256 / slots, s / 256 / slots / waves_per_simd);
}

// wall-clock version of the same: total time of a launch / MFMAs per SIMD -> ns per MFMA; with the shader clock from
// hipDeviceProp this gives cycles without trusting s_memtime's tick.
template <int KIND>
static void run_wall(float* out, unsigned long long* cyc, int waves_per_simd, double clock_ghz) {
  const int threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256 * 8), dim3(threads), 0, 0, out, cyc);
  hipEventRecord(e0);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256 * 8), dim3(threads), 0, 0, out, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // 256 * 8 workgroups over 256 CUs = 8 rounds (one workgroup per CU at a time only if it fills the CU; smaller ones share)
  const double per_simd = 4.0 * 8 * REPS * 8.0 * waves_per_simd * 1.0;  // launches x rounds x MFMAs per wave x waves
  const double ns = ms * 1e6 / per_simd;
  printf("%s  wall, %d wave(s)/SIMD: %.2f ns per MFMA and SIMD = %.1f cycles at %.2f GHz -> %.0f TFLOP/s chip\n", Op<KIND>::name,
         waves_per_simd, ns, ns * clock_ghz, clock_ghz, 2.0 * Op<KIND>::macs / ns * 1e9 * 1024 / 1e12);
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 8 * 512 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double ghz = prop.clockRate / 1e6;
  printf("device %s, %d CUs, clock %.2f GHz\n", prop.name, prop.multiProcessorCount, ghz);
  for (int w = 1; w <= 2; ++w) {
    run_rate<0>(out, cyc, w);
    run_rate<1>(out, cyc, w);
    run_rate<2>(out, cyc, w);
    run_rate<3>(out, cyc, w);
  }
  for (int w = 1; w <= 2; ++w) {
    run_wall<0>(out, cyc, w, ghz);
    run_wall<1>(out, cyc, w, ghz);
    run_wall<2>(out, cyc, w, ghz);
    run_wall<3>(out, cyc, w, ghz);
  }
  // the cost volume's slot (one 32-channel tile of a 32-ray unit, both maps): ~14 MFMAs next to ~110 vector instructions
  run_mix<0, 14, 112>(out, cyc, 1);
  run_mix<0, 14, 112>(out, cyc, 2);
  run_mix<0, 14, 0>(out, cyc, 2);
  run_mix<0, 0, 112>(out, cyc, 2);
  run_mix<1, 8, 112>(out, cyc, 2);
  return 0;
}
