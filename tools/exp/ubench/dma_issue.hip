// What does a wave pay to ISSUE a 1-KiB LDS-DMA request (global_load_lds_dwordx4), and is it the M0 choreography around it?
// Each wave loops: 8 requests in one of the forms below, a filler of FILL dependent v_fma (the "vector phase"), s_waitcnt vmcnt(0).
// cycles per iteration minus the filler-only loop = the requests' issue cost.
//   form 0: no requests (the filler alone)
//   form 1: the product's glds16_s: s_nop 4 | save m0 | m0 = dst | s_nop 0 | load | restore m0      (per request)
//   form 2: m0 written once per FOUR requests, the pieces addressed by the instruction offset (0, 1024, 2048, 3072: the offset
//           moves the global AND the LDS address), m0 restored once per four
//   form 3: m0 = dst | s_nop 0 | load per request, m0 never saved or restored (clobbered)
//   form 4: form 1 without the leading s_nop 4
// Form 2 / 3 also copy the LDS image out so that the host can check that the pieces landed where form 1 puts them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define REPS 256

template <int FORM>
__device__ __forceinline__ void issue8(const float* src, unsigned vo, unsigned lds0) {
  // 8 consecutive 1-KiB pieces: global src + k KiB -> LDS lds0 + k KiB
  if (FORM == 1 || FORM == 4) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      unsigned keep;
      const float* s = src + k * 256;
      const unsigned d = lds0 + k * 1024u;
      if (FORM == 1)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(s), "s"(d) : "memory");
      else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(s), "s"(d) : "memory");
    }
  } else if (FORM == 2) {
#pragma unroll
    for (int k = 0; k < 8; k += 4) {
      unsigned keep;
      const float* s = src + k * 256;
      const unsigned d = lds0 + k * 1024u;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
          "global_load_lds_dwordx4 %1, %2\n\t"
          "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
          "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
          "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep) : "v"(vo), "s"(s), "s"(d) : "memory");
    }
  } else if (FORM == 3) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float* s = src + k * 256;
      const unsigned d = lds0 + k * 1024u;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(s), "s"(d) : "memory", "m0");
    }
  }
}

template <int FORM, int FILL>
__global__ __launch_bounds__(512) void issue(const float* wsrc, float* out, unsigned long long* cyc, u32x4* image) {
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8 * 8 * 64; i += blockDim.x) lds[i] = (u32x4){0, 0, 0, 0};
  __syncthreads();
  float x = (float)tid, y = 1.0001f;
  unsigned vo = (unsigned)lane * 16u;
  asm volatile("" : "+v"(vo));
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)wave * 8192u);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REPS; ++r) {
    const float* src = wsrc + (size_t)__builtin_amdgcn_readfirstlane(((r * 8 + wave) * 8) & 511) * 256;
    issue8<FORM>(src, vo, lds0);
#pragma unroll
    for (int i = 0; i < FILL; ++i) x = __builtin_fmaf(x, y, 0.5f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  out[blockIdx.x * blockDim.x + tid] = x;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (image && blockIdx.x == 0)
    for (int i = tid; i < (int)(blockDim.x >> 6) * 8 * 64; i += blockDim.x) image[i] = lds[i];
}

template <int FORM, int FILL>
static double run(const char* name, int waves, const float* wsrc, float* out, unsigned long long* cyc, u32x4* image, double base) {
  const size_t lds = 64 * 1024;
  (void)hipFuncSetAttribute((const void*)issue<FORM, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((issue<FORM, FILL>), dim3(256), dim3(waves * 64), lds, 0, wsrc, out, cyc, image);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  int n = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < waves; ++w) sum += (double)h[b * 8 + w], ++n;
  const double per_iter = sum / n / REPS * (2400.0 / 100.0);  // s_memtime ticks at 100 MHz: x 24 = core cycles at 2.4 GHz
  printf("%-28s %d waves, filler %4d: %7.0f cycles per iteration", name, waves, FILL, per_iter);
  if (base > 0) printf("  -> %5.0f cycles per request", (per_iter - base) / 8);
  printf("\n");
  return per_iter;
}

int main() {
  float *wsrc, *out;
  unsigned long long* cyc;
  u32x4* image;
  std::vector<float> hsrc(512 * 256);
  for (size_t i = 0; i < hsrc.size(); ++i) hsrc[i] = (float)i;
  (void)hipMalloc(&wsrc, hsrc.size() * 4);
  (void)hipMemcpy(wsrc, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&cyc, 256 * 8 * 8);
  (void)hipMalloc(&image, 8 * 8 * 64 * 16);
  // correctness of the batched / unrestored forms: the LDS image of workgroup 0 against form 1's
  std::vector<unsigned> ref(8 * 8 * 64 * 4), got(ref.size());
  hipLaunchKernelGGL((issue<1, 64>), dim3(256), dim3(512), 64 * 1024, 0, wsrc, out, cyc, image);
  (void)hipMemcpy(ref.data(), image, ref.size() * 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL((issue<2, 64>), dim3(256), dim3(512), 64 * 1024, 0, wsrc, out, cyc, image);
  (void)hipMemcpy(got.data(), image, got.size() * 4, hipMemcpyDeviceToHost);
  printf("form 2 image %s form 1's (first words %u %u)\n", memcmp(ref.data(), got.data(), ref.size() * 4) ? "DIFFERS FROM" : "equals", ref[0], ref[256]);
  hipLaunchKernelGGL((issue<3, 64>), dim3(256), dim3(512), 64 * 1024, 0, wsrc, out, cyc, image);
  (void)hipMemcpy(got.data(), image, got.size() * 4, hipMemcpyDeviceToHost);
  printf("form 3 image %s form 1's\n", memcmp(ref.data(), got.data(), ref.size() * 4) ? "DIFFERS FROM" : "equals");
  for (int waves : {4, 8}) {
#define ALL(FILL)                                                                              \
  {                                                                                            \
    const double b = run<0, FILL>("no requests", waves, wsrc, out, cyc, nullptr, 0);           \
    run<1, FILL>("product form", waves, wsrc, out, cyc, nullptr, b);                           \
    run<4, FILL>("product form, no s_nop 4", waves, wsrc, out, cyc, nullptr, b);               \
    run<2, FILL>("m0 once per 4 (offsets)", waves, wsrc, out, cyc, nullptr, b);                \
    run<3, FILL>("m0 never restored", waves, wsrc, out, cyc, nullptr, b);                      \
  }
    ALL(256)
    ALL(768)
#undef ALL
  }
  return 0;
}
