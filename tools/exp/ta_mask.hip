// Micro-benchmark: what does a lane-masked global_load_dwordx4 cost on the texture-address path?
// Every wave streams dwordx4 loads out of an L1/L2-resident 16 KiB window with 64, 32, 16 (one row) or 16 (every
// 4th lane) lanes active.  Prints cycles per wave-instruction per CU.   hipcc --offload-arch=gfx950 -O3 ta_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  bool on = true;
  if (MODE == 1) on = lane < 32;
  if (MODE == 2) on = lane < 16;
  if (MODE == 3) on = (lane & 3) == 0;
  if (MODE == 4) on = (lane >> 4) == ((threadIdx.x >> 6) & 3);  // one 16-lane row, a different one per wave
  v4f acc = {0, 0, 0, 0};
  const char* base = reinterpret_cast<const char*>(src) + (blockIdx.x & 7) * 16384;
  unsigned off = (unsigned)lane * 32u;  // 32 B per lane as in the cost volume (two dwordx4 per tap)
  if (on) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const v4f* p = reinterpret_cast<const v4f*>(base + ((off + u * 2048u) & 16383u));
        acc += p[0];
        acc += p[1];
      }
      off += 512u;
    }
  }
  if (acc.x == 123.456f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
static void run(const char* name, const float* src, float* out) {
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)blocks * 4 * iters * 16 / 256.0;  // wave-level dwordx4 instructions per CU
  printf("%-28s %8.3f ms  %6.2f ns per wave-instruction per CU  (~%5.1f cycles at 2.4 GHz)\n", name, ms,
         ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
}

int main() {
  float *src, *out;
  hipMalloc(&src, 1 << 20);
  hipMalloc(&out, 4096);
  hipMemset(src, 0, 1 << 20);
  run<0>("64 lanes", src, out);
  run<1>("32 lanes (half)", src, out);
  run<2>("16 lanes (row 0)", src, out);
  run<3>("16 lanes (every 4th)", src, out);
  run<4>("16 lanes (row = wave id)", src, out);
  return 0;
}
