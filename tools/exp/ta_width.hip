// Micro-benchmark: cycles per wave-level global load instruction on the texture-address path by access width
// (dword / dwordx2 / dwordx4, all 64 lanes active, L1/L2-resident 16 KiB window, lanes 16 bytes apart).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int W, int STRIDE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  const char* base = reinterpret_cast<const char*>(src) + (blockIdx.x & 7) * 16384;
  unsigned off = (unsigned)lane * STRIDE;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const char* p = base + ((off + u * 1024u) & 16383u);
      if (W == 1) acc += *reinterpret_cast<const float*>(p);
      if (W == 2) { const v2f t = *reinterpret_cast<const v2f*>(p); acc += t.x + t.y; }
      if (W == 4) { const v4f t = *reinterpret_cast<const v4f*>(p); acc += t.x + t.y + t.z + t.w; }
    }
    off += 256u;
  }
  if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <int W, int STRIDE>
static void run(const char* name, const float* src, float* out) {
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<W, STRIDE>), dim3(blocks), dim3(256), 0, 0, src, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<W, STRIDE>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)blocks * 4 * iters * 16 / 256.0;
  printf("%-34s %8.3f ms  ~%5.1f cycles per wave-instruction per CU at 2.4 GHz\n", name, ms, ms * 1e6 / instr_per_cu * 2.4);
}

int main() {
  float *src, *out;
  hipMalloc(&src, 1 << 20);
  hipMalloc(&out, 4096);
  hipMemset(src, 0, 1 << 20);
  run<1, 4>("dword,   lanes contiguous (256 B)", src, out);
  run<2, 8>("dwordx2, lanes contiguous (512 B)", src, out);
  run<4, 16>("dwordx4, lanes contiguous (1 KiB)", src, out);
  run<1, 16>("dword,   lanes 16 B apart", src, out);
  run<1, 128>("dword,   lanes 128 B apart (64 lines)", src, out);
  run<4, 128>("dwordx4, lanes 128 B apart (64 lines)", src, out);
  return 0;
}
