// experiment: operand / result layout of v_mfma_f32_4x4x1_16b_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  int l = threadIdx.x;
  float a = (float)(l + 1), b = 100.0f * (float)(l + 1);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    float expect = (float)(4 * (l / 4) + r + 1) * 100.0f * (float)(l + 1);   // D[blk=l/4][i=r][j=l%4] = A[blk][i] * B[blk][j]
    if (h[l * 4 + r] != expect) ok = 0;
  }
  printf("hypothesis D[r](lane l) = a(lane 4*(l/4)+r) * b(lane l): %s\n", ok ? "CONFIRMED" : "WRONG");
  for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
