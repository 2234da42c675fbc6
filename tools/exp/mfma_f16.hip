// Layout and subnormal check of v_mfma_f32_32x32x16_f16 on gfx950 (assumptions of the split-fp16 decoder path):
//   A: lane (n, half) holds A[row n][k = 8*half + j], j < 8;  B: lane (n, half) holds B[k = 8*half + j][col n];
//   D: register r of lane (n, half) = D[row (r&3) + 8*(r>>2) + 4*half][col n]; fp16 subnormal inputs are not flushed.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_f16 tools/exp/mfma_f16.hip && /tmp/mfma_f16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D) {  // A [32][16], B [16][32] row-major, D [32][32]
  const int lane = threadIdx.x, n = lane & 31, half = lane >> 5;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)A[n * 16 + 8 * half + j];
    b[j] = (_Float16)B[(8 * half + j) * 32 + n];
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + n] = c[r];
}
int main() {
  float hA[32 * 16], hB[16 * 32], hD[32 * 32], *dA, *dB, *dD;
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) hA[i * 16 + kk] = (float)((i * 7 + kk * 3) % 13) - 6.0f;
  for (int kk = 0; kk < 16; ++kk) for (int j = 0; j < 32; ++j) hB[kk * 32 + j] = (float)((kk * 5 + j * 11) % 17) - 8.0f;
  hA[3 * 16 + 2] = 3.0e-6f;  // fp16 subnormal (min normal 6.1e-5)
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0; double sub_err = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double ref = 0;
    for (int kk = 0; kk < 16; ++kk) ref += (double)(float)(_Float16)hA[i * 16 + kk] * (double)hB[kk * 32 + j];
    if (fabs(ref - hD[i * 32 + j]) > 1e-3) ++bad;
    if (i == 3) sub_err = fmax(sub_err, fabs(ref - hD[i * 32 + j]));
  }
  // row 3 with the subnormal flushed would be off by 3e-6 * |B[2][j]| (up to 2.4e-5)
  printf("mfma_f32_32x32x16_f16 layout: %s (%d mismatches); subnormal row max err %.2e (%s)\n", bad ? "WRONG" : "OK", bad,
         sub_err, sub_err < 2e-6 ? "kept" : "flushed?");
  return bad != 0;
}
