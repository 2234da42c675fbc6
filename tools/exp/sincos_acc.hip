// experiment: accuracy of (two-float 1/2pi range reduction + v_sin_f32/v_cos_f32) vs fp64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void fast_sincos(float arg, float& s, float& c) {
  const float C_HI = 0.15915494309189535f;             // fl(1/2pi)
  const float C_LO = (float)(0.15915494309189535 - (double)0.15915494309189535f);
  float th = arg * C_HI;
  float tl = __builtin_fmaf(arg, C_HI, -th) + arg * C_LO;
  float r = (th - rintf(th)) + tl;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}
__global__ void k(const float* x, float* s1, float* c1, float* s2, float* c2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fast_sincos(x[i], s1[i], c1[i]);
  sincosf(x[i], &s2[i], &c2[i]);
}
int main() {
  const int n = 1 << 20;
  std::vector<float> x(n);
  for (int i = 0; i < n; ++i) { float u = (float)i / n; int l = i % 10; x[i] = (u * 1.3f - 0.15f) * ldexpf(1.0f, l) * ((i & 1) ? 3.14159265f : 1.0f); }
  float *dx, *d[4]; hipMalloc(&dx, n * 4); for (auto& p : d) hipMalloc(&p, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d[0], d[1], d[2], d[3], n);
  std::vector<float> r[4]; for (int j = 0; j < 4; ++j) { r[j].resize(n); hipMemcpy(r[j].data(), d[j], n * 4, hipMemcpyDeviceToHost); }
  double e[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) { double s = sin((double)x[i]), c = cos((double)x[i]);
    e[0] = fmax(e[0], fabs(r[0][i] - s)); e[1] = fmax(e[1], fabs(r[1][i] - c)); e[2] = fmax(e[2], fabs(r[2][i] - s)); e[3] = fmax(e[3], fabs(r[3][i] - c)); }
  printf("max abs err  fast sin %.3e cos %.3e | ocml sincosf sin %.3e cos %.3e  (|x| up to %.1f)\n", e[0], e[1], e[2], e[3], 1.15 * 512 * 3.1416);
  return 0;
}
