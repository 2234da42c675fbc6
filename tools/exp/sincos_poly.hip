// experiment: two-float 1/2pi range reduction + odd minimax polynomial, cos = sin(quarter shift)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
__device__ __forceinline__ float sin_or_cos(float arg, int want_cos) {
  const float C_HI = 0.15915494309189535f;
  const float C_LO = (float)(0.15915494309189535 - (double)0.15915494309189535f);
  float th = arg * C_HI;
  float tl = __builtin_fmaf(arg, C_LO, __builtin_fmaf(arg, C_HI, -th));
  float r = th - rintf(th);                 // exact, in [-0.5, 0.5]
  r = r + (tl + (want_cos ? 0.25f : 0.0f)); // cos(x) = sin(x + quarter turn)
  r = r - rintf(r);
  float a = fabsf(r);
  r = (a > 0.25f) ? (copysignf(0.5f, r) - r) : r;   // sin(pi - t) = sin(t)
  float x2 = r * r;
  float p = 39.536705017089844f;
  p = __builtin_fmaf(p, x2, -76.5497817993164f);
  p = __builtin_fmaf(p, x2, 81.60100555419922f);
  p = __builtin_fmaf(p, x2, -41.34165573120117f);
  p = __builtin_fmaf(p, x2, 6.283185005187988f);
  return p * r;
}
#pragma clang fp contract(fast)
__global__ void k(const float* x, float* s1, float* c1, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s1[i] = sin_or_cos(x[i], 0);
  c1[i] = sin_or_cos(x[i], 1);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n);
  for (int i = 0; i < n; ++i) { float u = (float)(i >> 4) / (n >> 4); int l = i % 10; x[i] = (u * 1.6f - 0.3f) * ldexpf(1.0f, l) * ((i & 1) ? 3.14159265f : 1.0f); }
  float *dx, *d[2]; (void)hipMalloc(&dx, n * 4); for (auto& p : d) (void)hipMalloc(&p, n * 4);
  (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d[0], d[1], n);
  std::vector<float> r[2]; for (int j = 0; j < 2; ++j) { r[j].resize(n); (void)hipMemcpy(r[j].data(), d[j], n * 4, hipMemcpyDeviceToHost); }
  double e[2] = {0, 0}; double worst = 0;
  for (int i = 0; i < n; ++i) { double s = sin((double)x[i]), c = cos((double)x[i]);
    double es = fabs(r[0][i] - s), ec = fabs(r[1][i] - c); if (es > e[0]) { e[0] = es; worst = x[i]; } e[1] = fmax(e[1], ec); }
  printf("poly: max abs err sin %.3e cos %.3e (worst arg %.3f)\n", e[0], e[1], worst);
  return 0;
}
