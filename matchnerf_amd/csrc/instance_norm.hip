// InstanceNorm2d (no affine, biased variance) + the activations / residual add that follow it in the GMFlow CNN
// backbone, one kernel per use (gfx950).
//
// Replaces (paths relative to /root/reference/models/gmflow/backbone.py):
//   backbone.py:27-35   ResidualBlock.forward: relu(norm1(conv1 x)), relu(norm2(conv2 y)), norm3(downsample x),
//                       relu(x + y)
//   backbone.py:101-103 CNNEncoder.forward: relu(norm1(conv1 x))
// torch runs each InstanceNorm as three kernels (batch-norm statistics, inverse std, transform) followed by a clamp
// and, at the end of a block, an add and another clamp: five to seven passes over the activation.  Here a workgroup
// owns one (image, channel) plane of the NCHW tensor, keeps it in REGISTERS (up to 160 floats per lane at 512 lanes:
// the 256x320 plane of the first stage), takes mean and centred variance from the registers (two workgroup
// reductions) and writes the finished value once: one read, one write - the kernel is HBM/L2-bound by construction.
// Planes that do not fit (or whose size is not a multiple of four) take a three-pass streaming kernel.
#include "common.hpp"

template <int THREADS>
__device__ __forceinline__ float in_block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // red may still be read by the previous reduction
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float t = 0.0f;
#pragma unroll
  for (int i = 0; i < THREADS / 64; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ float in_finish(float x, float mean, float rstd, float res, bool has_res, int relu_inner,
                                           int relu_outer) {
  float v = (x - mean) * rstd;
  if (relu_inner) v = fmaxf(v, 0.0f);
  if (has_res) v += res;
  if (relu_outer) v = fmaxf(v, 0.0f);
  return v;
}

// largest |out| of the plane -> the tensor's absmax region (common.hpp): the operand scale of the split-fp16
// convolution that reads the result (conv.hip).  One atomic per workgroup.
template <int THREADS>
__device__ __forceinline__ void in_merge_absmax(float m, float* red, float* out_absmax) {
  if (!out_absmax) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < THREADS / 64; ++i) m = fmaxf(m, red[i]);
    mnerf_absmax_merge(m, out_absmax);
  }
}

template <int THREADS, int VPT>
__global__ __launch_bounds__(THREADS) void instance_norm_cached_kernel(const float* x,
                                                                       const float* __restrict__ residual,
                                                                       float* out, int plane_size, float eps,
                                                                       int relu_inner, int relu_outer,
                                                                       float* __restrict__ out_absmax) {
  __shared__ float red[THREADS / 64];
  const size_t base = (size_t)blockIdx.x * plane_size;
  const float4* src = reinterpret_cast<const float4*>(x + base);
  const int n4 = plane_size >> 2;
  float4 v[VPT];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int j = i * THREADS + threadIdx.x;
    v[i] = j < n4 ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float inv_n = 1.0f / (float)plane_size;
  const float mean = in_block_sum<THREADS>(s, red) * inv_n;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int j = i * THREADS + threadIdx.x;
    if (j < n4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float var = in_block_sum<THREADS>(q, red) * inv_n;
  const float rstd = 1.0f / sqrtf(var + eps);
  // (an opaque copy of the mean: otherwise the centred values of the variance pass are kept for the output pass — a
  // second copy of the plane in registers, which without packed arithmetic no longer fits 256 of them)
  float mean_o = mean;
  asm volatile("" : "+v"(mean_o));
  const float4* res = residual ? reinterpret_cast<const float4*>(residual + base) : nullptr;
  float4* dst = reinterpret_cast<float4*>(out + base);
  float omax = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int j = i * THREADS + threadIdx.x;
    if (j < n4) {
      const float4 r = res ? res[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o;
      o.x = in_finish(v[i].x, mean_o, rstd, r.x, res != nullptr, relu_inner, relu_outer);
      o.y = in_finish(v[i].y, mean_o, rstd, r.y, res != nullptr, relu_inner, relu_outer);
      o.z = in_finish(v[i].z, mean_o, rstd, r.z, res != nullptr, relu_inner, relu_outer);
      o.w = in_finish(v[i].w, mean_o, rstd, r.w, res != nullptr, relu_inner, relu_outer);
      dst[j] = o;
      omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  in_merge_absmax<THREADS>(omax, red, out_absmax);
}

// any plane size: three passes (the plane of a running workgroup stays in L2 / Infinity Cache between them)
__global__ __launch_bounds__(256) void instance_norm_stream_kernel(const float* x,
                                                                   const float* __restrict__ residual,
                                                                   float* out, int plane_size, float eps,
                                                                   int relu_inner, int relu_outer,
                                                                   float* __restrict__ out_absmax) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * plane_size;
  const float* src = x + base;
  float s = 0.0f;
  for (int j = threadIdx.x; j < plane_size; j += 256) s += src[j];
  const float inv_n = 1.0f / (float)plane_size;
  const float mean = in_block_sum<256>(s, red) * inv_n;
  float q = 0.0f;
  for (int j = threadIdx.x; j < plane_size; j += 256) {
    const float a = src[j] - mean;
    q += a * a;
  }
  const float rstd = 1.0f / sqrtf(in_block_sum<256>(q, red) * inv_n + eps);
  float omax = 0.0f;
  for (int j = threadIdx.x; j < plane_size; j += 256) {
    const float o = in_finish(src[j], mean, rstd, residual ? residual[base + j] : 0.0f, residual != nullptr, relu_inner,
                              relu_outer);
    out[base + j] = o;
    omax = fmaxf(omax, fabsf(o));
  }
  in_merge_absmax<256>(omax, red, out_absmax);
}

extern "C" int mnerf_instance_norm(const float* x, const float* residual, float* out, int64_t planes,
                                   int64_t plane_size, float eps, int32_t relu_inner, int32_t relu_outer,
                                   float* out_absmax, void* stream) {
  MNERF_REQUIRE(planes >= 0 && planes <= 0x7fffffffLL && plane_size >= 1 && plane_size <= 0x7fffffffLL, MNERF_E_RANGE,
                "mnerf_instance_norm: planes=%lld plane_size=%lld", (long long)planes, (long long)plane_size);
  MNERF_REQUIRE(eps >= 0.0f, MNERF_E_RANGE, "mnerf_instance_norm: eps=%g", (double)eps);
  if (planes == 0) return MNERF_OK;
  MNERF_REQUIRE(x && out, MNERF_E_NULL, "mnerf_instance_norm: NULL buffer");
  hipStream_t st = (hipStream_t)stream;
  const int n = (int)plane_size;
  const bool vec = (n & 3) == 0 && mnerf_aligned16(x) && mnerf_aligned16(out) && (!residual || mnerf_aligned16(residual));
  const dim3 grid((unsigned)planes);
#define IN_LAUNCH(T, V)                                                                                              \
  hipLaunchKernelGGL((instance_norm_cached_kernel<T, V>), grid, dim3(T), 0, st, x, residual, out, n, eps, relu_inner, \
                     relu_outer, out_absmax)
  if (vec && n <= 256 * 8 * 4) IN_LAUNCH(256, 8);
  else if (vec && n <= 256 * 20 * 4) IN_LAUNCH(256, 20);
  else if (vec && n <= 512 * 40 * 4) IN_LAUNCH(512, 40);  // two workgroups per CU: one plane loads while another stores
  else
    hipLaunchKernelGGL(instance_norm_stream_kernel, grid, dim3(256), 0, st, x, residual, out, n, eps, relu_inner, relu_outer,
                       out_absmax);
#undef IN_LAUNCH
  return mnerf_check_launch("mnerf_instance_norm");
}

// ---------------------------------------------------------------------------------------------------------------- backward
// y = [relu](IN(x)):  g = dy where the output passed the ReLU (xhat > 0), else 0;
//   dx = rstd (g - mean(g) - xhat mean(g xhat)),   xhat = (x - mean) rstd   (biased variance, no affine: backbone.py:12-14)
// One workgroup per plane, the statistics of x are re-derived as the forward derives them (mean, then the centred variance), so the
// training path keeps nothing but x itself from the forward.
// the streaming form: any plane size
__global__ __launch_bounds__(512) void instance_norm_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                                     int plane_size, float eps, int relu) {
  __shared__ float red[8];
  const size_t base = (size_t)blockIdx.x * plane_size;
  const float* xs = x + base;
  const float* gs = dy + base;
  const float inv_n = 1.0f / (float)plane_size;
  float s = 0.0f;
  for (int j = threadIdx.x; j < plane_size; j += 512) s += xs[j];
  const float mean = in_block_sum<512>(s, red) * inv_n;
  float q = 0.0f;
  for (int j = threadIdx.x; j < plane_size; j += 512) {
    const float a = xs[j] - mean;
    q += a * a;
  }
  const float rstd = 1.0f / sqrtf(in_block_sum<512>(q, red) * inv_n + eps);
  float sg = 0.0f, sgx = 0.0f;
  for (int j = threadIdx.x; j < plane_size; j += 512) {
    const float xh = (xs[j] - mean) * rstd;
    const float g = (relu && !(xh > 0.0f)) ? 0.0f : gs[j];
    sg += g;
    sgx += g * xh;
  }
  const float mg = in_block_sum<512>(sg, red) * inv_n;
  const float mgx = in_block_sum<512>(sgx, red) * inv_n;
  for (int j = threadIdx.x; j < plane_size; j += 512) {
    const float xh = (xs[j] - mean) * rstd;
    const float g = (relu && !(xh > 0.0f)) ? 0.0f : gs[j];
    dx[base + j] = rstd * (g - mg - xh * mgx);
  }
}

// the plane of x in REGISTERS (the forward's instance_norm_cached_kernel): x is read once, dy twice, dx written once, all as 16-byte
// pieces (the streaming form above reads a 256 x 320 plane five times with 4-byte loads: 339 us against ~40)
template <int THREADS, int VPT>
__global__ __launch_bounds__(THREADS) void instance_norm_backward_cached_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                                float* __restrict__ dx, int plane_size, float eps, int relu) {
  __shared__ float red[THREADS / 64];
  const size_t base = (size_t)blockIdx.x * plane_size;
  const float4* src = reinterpret_cast<const float4*>(x + base);
  const float4* gsrc = reinterpret_cast<const float4*>(dy + base);
  float4* dst = reinterpret_cast<float4*>(dx + base);
  const int n4 = plane_size >> 2;
  float4 v[VPT];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int j = i * THREADS + threadIdx.x;
    v[i] = j < n4 ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float inv_n = 1.0f / (float)plane_size;
  const float mean = in_block_sum<THREADS>(s, red) * inv_n;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int j = i * THREADS + threadIdx.x;
    if (j < n4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(in_block_sum<THREADS>(q, red) * inv_n + eps);
  // the normalised values replace x in the registers
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    v[i].x = (v[i].x - mean) * rstd, v[i].y = (v[i].y - mean) * rstd;
    v[i].z = (v[i].z - mean) * rstd, v[i].w = (v[i].w - mean) * rstd;
  }
  const bool rl = relu != 0;
  float sg = 0.0f, sgx = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int j = i * THREADS + threadIdx.x;
    if (j < n4) {
      float4 g = gsrc[j];
      g.x = (rl && !(v[i].x > 0.0f)) ? 0.0f : g.x, g.y = (rl && !(v[i].y > 0.0f)) ? 0.0f : g.y;
      g.z = (rl && !(v[i].z > 0.0f)) ? 0.0f : g.z, g.w = (rl && !(v[i].w > 0.0f)) ? 0.0f : g.w;
      sg += (g.x + g.y) + (g.z + g.w);
      sgx += (g.x * v[i].x + g.y * v[i].y) + (g.z * v[i].z + g.w * v[i].w);
    }
  }
  const float mg = in_block_sum<THREADS>(sg, red) * inv_n;
  const float mgx = in_block_sum<THREADS>(sgx, red) * inv_n;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int j = i * THREADS + threadIdx.x;
    if (j < n4) {
      float4 g = gsrc[j];
      g.x = (rl && !(v[i].x > 0.0f)) ? 0.0f : g.x, g.y = (rl && !(v[i].y > 0.0f)) ? 0.0f : g.y;
      g.z = (rl && !(v[i].z > 0.0f)) ? 0.0f : g.z, g.w = (rl && !(v[i].w > 0.0f)) ? 0.0f : g.w;
      float4 o;
      o.x = rstd * (g.x - mg - v[i].x * mgx), o.y = rstd * (g.y - mg - v[i].y * mgx);
      o.z = rstd * (g.z - mg - v[i].z * mgx), o.w = rstd * (g.w - mg - v[i].w * mgx);
      dst[j] = o;
    }
  }
}

extern "C" int mnerf_instance_norm_backward(const float* x, const float* dy, float* dx, int64_t planes, int64_t plane_size, float eps,
                                            int32_t relu, void* stream) {
  MNERF_REQUIRE(planes >= 0 && planes <= 0x7fffffffLL && plane_size >= 1 && plane_size <= 0x7fffffffLL, MNERF_E_RANGE,
                "mnerf_instance_norm_backward: planes=%lld plane_size=%lld", (long long)planes, (long long)plane_size);
  MNERF_REQUIRE(eps >= 0.0f, MNERF_E_RANGE, "mnerf_instance_norm_backward: eps=%g", (double)eps);
  if (planes == 0) return MNERF_OK;
  MNERF_REQUIRE(x && dy && dx, MNERF_E_NULL, "mnerf_instance_norm_backward: NULL buffer");
  hipStream_t st = (hipStream_t)stream;
  const int n = (int)plane_size;
  const dim3 grid((unsigned)planes);
  const bool vec = (n & 3) == 0 && mnerf_aligned16(x) && mnerf_aligned16(dy) && mnerf_aligned16(dx);
#define INB_LAUNCH(T, V) hipLaunchKernelGGL((instance_norm_backward_cached_kernel<T, V>), grid, dim3(T), 0, st, x, dy, dx, n, eps, relu)
  if (vec && n <= 256 * 8 * 4) INB_LAUNCH(256, 8);
  else if (vec && n <= 256 * 20 * 4) INB_LAUNCH(256, 20);
  else if (vec && n <= 512 * 40 * 4) INB_LAUNCH(512, 40);
  else hipLaunchKernelGGL(instance_norm_backward_kernel, grid, dim3(512), 0, st, x, dy, dx, n, eps, relu);
#undef INB_LAUNCH
  return mnerf_check_launch("mnerf_instance_norm_backward");
}

// ---------------------------------------------------------------------------------------------------------------- bilinear 2x
// F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) of the up-sampler's training path (superres.py:37) and its
// backward - what ATen's upsample_bilinear2d_out_frame / _backward_nhwc_out_frame did in 335 + 191 us per call; both are separable
// two-tap filters with fixed weights: output 2 i reads (i - 1, i) with (0.25, 0.75), output 2 i + 1 reads (i, i + 1) with (0.75,
// 0.25), the source index clamped at both ends (torch clamps the source coordinate at 0 and the upper neighbour at n - 1).
__device__ __forceinline__ void up2_taps(int o, int n, int& i0, int& i1, float& w0, float& w1) {
  const int i = o >> 1;
  if (o & 1) {
    i0 = i, i1 = min(i + 1, n - 1), w0 = 0.75f, w1 = 0.25f;
  } else if (i == 0) {
    i0 = 0, i1 = 0, w0 = 1.0f, w1 = 0.0f;
  } else {
    i0 = i - 1, i1 = i, w0 = 0.25f, w1 = 0.75f;
  }
}
__global__ __launch_bounds__(256) void upsample_bilinear2x_kernel(const float* __restrict__ in, const float* __restrict__ add, float* __restrict__ out,
                                                                  int h, int w, long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // index into out [planes][2 h][2 w]
  if (e >= total) return;
  const int W2 = 2 * w, H2 = 2 * h;
  const int x = (int)(e % W2), y = (int)((e / W2) % H2);
  const long long plane = e / ((long long)W2 * H2);
  int y0, y1, x0, x1;
  float wy0, wy1, wx0, wx1;
  up2_taps(y, h, y0, y1, wy0, wy1);
  up2_taps(x, w, x0, x1, wx0, wx1);
  const float* p = in + plane * (long long)h * w;
  const float v = wy0 * (wx0 * p[(long long)y0 * w + x0] + wx1 * p[(long long)y0 * w + x1]) +
                  wy1 * (wx0 * p[(long long)y1 * w + x0] + wx1 * p[(long long)y1 * w + x1]);
  out[e] = add ? v + add[e] : v;
}
// din[i][j] = sum over the (up to) 4 x 4 outputs that read input (i, j): rows 2 i - 1 .. 2 i + 2 with (0.25, 0.75 | 1 at i = 0,
// 0.75 | 1 at i = n - 1, 0.25)
__device__ __forceinline__ void up2_adjoint(int i, int n, float (&wt)[4]) {
  wt[0] = i >= 1 ? 0.25f : 0.0f;
  wt[1] = i == 0 ? 1.0f : 0.75f;
  wt[2] = i == n - 1 ? 1.0f : 0.75f;
  wt[3] = i <= n - 2 ? 0.25f : 0.0f;
}
__global__ __launch_bounds__(256) void upsample_bilinear2x_backward_kernel(const float* __restrict__ dout, float* __restrict__ din, int h, int w,
                                                                           long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // index into din [planes][h][w]
  if (e >= total) return;
  const int j = (int)(e % w), i = (int)((e / w) % h);
  const long long plane = e / ((long long)w * h);
  float wy[4], wx[4];
  up2_adjoint(i, h, wy);
  up2_adjoint(j, w, wx);
  const float* p = dout + plane * (long long)(4 * h) * w;
  float s = 0.0f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int y = 2 * i - 1 + a;
    if (wy[a] == 0.0f) continue;
    float r = 0.0f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int x = 2 * j - 1 + b;
      if (wx[b] != 0.0f) r += wx[b] * p[(long long)y * (2 * w) + x];
    }
    s += wy[a] * r;
  }
  din[e] = s;
}

extern "C" int mnerf_upsample_bilinear2x(const float* in, const float* add, float* out, int64_t planes, int32_t h, int32_t w, void* stream) {
  MNERF_REQUIRE(planes >= 0 && h >= 1 && w >= 1, MNERF_E_RANGE, "mnerf_upsample_bilinear2x: planes=%lld h=%d w=%d", (long long)planes, h, w);
  if (planes == 0) return MNERF_OK;
  MNERF_REQUIRE(in && out, MNERF_E_NULL, "mnerf_upsample_bilinear2x: NULL buffer");
  const long long total = planes * 4ll * h * w;
  hipLaunchKernelGGL(upsample_bilinear2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, add, out, h, w, total);
  return mnerf_check_launch("mnerf_upsample_bilinear2x");
}
extern "C" int mnerf_upsample_bilinear2x_backward(const float* dout, float* din, int64_t planes, int32_t h, int32_t w, void* stream) {
  MNERF_REQUIRE(planes >= 0 && h >= 1 && w >= 1, MNERF_E_RANGE, "mnerf_upsample_bilinear2x_backward: planes=%lld h=%d w=%d", (long long)planes, h,
                w);
  if (planes == 0) return MNERF_OK;
  MNERF_REQUIRE(dout && din, MNERF_E_NULL, "mnerf_upsample_bilinear2x_backward: NULL buffer");
  const long long total = planes * (long long)h * w;
  hipLaunchKernelGGL(upsample_bilinear2x_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, din, h, w,
                     total);
  return mnerf_check_launch("mnerf_upsample_bilinear2x_backward");
}
