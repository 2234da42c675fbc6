// Backward of the GMFlow backbone / up-sampler convolutions (round 6) — the training path's last library dependency.
//
// Replaces what autograd runs behind /root/reference/models/gmflow/backbone.py:6-122 and superres.py:5-38 when
// /root/reference/coach.py:215-243 calls loss.backward(): the data gradient and the weight gradient of Conv2d(c_in, c_out, k,
// stride, padding = k / 2) for the shapes conv.hip builds (c_in, c_out multiples of 32, k = 1 / 3, stride 1 / 2), NCHW fp32.
// Until round 6 these ran on MIOpen (igemm_wrw / igemm_bwd kernels + its layout transposes, 3.8 ms of a 35 ms iteration).
//
// Arithmetic: EXACT fp32 products on v_mfma_f32_32x32x2_f32 with fp32 accumulation - gradients span many binades (1e-9 .. 1e-2
// in one tensor), and the split-fp16 path of the forward kernels needs one power-of-two gain per operand tensor; the exact-f32
// matrix instruction needs no range management at all and the two kernels are bound by it (157 TFLOP/s peak; the backward of
// all 14 convolutions at the DTU shape is ~260 GFLOP).  No LDS: both operands are read with coalesced (data gradient) or
// 16-byte (weight gradient) loads straight into the lanes the instruction wants them in.
//
//   data gradient   dX[ci][y][x] = sum_{tap, co} W[co][ci][tap] dY[co][(y + p - ky) / s][(x + p - kx) / s]
//     a wave owns 32 positions of one output row that share the x parity class (stride 2) x ALL input channels:
//     D[ci 32-block][position] += A[ci][co pair] B[co pair][position], A = the weights transposed to [tap][co][ci] (the caller
//     does that once per step: one permute), B = dY rows; per K-step (two output channels) one coalesced load of B, CIB of A,
//     CIB matrix instructions.  Taps whose (y + p - ky) is not a multiple of the stride or falls outside dY are skipped
//     (wave-uniform), columns outside dY contribute zeros.
//   weight gradient dW[co][ci][ky][kx] = sum_{n, yo, xo} dY[co][yo][xo] X[ci][s yo + ky - p][s xo + kx - p]
//     a wave owns (a chunk of dY rows, one filter row ky, one 32-block of output channels) x all kx x all input channels:
//     D[co][ci] per kx; the K dimension is positions, eight per step: lane (channel, k-half) reads FOUR consecutive positions
//     of its channel's row with one 16-byte load (the order of K inside a step is free as long as both operands agree), four
//     matrix instructions consume them.  Partial sums per chunk go to a workspace, a second kernel adds the chunks in a
//     fixed order (bit-reproducible) and writes torch's [co][ci][ky][kx].
#include "common.hpp"

typedef float cb_f16 __attribute__((ext_vector_type(16)));

struct ConvBwdGeom {
  int n, c_in, c_out, h, w, ho, wo, k, s, pad;
};

__device__ __forceinline__ cb_f16 cb_mfma(float a, float b, cb_f16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ============================================================================ data gradient
template <int CIB>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wt, float* __restrict__ dx,
                                                         ConvBwdGeom G, int nseg, long long n_waves) {
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (gw >= n_waves) return;
  const int nn = lane & 31, kk = lane >> 5;
  // wave -> (image, row y, x parity px, segment of 32 positions of that parity)
  long long t = gw;
  const int seg = (int)(t % nseg);
  t /= nseg;
  const int px = (int)(t % G.s);
  t /= G.s;
  const int y = (int)(t % G.h);
  const int img = (int)(t / G.h);
  const int xq = seg * 32 + nn;          // index inside the parity class
  const int x = G.s * xq + px;           // this lane's column of dX
  cb_f16 acc[CIB];
#pragma unroll
  for (int c = 0; c < CIB; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.0f;
  const size_t plane_o = (size_t)G.ho * G.wo;
  for (int ky = 0; ky < G.k; ++ky) {
    const int ty = y + G.pad - ky;
    if (ty < 0 || ty % G.s != 0) continue;  // (wave-uniform)
    const int yo = ty / G.s;
    if (yo >= G.ho) continue;
    for (int kx = 0; kx < G.k; ++kx) {
      const int tx0 = px + G.pad - kx;
      if (((tx0 % G.s) + G.s) % G.s != 0) continue;  // (wave-uniform: the parity class fixes which kx contribute)
      // xo = (x + pad - kx) / s = xq + tx0 / s   (tx0 a multiple of s, possibly negative)
      const int xo = xq + (tx0 >= 0 ? tx0 / G.s : -((-tx0) / G.s));
      const bool ok = xo >= 0 && xo < G.wo;
      const float* brow = dy + ((size_t)img * G.c_out + kk) * plane_o + (size_t)yo * G.wo + (ok ? xo : 0);
      const float* arow = wt + ((size_t)(ky * G.k + kx) * G.c_out + kk) * G.c_in + nn;
      for (int co = 0; co < G.c_out; co += 8) {  // four K-steps per trip (c_out is a multiple of 32): their loads go out together
        float b[4], a[4][CIB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          b[u] = brow[(size_t)(co + 2 * u) * plane_o];
#pragma unroll
          for (int c = 0; c < CIB; ++c) a[u][c] = arow[(size_t)(co + 2 * u) * G.c_in + c * 32];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float bu = ok ? b[u] : 0.0f;
#pragma unroll
          for (int c = 0; c < CIB; ++c) acc[c] = cb_mfma(a[u][c], bu, acc[c]);
        }
      }
    }
  }
  // D[m = input channel][n = position]: register i of lane (nn, kk) is row 8 (i / 4) + 4 kk + i % 4
  if (x < G.w) {
    float* out = dx + ((size_t)img * G.c_in * G.h + y) * G.w + x;
#pragma unroll
    for (int c = 0; c < CIB; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = c * 32 + 8 * (i >> 2) + 4 * kk + (i & 3);
        out[(size_t)ci * G.h * G.w] = acc[c][i];
      }
  }
}

// ============================================================================ weight gradient
// four consecutive floats of a row starting at column x0 (any alignment), zeros outside [0, limit)
__device__ __forceinline__ float4 cb_load4(const float* row, int x0, int limit) {
  if (x0 >= 0 && x0 + 3 < limit) {
    float4 v;
    __builtin_memcpy(&v, row + x0, 16);  // (4-byte aligned 16-byte load)
    return v;
  }
  float4 v;
  v.x = x0 >= 0 && x0 < limit ? row[x0] : 0.0f;
  v.y = x0 + 1 >= 0 && x0 + 1 < limit ? row[x0 + 1] : 0.0f;
  v.z = x0 + 2 >= 0 && x0 + 2 < limit ? row[x0 + 2] : 0.0f;
  v.w = x0 + 3 >= 0 && x0 + 3 < limit ? row[x0 + 3] : 0.0f;
  return v;
}
// the same with a stride of two between the elements
__device__ __forceinline__ float4 cb_load4_s2(const float* row, int x0, int limit) {
  float4 v;
  v.x = x0 >= 0 && x0 < limit ? row[x0] : 0.0f;
  v.y = x0 + 2 >= 0 && x0 + 2 < limit ? row[x0 + 2] : 0.0f;
  v.z = x0 + 4 >= 0 && x0 + 4 < limit ? row[x0 + 4] : 0.0f;
  v.w = x0 + 6 >= 0 && x0 + 6 < limit ? row[x0 + 6] : 0.0f;
  return v;
}

template <int CIB, int KW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                         ConvBwdGeom G, int chunks, int rows_per_chunk, long long n_waves) {
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (gw >= n_waves) return;
  const int nn = lane & 31, kk = lane >> 5;
  const int cobs = G.c_out / 32, cigs = G.c_in / (32 * CIB);
  // wave -> (chunk, ky, input-channel group of CIB blocks, output-channel block): neighbours read the same dY / X rows
  long long t = gw;
  const int cob = (int)(t % cobs);
  t /= cobs;
  const int ci0 = (int)(t % cigs) * CIB * 32;
  t /= cigs;
  const int ky = (int)(t % G.k);
  const int chunk = (int)(t / G.k);
  cb_f16 acc[KW][CIB];
#pragma unroll
  for (int q = 0; q < KW; ++q)
#pragma unroll
    for (int c = 0; c < CIB; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[q][c][i] = 0.0f;
  const int row_end = min((chunk + 1) * rows_per_chunk, G.n * G.ho);
  for (int r = chunk * rows_per_chunk; r < row_end; ++r) {
    const int img = r / G.ho, yo = r - img * G.ho;
    const int yi = G.s * yo + ky - G.pad;
    if (yi < 0 || yi >= G.h) continue;  // (wave-uniform: the filter row looks at padding)
    const float* arow = dy + (((size_t)img * G.c_out + cob * 32 + nn) * G.ho + yo) * G.wo;
    const float* brow0 = x + (((size_t)img * G.c_in + ci0 + nn) * G.h + yi) * G.w;
    for (int x0 = 0; x0 < G.wo; x0 += 8) {
      const float4 a = cb_load4(arow, x0 + 4 * kk, G.wo);
#pragma unroll
      for (int q = 0; q < KW; ++q) {
        const int xi0 = G.s * (x0 + 4 * kk) + q - G.pad;
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
          const float* brow = brow0 + (size_t)c * 32 * G.h * G.w;
          const float4 b = G.s == 1 ? cb_load4(brow, xi0, G.w) : cb_load4_s2(brow, xi0, G.w);
          acc[q][c] = cb_mfma(a.x, b.x, acc[q][c]);
          acc[q][c] = cb_mfma(a.y, b.y, acc[q][c]);
          acc[q][c] = cb_mfma(a.z, b.z, acc[q][c]);
          acc[q][c] = cb_mfma(a.w, b.w, acc[q][c]);
        }
      }
    }
  }
  // D[m = output channel][n = input channel] -> part[chunk][tap][co][ci]
#pragma unroll
  for (int q = 0; q < KW; ++q) {
    float* dst = part + (((size_t)chunk * G.k * G.k + ky * G.k + q) * G.c_out + cob * 32) * G.c_in + ci0 + nn;
#pragma unroll
    for (int c = 0; c < CIB; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = 8 * (i >> 2) + 4 * kk + (i & 3);
        dst[(size_t)m * G.c_in + c * 32] = acc[q][c][i];
      }
  }
}

// dW[co][ci][ky][kx] = sum over the chunks, in chunk order
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int chunks, int taps,
                                                                int c_out, int c_in) {
  const int per_chunk = taps * c_out * c_in;
  const int e = blockIdx.x * 256 + threadIdx.x;  // index into [tap][co][ci]
  if (e >= per_chunk) return;
  float s = 0.0f;
  for (int c = 0; c < chunks; ++c) s += part[(size_t)c * per_chunk + e];
  const int ci = e % c_in, co = (e / c_in) % c_out, tap = e / (c_in * c_out);
  dw[((size_t)co * c_in + ci) * taps + tap] = s;
}

// ============================================================================ host
static int cb_geom(const char* who, ConvBwdGeom& G, int32_t n, int32_t c_in, int32_t c_out, int32_t h, int32_t w, int32_t ksize,
                   int32_t stride) {
  MNERF_REQUIRE(n >= 0 && h >= 1 && w >= 1, MNERF_E_RANGE, "%s: n=%d h=%d w=%d", who, n, h, w);
  MNERF_REQUIRE(c_in >= 32 && c_in <= 128 && c_in % 32 == 0 && c_out >= 32 && c_out <= 128 && c_out % 32 == 0, MNERF_E_UNSUPPORTED,
                "%s: channels %d -> %d (multiples of 32 up to 128 are built)", who, c_in, c_out);
  MNERF_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), MNERF_E_UNSUPPORTED, "%s: ksize=%d stride=%d", who, ksize, stride);
  G.n = n, G.c_in = c_in, G.c_out = c_out, G.h = h, G.w = w, G.k = ksize, G.s = stride, G.pad = ksize / 2;
  G.ho = (h + 2 * G.pad - ksize) / stride + 1;
  G.wo = (w + 2 * G.pad - ksize) / stride + 1;
  return MNERF_OK;
}

extern "C" int mnerf_conv2d_backward_data(const float* dy, const float* w_tap_major, float* dx, int32_t n_img, int32_t c_in, int32_t c_out,
                                          int32_t h_in, int32_t w_in, int32_t ksize, int32_t stride, void* stream) {
  const char* who = "mnerf_conv2d_backward_data";
  ConvBwdGeom G;
  if (const int rc = cb_geom(who, G, n_img, c_in, c_out, h_in, w_in, ksize, stride)) return rc;
  if (n_img == 0) return MNERF_OK;
  MNERF_REQUIRE(dy && w_tap_major && dx, MNERF_E_NULL, "%s: NULL buffer", who);
  const int nseg = ((w_in + stride - 1) / stride + 31) / 32;
  const long long waves = (long long)n_img * h_in * stride * nseg;
  const dim3 grid((unsigned)((waves + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  switch (c_in / 32) {
    case 1: hipLaunchKernelGGL(conv_dgrad_kernel<1>, grid, dim3(256), 0, st, dy, w_tap_major, dx, G, nseg, waves); break;
    case 2: hipLaunchKernelGGL(conv_dgrad_kernel<2>, grid, dim3(256), 0, st, dy, w_tap_major, dx, G, nseg, waves); break;
    case 3: hipLaunchKernelGGL(conv_dgrad_kernel<3>, grid, dim3(256), 0, st, dy, w_tap_major, dx, G, nseg, waves); break;
    default: hipLaunchKernelGGL(conv_dgrad_kernel<4>, grid, dim3(256), 0, st, dy, w_tap_major, dx, G, nseg, waves); break;
  }
  return mnerf_check_launch(who);
}

// chunks of dY rows: enough waves for the chip, partial sums of at most ~32 MiB
static void cb_chunks(const ConvBwdGeom& G, int& chunks, int& rpc) {
  const long long rows = (long long)G.n * G.ho;
  const long long per_chunk_bytes = (long long)G.k * G.k * G.c_out * G.c_in * 4;
  long long want = 4096 / ((long long)G.k * (G.c_out / 32) * (G.c_in == 128 ? 2 : 1));
  const long long cap = (32ll << 20) / per_chunk_bytes;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  if (want > rows) want = rows;
  rpc = (int)((rows + want - 1) / want);
  chunks = (int)((rows + rpc - 1) / rpc);
}

extern "C" size_t mnerf_conv2d_backward_weight_workspace_bytes(int32_t n_img, int32_t c_in, int32_t c_out, int32_t h_in, int32_t w_in,
                                                              int32_t ksize, int32_t stride) {
  ConvBwdGeom G;
  if (cb_geom("mnerf_conv2d_backward_weight_workspace_bytes", G, n_img, c_in, c_out, h_in, w_in, ksize, stride) || n_img == 0) return 0;
  int chunks, rpc;
  cb_chunks(G, chunks, rpc);
  return (size_t)chunks * G.k * G.k * G.c_out * G.c_in * sizeof(float);
}

extern "C" int mnerf_conv2d_backward_weight(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int32_t n_img,
                                            int32_t c_in, int32_t c_out, int32_t h_in, int32_t w_in, int32_t ksize, int32_t stride,
                                            void* stream) {
  const char* who = "mnerf_conv2d_backward_weight";
  ConvBwdGeom G;
  if (const int rc = cb_geom(who, G, n_img, c_in, c_out, h_in, w_in, ksize, stride)) return rc;
  MNERF_REQUIRE(dw, MNERF_E_NULL, "%s: dw is NULL", who);
  hipStream_t st = (hipStream_t)stream;
  if (n_img == 0) {
    const hipError_t e = hipMemsetAsync(dw, 0, (size_t)c_out * c_in * ksize * ksize * sizeof(float), st);
    return e == hipSuccess ? MNERF_OK : (int)e;
  }
  MNERF_REQUIRE(x && dy && workspace, MNERF_E_NULL, "%s: NULL buffer", who);
  int chunks, rpc;
  cb_chunks(G, chunks, rpc);
  const size_t need = (size_t)chunks * G.k * G.k * G.c_out * G.c_in * sizeof(float);
  MNERF_REQUIRE(workspace_bytes >= need, MNERF_E_RANGE, "%s: workspace %zu bytes < %zu", who, workspace_bytes, need);
  float* part = reinterpret_cast<float*>(workspace);
  // input-channel blocks per wave: 1 / 2 / 3 for 32 / 64 / 96 channels, two groups of 2 for 128 (3 x 4 x 16 accumulators would not fit)
  const int cib = c_in == 128 ? 2 : c_in / 32;
  const long long waves = (long long)chunks * G.k * (G.c_in / (32 * cib)) * (G.c_out / 32);
  const dim3 grid((unsigned)((waves + 3) / 4));
#define CB_WG(CIB_, KW_) hipLaunchKernelGGL((conv_wgrad_kernel<CIB_, KW_>), grid, dim3(256), 0, st, x, dy, part, G, chunks, rpc, waves)
  if (ksize == 3) {
    switch (cib) {
      case 1: CB_WG(1, 3); break;
      case 2: CB_WG(2, 3); break;
      default: CB_WG(3, 3); break;
    }
  } else {
    switch (cib) {
      case 1: CB_WG(1, 1); break;
      case 2: CB_WG(2, 1); break;
      default: CB_WG(3, 1); break;
    }
  }
#undef CB_WG
  const int per_chunk = G.k * G.k * G.c_out * G.c_in;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((per_chunk + 255) / 256)), dim3(256), 0, st, part, dw, chunks, G.k * G.k, G.c_out,
                     G.c_in);
  return mnerf_check_launch(who);
}
