// Convolutions of the GMFlow CNN backbone and feature up-sampler as implicit GEMMs on the matrix cores (gfx950).
//
// Replaces the library convolutions behind (paths relative to /root/reference/models/gmflow):
//   backbone.py:6-36, 39-122   ResidualBlock / CNNEncoder: 3x3 (stride 1 / 2) and 1x1 (stride 1 / 2) convolutions,
//                              64 / 96 / 128 channels (the 7x7 stem on 3 channels stays a library call)
//   superres.py:5-38           the up-sampler's 3x3 convolutions, incl. the one that follows a nearest 2x up-sampling
//                              (read through the up-sampling here: the 4x larger tensor is never written) and its
//                              LeakyReLU(0.2)
// MIOpen runs these in fp32 on v_mfma_f32_32x32x2_f32 at ~100 TFLOP/s (65 % of that instruction's peak).  Here the
// products are fp32-grade split-fp16 (split_f16.hpp: two fp16 terms per operand, three products per MAC on
// v_mfma_f32_32x32x16_f16 - the arithmetic of the decoder and of the encoder block kernel).
//
// Formulation: Y^T[out channel, pixel] = W[out, (tap, in channel)] . X^T[(tap, in channel), pixel] - the transposed
// chain again: a wave owns 2 x 32 output pixels (lane & 31 = pixel, consecutive along the row: coalesced NCHW
// accesses), accumulators hold all output channels of its pixels.  One K16-step = 16 input channels of one filter tap;
// lane (pixel, half) loads its 8 channels of that tap straight from global memory (predicated: zero padding),
// one step ahead of their use, and splits them with ONE power-of-two gain per input TENSOR - its largest magnitude,
// left in an absmax region (common.hpp) by the kernel that produced it (mnerf_instance_norm, this kernel, mnerf_absmax); the IN
// output of a plane is bounded by sqrt(plane size), so the typical value keeps >= 21 significant bits.
// Weights: packed by the host as A-operand fragments (matchnerf_amd/gmflow.py, pack_conv), streamed in segments of
// whole K16-step pairs (<= 36 KiB) through a 2 x 36 KiB LDS double buffer by LDS-DMA, one barrier per segment, two
// workgroups of four waves per CU.
#include "split_f16.hpp"

#define CONV_NW 4
#ifndef CONV_EXP
#define CONV_EXP 0  // removal experiments (wrong results, meaningful times; tools/exp/conv_removal.sh): 1 no matrix instructions,
#endif              // 2 no operand loads, 3 no operand split, 4 no weight requests / segment barriers, 5 no tap geometry
#define CONV_BUF_BYTES 36864u  // one LDS weight buffer (36 KiB: six K16-steps x three 32-row blocks)

#if CONV_EXP == 1
__device__ __forceinline__ f32x16 conv_nomfma(f16x8 a, f16x8 b, f32x16 c) {
  asm volatile("" : "+v"(a), "+v"(b));
  return c;
}
#define CONV_MFMA conv_nomfma
#else
#define CONV_MFMA mfma16h
#endif
#ifdef CONV_TIMELINE
// debug build (tools/exp/conv_timeline.py): s_memtime stamps of wave 0 of a few workgroups at the phase boundaries of every
// iteration: [workgroup slot][iteration][point]  0 top | 1 operands arrived + split | 2 next loads issued | 3 matrix instructions
// issued | (last iteration of a segment) 4 weight requests landed | 5 behind the barrier
#define CONV_TL_POINTS 6
#define CONV_TL_ITERS 64
#define CONV_TL_STAMP(k)                                                                                       \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (tl_out && iter < CONV_TL_ITERS) tl_out[iter * CONV_TL_POINTS + (k)] = __builtin_amdgcn_s_memtime();    \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#else
#define CONV_TL_STAMP(k) do {} while (0)
#endif
struct ConvParams {
  unsigned long long* tl;  // timeline build only (NULL otherwise)
  const float* in;
  const float* wstream;
  const float* bias;
  float* out;
  const float* in_absmax;
  float* out_absmax;
  const float* add_up;  // [n_img, c_out, h_out / 2, w_out / 2] or NULL: its bilinear 2x up-sampling is added to the result
  int n_img, c_in, h_in, w_in, h_out, w_out;
  int ksize, stride, up;      // up = 1: the input is read through a nearest 2x up-sampling
  long long sc, sy, sx, si;   // element strides of the stored input: channel, row, column, image
  int ew, seg_steps, n_seg;
  float leaky;                // LeakyReLU slope applied to the result (1 = none)
  int out_layout;             // MNERF_CONV_OUT_*: NCHW, channel-last tokens, pair-major channel-last
  const float* add_cl;        // [h_out * w_out][c_out] added to a channel-last result (position tile), or NULL
};

// NMB: 32-row blocks of output channels; TPW: 32-pixel tiles per wave (2, or 1 when that is what it takes to give
// every CU a workgroup); CL: the input is stored channel-last (a lane's 8 channels are two float4)
template <int NMB, int TPW, bool CL>
__global__ __launch_bounds__(CONV_NW * 64, 2) void conv_kernel(ConvParams P) {
  extern __shared__ __attribute__((aligned(16))) float conv_smem[];
  const unsigned buf0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)conv_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hl = lane >> 5;
  const int seg_units = P.seg_steps * NMB;
  const int pieces = seg_units * 2;  // 1 KiB each

  auto stage = [&](int seg) {
    const float* src = P.wstream + (size_t)seg * (size_t)seg_units * 512 + lane * 4;
    const unsigned dst = buf0 + (unsigned)(seg & 1) * CONV_BUF_BYTES;
    glds_segment(src, dst, pieces, wave, CONV_NW);
  };
  stage(0);

  // ---- this lane's output pixels
  const int hw_out = P.h_out * P.w_out;
  const long long n_pix = (long long)P.n_img * hw_out;
  int oy[TPW], ox[TPW], oimg[TPW];
  long long ibase[TPW], obase[TPW];
  bool live[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const long long p = ((long long)blockIdx.x * CONV_NW + wave) * (32 * TPW) + t * 32 + n;
    live[t] = p < n_pix;
    const long long pc = live[t] ? p : n_pix - 1;
    const int img = (int)(pc / hw_out), rem = (int)(pc - (long long)img * hw_out);
    oimg[t] = img;
    oy[t] = rem / P.w_out;
    ox[t] = rem - oy[t] * P.w_out;
    ibase[t] = (long long)img * P.si + (long long)(8 * hl) * P.sc;
    obase[t] = (long long)img * (32 * NMB) * hw_out + rem;
  }
  const int pad = P.ksize >> 1;
  const int h_eff = P.h_in << P.up, w_eff = P.w_in << P.up;
  const int csteps = P.c_in >> 4, n_steps = P.ksize * P.ksize * csteps;

  const int eg = gain_exp(mnerf_absmax_read(P.in_absmax));
  const float mult = pow2i(eg);

  // ---- operand pipeline: an iteration covers TWO K16-steps (32 input channels of one tap); the values of iteration
  // i+1 are requested before the 12..24 x 2 matrix instructions of iteration i, so an L2 round trip fits under them.
  // Loads are unconditional from a clamped address and zeroed by a select (padding): no branches in the loop.
  int tap = 0, cpair = 0;
  long long toff[TPW];
  bool tok[TPW];
  auto tap_geometry = [&]() {
    const int dy = tap / P.ksize - pad, dx = tap % P.ksize - pad;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int yy = oy[t] * P.stride + dy, xx = ox[t] * P.stride + dx;
      tok[t] = live[t] && yy >= 0 && yy < h_eff && xx >= 0 && xx < w_eff;
      const int yc = min(max(yy, 0), h_eff - 1), xc = min(max(xx, 0), w_eff - 1);
      toff[t] = ibase[t] + (long long)(yc >> P.up) * P.sy + (long long)(xc >> P.up) * P.sx;
    }
  };
  const int cpairs = P.c_in >> 5, n_iter = n_steps >> 1;
  float vn[TPW][16];  // [tile][K16-step of the pair * 8 + j], as loaded: the padding mask is applied at the USE
  bool vok[TPW];      // (a select next to the load would make the wave wait for the data one iteration early)
  auto load_pair = [&]() {
    if (CONV_EXP == 2) return;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const float* src = P.in + toff[t] + (long long)(32 * cpair) * P.sc;
      vok[t] = tok[t];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if constexpr (CL) {
          const float4 a4 = *reinterpret_cast<const float4*>(src + 16 * u), b4 = *reinterpret_cast<const float4*>(src + 16 * u + 4);
          vn[t][8 * u + 0] = a4.x; vn[t][8 * u + 1] = a4.y; vn[t][8 * u + 2] = a4.z; vn[t][8 * u + 3] = a4.w;
          vn[t][8 * u + 4] = b4.x; vn[t][8 * u + 5] = b4.y; vn[t][8 * u + 6] = b4.z; vn[t][8 * u + 7] = b4.w;
        } else {
          // (round 6) one wave-uniform 64-bit base per channel (scalar adds) + this lane's 32-bit byte offset: the global load's
          // scalar-base form, instead of a 64-bit vector address per load (16 x TPW x ~4 vector instructions per iteration:
          // half of this kernel's vector instructions by the counters).  The host checks that the input is below 4 GiB.
          typedef const char __attribute__((address_space(1)))* gchar_cptr;   // (explicitly GLOBAL: behind the asm below a generic
          typedef const float __attribute__((address_space(1)))* gfloat_cptr;  //  pointer would be read with flat loads)
          gchar_cptr cb = (gchar_cptr)(size_t)P.in + ((long long)(32 * cpair + 16 * u) * P.sc) * 4;
          const unsigned lo = (unsigned)toff[t] * 4u;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            gchar_cptr cj = cb + ((long long)j * P.sc) * 4;
            asm volatile("" : "+s"(cj));  // (opaque and in scalar registers: hipcc otherwise re-associates to (base + lane) + channel)
            vn[t][8 * u + j] = *(gfloat_cptr)(cj + (size_t)lo);
          }
        }
      }
    }
    if (++cpair == cpairs) {
      cpair = 0;
      if (++tap < P.ksize * P.ksize && CONV_EXP != 5) tap_geometry();
    }
  };
  tap_geometry();
  load_pair();

  f32x16 acc[TPW][NMB];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int m = 0; m < NMB; ++m) acc[t][m] = (f32x16)(0.0f);

  segment_wait();
  __syncthreads();

  int iter = 0;
  const int seg_iters = P.seg_steps >> 1;
#ifdef CONV_TIMELINE
  // workgroups 0, 1, g/2, g/2 + 1 (two co-resident pairs), wave 0, lane 0
  const int tl_slot = blockIdx.x < 2 ? (int)blockIdx.x : (blockIdx.x == gridDim.x / 2 ? 2 : (blockIdx.x == gridDim.x / 2 + 1 ? 3 : -1));
  unsigned long long* tl_out = (P.tl && tl_slot >= 0 && tid == 0) ? P.tl + (size_t)tl_slot * CONV_TL_ITERS * CONV_TL_POINTS : nullptr;
#endif
  for (int seg = 0; seg < P.n_seg; ++seg) {
    const unsigned cur = buf0 + (unsigned)(seg & 1) * CONV_BUF_BYTES;
    for (int it = 0; it < seg_iters; ++it, ++iter) {
      CONV_TL_STAMP(0);
      PartsH b[2][TPW];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          float v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = vok[t] ? vn[t][8 * u + j] : 0.0f;
          if (CONV_EXP == 3) {
            b[u][t].hi = __builtin_bit_cast(f16x8, (u32x4){__float_as_uint(v8[0]), __float_as_uint(v8[1]), __float_as_uint(v8[2]), __float_as_uint(v8[3])});
            b[u][t].lo = __builtin_bit_cast(f16x8, (u32x4){__float_as_uint(v8[4]), __float_as_uint(v8[5]), __float_as_uint(v8[6]), __float_as_uint(v8[7])});
          } else
          b[u][t] = split8h(v8, mult);
        }
      // The next weight segment is requested HERE, behind the wait for this iteration's operands and in front of the next
      // iteration's operand loads (round 6).  hipcc does not see the asm LDS-DMA: its `s_waitcnt vmcnt(0)` for the operands also
      // waits for every request in flight - asked for at the top of the segment, the requests were waited for right away, once per
      // segment; in this place they are older than the loads the next wait is for and arrive before them.
      CONV_TL_STAMP(1);
      if (CONV_EXP != 4 && it == 0 && seg + 1 < P.n_seg) stage(seg + 1);
      if (iter + 1 < n_iter) load_pair();
      CONV_TL_STAMP(2);
      lds_u32x4_cptr a = (lds_u32x4_cptr)(size_t)(cur + (unsigned)(2 * it * NMB) * H16_UNIT_BYTES) + lane;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < NMB; ++m) {
          const int unit = u * NMB + m;
          const f16x8 ah = __builtin_bit_cast(f16x8, a[unit * 128]), al = __builtin_bit_cast(f16x8, a[unit * 128 + 64]);
          // the two pixel tiles alternate: no matrix instruction has the accumulator of its predecessor
#pragma unroll
          for (int t = 0; t < TPW; ++t) acc[t][m] = CONV_MFMA(ah, b[u][t].lo, acc[t][m]);
#pragma unroll
          for (int t = 0; t < TPW; ++t) acc[t][m] = CONV_MFMA(al, b[u][t].hi, acc[t][m]);
#pragma unroll
          for (int t = 0; t < TPW; ++t) acc[t][m] = CONV_MFMA(ah, b[u][t].hi, acc[t][m]);
        }
      CONV_TL_STAMP(3);
    }
    if (CONV_EXP == 4) continue;
    segment_wait();   // this wave's pieces of the next segment (and its operand prefetch) have landed
#ifdef CONV_TIMELINE
    --iter;
    CONV_TL_STAMP(4);
#endif
    __syncthreads();  // ... everybody's; the current buffer is free
#ifdef CONV_TIMELINE
    CONV_TL_STAMP(5);
    ++iter;
#endif
  }

  // ---- epilogue: scale back, bias, LeakyReLU, largest magnitude, NCHW store (32 consecutive pixels per register)
  const float cm = pow2i(-(P.ew + eg));
  float* bias_lds = conv_smem;  // the weight buffers are free now
  for (int i = tid; i < 32 * NMB; i += CONV_NW * 64) bias_lds[i] = P.bias ? P.bias[i] : 0.0f;
  __syncthreads();
  float omax = 0.0f;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (!live[t]) continue;
    // F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) of `add_up` at this pixel (superres.py:37):
    // source index max(0.5 (dst + 0.5) - 0.5, 0), second tap clamped at the border - torch's formula and order
    const int hs = P.h_out >> 1, ws = P.w_out >> 1;
    const float sy = fmaxf(0.5f * ((float)oy[t] + 0.5f) - 0.5f, 0.0f), sx = fmaxf(0.5f * ((float)ox[t] + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < hs - 1 ? 1 : 0), x1 = x0 + (x0 < ws - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* up0 = P.add_up ? P.add_up + (long long)oimg[t] * (32 * NMB) * hs * ws : nullptr;
    // channel-last image slot: the image itself, or (pair i mod P, side i div P) of the cost volume's pair-major layout
    const int half_n = P.n_img >> 1;
    const int cl_slot = P.out_layout != MNERF_CONV_OUT_PAIR_MAJOR ? oimg[t] : (oimg[t] < half_n ? 2 * oimg[t] : 2 * (oimg[t] - half_n) + 1);
    const long long cl_base = ((long long)cl_slot * hw_out + (oy[t] * P.w_out + ox[t])) * (32 * NMB);
#pragma unroll
    for (int m = 0; m < NMB; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + 32 * m + 8 * g + 4 * hl);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
        float bb4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = 32 * m + 8 * g + 4 * hl + q;
          float v = acc[t][m][4 * g + q] * cm + bb[q];
          v = v < 0.0f ? v * P.leaky : v;
          if (up0) {
            const float* pl = up0 + (long long)ch * hs * ws;
            v = (ly0 * (lx0 * pl[y0 * ws + x0] + lx1 * pl[y0 * ws + x1]) + ly1 * (lx0 * pl[y1 * ws + x0] + lx1 * pl[y1 * ws + x1])) + v;
          }
          if (P.out_layout == MNERF_CONV_OUT_NCHW) {
            omax = fmaxf(omax, fabsf(v));
            P.out[obase[t] + (long long)ch * hw_out] = v;
          }
          bb4[q] = v;
        }
        if (P.out_layout != MNERF_CONV_OUT_NCHW) {  // channel-last: 16 bytes per lane and store
          const int c0 = 32 * m + 8 * g + 4 * hl;
          if (P.add_cl) {
            const float4 a4 = *reinterpret_cast<const float4*>(P.add_cl + (long long)(oy[t] * P.w_out + ox[t]) * (32 * NMB) + c0);
            bb4[0] += a4.x; bb4[1] += a4.y; bb4[2] += a4.z; bb4[3] += a4.w;
          }
          omax = fmaxf(fmaxf(omax, fmaxf(fabsf(bb4[0]), fabsf(bb4[1]))), fmaxf(fabsf(bb4[2]), fabsf(bb4[3])));
          *reinterpret_cast<float4*>(P.out + cl_base + c0) = make_float4(bb4[0], bb4[1], bb4[2], bb4[3]);
        }
      }
  }
  if (P.out_absmax) {  // one atomic per workgroup into the tensor's absmax region
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off, 64));
    __syncthreads();
    if (lane == 0) bias_lds[wave] = omax;
    __syncthreads();
    if (tid == 0) mnerf_absmax_merge(fmaxf(fmaxf(bias_lds[0], bias_lds[1]), fmaxf(bias_lds[2], bias_lds[3])), P.out_absmax);
  }
}

// largest |x| of a tensor into an absmax region (the caller zeroes it)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.0f;
  const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
  const bool vec = (reinterpret_cast<size_t>(x) & 15) == 0;
  if (vec) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 v = x4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  }
  for (long long i = (vec ? n4 * 4 : 0) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) mnerf_absmax_merge(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), out);
}

extern "C" int mnerf_absmax(const float* x, int64_t n, float* out, void* stream) {
  MNERF_REQUIRE(n >= 0, MNERF_E_RANGE, "mnerf_absmax: n=%lld", (long long)n);
  MNERF_REQUIRE(out, MNERF_E_NULL, "mnerf_absmax: out is NULL");
  if (n == 0) return MNERF_OK;
  MNERF_REQUIRE(x, MNERF_E_NULL, "mnerf_absmax: x is NULL");
  const long long blocks = (n + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)n, out);
  return mnerf_check_launch("mnerf_absmax");
}

static int conv_seg_steps(int n_steps, int nmb) {  // largest even divisor of n_steps whose units fit one LDS buffer
  int best = 2;
  for (int s = 2; s <= n_steps; s += 2)
    if (n_steps % s == 0 && s * nmb * H16_UNIT_BYTES <= (int)CONV_BUF_BYTES) best = s;
  return best;
}

extern "C" int64_t mnerf_conv_wstream_floats(int32_t c_in, int32_t c_out, int32_t ksize) {
  if (c_in < 32 || c_in % 32 || c_out < 32 || c_out % 32 || ksize < 1) return 0;
  return (int64_t)ksize * ksize * (c_in / 16) * (c_out / 32) * (H16_UNIT_BYTES / 4);
}

extern "C" int mnerf_conv2d(const mnerf_conv* cv, const float* in, int32_t in_channels_last, int32_t upsample2x,
                            const float* in_absmax, const float* add_bilinear2x, const float* add_channel_last,
                            float* out, int32_t out_layout, float* out_absmax, int32_t n_img, int32_t h_in,
                            int32_t w_in, void* stream) {
  const char* who = "mnerf_conv2d";
  MNERF_REQUIRE(cv, MNERF_E_NULL, "%s: cv is NULL", who);
  MNERF_REQUIRE(cv->c_in >= 32 && cv->c_in % 32 == 0 && (cv->c_out == 64 || cv->c_out == 96 || cv->c_out == 128) &&
                    (cv->ksize == 1 || cv->ksize == 3) && (cv->stride == 1 || cv->stride == 2),
                MNERF_E_UNSUPPORTED, "%s: c_in=%d c_out=%d ksize=%d stride=%d (built: c_in %% 32 == 0, c_out 64/96/128, 1x1 / 3x3, stride 1 / 2)",
                who, cv->c_in, cv->c_out, cv->ksize, cv->stride);
  MNERF_REQUIRE(n_img >= 0 && h_in >= 1 && w_in >= 1, MNERF_E_RANGE, "%s: n_img=%d h_in=%d w_in=%d", who, n_img, h_in, w_in);
  MNERF_REQUIRE(cv->leaky_slope >= 0.0f, MNERF_E_RANGE, "%s: leaky_slope=%g", who, (double)cv->leaky_slope);
  if (n_img == 0) return MNERF_OK;
  MNERF_REQUIRE(cv->wstream && in && out && in_absmax, MNERF_E_NULL, "%s: NULL buffer", who);
  MNERF_REQUIRE(mnerf_aligned16(cv->wstream), MNERF_E_ALIGN, "%s: wstream must be 16-byte aligned", who);
  MNERF_REQUIRE(cv->wstream_floats == mnerf_conv_wstream_floats(cv->c_in, cv->c_out, cv->ksize), MNERF_E_RANGE,
                "%s: wstream has %lld floats, expected %lld", who, (long long)cv->wstream_floats,
                (long long)mnerf_conv_wstream_floats(cv->c_in, cv->c_out, cv->ksize));
  const int up = upsample2x ? 1 : 0, pad = cv->ksize / 2;
  ConvParams p;
  p.in = in;
  p.wstream = cv->wstream;
  p.bias = cv->bias;
  p.out = out;
  p.tl = nullptr;
#ifdef CONV_TIMELINE
  if (const char* e = getenv("MNERF_CONV_TL")) p.tl = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 16));
#endif
  p.in_absmax = in_absmax;
  p.out_absmax = out_absmax;
  p.add_up = add_bilinear2x;
  p.n_img = n_img;
  p.c_in = cv->c_in;
  p.h_in = h_in;
  p.w_in = w_in;
  p.h_out = ((h_in << up) + 2 * pad - cv->ksize) / cv->stride + 1;
  p.w_out = ((w_in << up) + 2 * pad - cv->ksize) / cv->stride + 1;
  MNERF_REQUIRE(!add_bilinear2x || (p.h_out % 2 == 0 && p.w_out % 2 == 0), MNERF_E_RANGE,
                "%s: add_bilinear2x needs an even output size, got %dx%d", who, p.h_out, p.w_out);
  p.ksize = cv->ksize;
  p.stride = cv->stride;
  p.up = up;
  if (in_channels_last) {
    p.sc = 1; p.sx = cv->c_in; p.sy = (long long)w_in * cv->c_in; p.si = (long long)h_in * w_in * cv->c_in;
  } else {
    p.sx = 1; p.sy = w_in; p.sc = (long long)h_in * w_in; p.si = (long long)h_in * w_in * cv->c_in;
    // (the kernel addresses a lane's pixel with a 32-bit byte offset next to a scalar channel base)
    MNERF_REQUIRE((long long)n_img * p.si * 4 < (1ll << 32), MNERF_E_UNSUPPORTED, "%s: an NCHW input of %lld bytes (built: below 4 GiB)", who,
                  (long long)n_img * p.si * 4);
  }
  p.ew = cv->ew;
  const int nmb = cv->c_out / 32, n_steps = cv->ksize * cv->ksize * (cv->c_in / 16);
  p.seg_steps = conv_seg_steps(n_steps, nmb);
  p.n_seg = n_steps / p.seg_steps;
  p.leaky = cv->leaky_slope;
  MNERF_REQUIRE(out_layout == MNERF_CONV_OUT_NCHW || out_layout == MNERF_CONV_OUT_CHANNEL_LAST ||
                    out_layout == MNERF_CONV_OUT_PAIR_MAJOR, MNERF_E_RANGE, "%s: out_layout=%d", who, out_layout);
  p.out_layout = out_layout;
  p.add_cl = add_channel_last;
  MNERF_REQUIRE(out_layout == MNERF_CONV_OUT_NCHW || mnerf_aligned16(out), MNERF_E_ALIGN,
                "%s: a channel-last output must be 16-byte aligned", who);
  MNERF_REQUIRE(out_layout != MNERF_CONV_OUT_PAIR_MAJOR || n_img % 2 == 0, MNERF_E_RANGE,
                "%s: pair-major output needs an even number of images, got %d", who, n_img);
  MNERF_REQUIRE(!add_channel_last || (out_layout != MNERF_CONV_OUT_NCHW && mnerf_aligned16(add_channel_last)), MNERF_E_RANGE,
                "%s: add_channel_last needs a channel-last output layout and a 16-byte aligned tile", who);
  const long long n_pix = (long long)n_img * p.h_out * p.w_out;
  // two pixel tiles per wave share every weight fragment read; one tile per wave when that grid would leave CUs idle
  const int tpw = (n_pix + 255) / 256 >= 256 ? 2 : 1;
  const long long per_wg = 32 * tpw * CONV_NW;
  const dim3 grid((unsigned)((n_pix + per_wg - 1) / per_wg));
  const size_t lds = 2 * CONV_BUF_BYTES;
  hipStream_t st = (hipStream_t)stream;
  const bool cl = in_channels_last != 0;
  MNERF_REQUIRE(!cl || nmb == 4, MNERF_E_UNSUPPORTED, "%s: channel-last input is built for c_out = 128 only", who);
#define CONV_CASE(NMB, TPW, CLV)                                                                                      \
  if (nmb == NMB && tpw == TPW && cl == CLV) {                                                                        \
    static std::atomic<unsigned long long> attr{0};                                                                   \
    if (mnerf_once_per_device(attr))                                                                                  \
      (void)hipFuncSetAttribute((const void*)conv_kernel<NMB, TPW, CLV>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                (int)lds);                                                                            \
    hipLaunchKernelGGL((conv_kernel<NMB, TPW, CLV>), grid, dim3(CONV_NW * 64), lds, st, p);                           \
  }
  CONV_CASE(2, 2, false) CONV_CASE(2, 1, false) CONV_CASE(3, 2, false) CONV_CASE(3, 1, false)
  CONV_CASE(4, 2, false) CONV_CASE(4, 1, false) CONV_CASE(4, 2, true) CONV_CASE(4, 1, true)
#undef CONV_CASE
  return mnerf_check_launch(who);
}

// ============================================================================ the 7x7 stride-2 stem (3 -> 64 channels)
// backbone.py:45, 101: conv1 = Conv2d(3, 64, 7, stride 2, padding 3, bias=False).  K = 49 taps x 3 channels = 147,
// padded to ten K16-steps with zero weight columns; K index k = 3 tap + c.  The whole weight matrix (40 KiB of
// fragments) sits in LDS for the lifetime of the workgroup; an LDS table gives every k its input offset and tap
// displacement.  Lanes run along the output row (input pixels 8 bytes apart: stride 2).
#define STEM_STEPS 10
#define STEM_K 147
#define STEM_W_BYTES (STEM_STEPS * 2 * H16_UNIT_BYTES)  // 40 KiB

struct StemParams {
  const float* in;       // [n_img, 3, h_in, w_in]
  const float* wstream;  // STEM_STEPS x 2 units
  float* out;            // [n_img, 64, h_out, w_out]
  const float* in_absmax;
  int n_img, h_in, w_in, h_out, w_out, ew;
};

__global__ __launch_bounds__(CONV_NW * 64, 2) void conv_stem_kernel(StemParams P) {
  extern __shared__ __attribute__((aligned(16))) float stem_smem[];
  const unsigned buf0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)stem_smem;
  int* tab_off = reinterpret_cast<int*>(stem_smem) + STEM_W_BYTES / 4;  // [160] input offset of k from the centre pixel
  int* tab_dy = tab_off + 16 * STEM_STEPS;                              // [160] dy (-3..3), 100 for the zero padding columns
  int* tab_dx = tab_dy + 16 * STEM_STEPS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hl = lane >> 5;
  {
    const float* src = P.wstream + lane * 4;
    glds_segment(src, buf0, STEM_W_BYTES / 1024, wave, CONV_NW);
  }
  const int hw_in = P.h_in * P.w_in;
  for (int k = tid; k < 16 * STEM_STEPS; k += CONV_NW * 64) {
    const int tap = k / 3, c = k - 3 * tap, dy = tap / 7 - 3, dx = tap % 7 - 3;
    const bool real = k < STEM_K;
    tab_off[k] = real ? c * hw_in + dy * P.w_in + dx : 0;
    tab_dy[k] = real ? dy : 100;
    tab_dx[k] = real ? dx : 100;
  }
  const int hw_out = P.h_out * P.w_out;
  const long long n_pix = (long long)P.n_img * hw_out;
  constexpr int TPW = 2;
  int cy[TPW], cx[TPW];
  long long ibase[TPW], obase[TPW];
  bool live[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const long long p = ((long long)blockIdx.x * CONV_NW + wave) * (32 * TPW) + t * 32 + n;
    live[t] = p < n_pix;
    const long long pc = live[t] ? p : n_pix - 1;
    const int img = (int)(pc / hw_out), rem = (int)(pc - (long long)img * hw_out);
    const int oy = rem / P.w_out, ox = rem - oy * P.w_out;
    cy[t] = 2 * oy;
    cx[t] = 2 * ox;
    ibase[t] = (long long)img * 3 * hw_in + (long long)cy[t] * P.w_in + cx[t];
    obase[t] = (long long)img * 64 * hw_out + rem;
  }
  const int eg = gain_exp(mnerf_absmax_read(P.in_absmax));
  const float mult = pow2i(eg);
  f32x16 acc[TPW][2];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t][0] = acc[t][1] = (f32x16)(0.0f);
  segment_wait();
  __syncthreads();

  // A tile whose 32 pixels keep all 49 taps inside the image (the usual case) loads without any per-tap test: the zero
  // padding COLUMNS (k >= 147) have zero weights, so what they load (the centre pixel) does not matter.  Values of
  // step s+1 are requested before the 12 matrix instructions of step s.
  bool inside[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
    inside[t] = __all(live[t] && cy[t] >= 3 && cy[t] + 3 < P.h_in && cx[t] >= 3 && cx[t] + 3 < P.w_in);
  float vn[TPW][8];
  auto load_step = [&](int s) {
    const int k0 = 16 * s + 8 * hl;
    int off[8];
    {
      const int4 o0 = *reinterpret_cast<const int4*>(tab_off + k0), o1 = *reinterpret_cast<const int4*>(tab_off + k0 + 4);
      off[0] = o0.x; off[1] = o0.y; off[2] = o0.z; off[3] = o0.w; off[4] = o1.x; off[5] = o1.y; off[6] = o1.z; off[7] = o1.w;
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (inside[t]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) vn[t][j] = P.in[ibase[t] + off[j]];
      } else {
        int dy[8], dx[8];
        const int4 y0 = *reinterpret_cast<const int4*>(tab_dy + k0), y1 = *reinterpret_cast<const int4*>(tab_dy + k0 + 4);
        const int4 x0 = *reinterpret_cast<const int4*>(tab_dx + k0), x1 = *reinterpret_cast<const int4*>(tab_dx + k0 + 4);
        dy[0] = y0.x; dy[1] = y0.y; dy[2] = y0.z; dy[3] = y0.w; dy[4] = y1.x; dy[5] = y1.y; dy[6] = y1.z; dy[7] = y1.w;
        dx[0] = x0.x; dx[1] = x0.y; dx[2] = x0.z; dx[3] = x0.w; dx[4] = x1.x; dx[5] = x1.y; dx[6] = x1.z; dx[7] = x1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int yy = cy[t] + dy[j], xx = cx[t] + dx[j];
          const bool ok = live[t] && yy >= 0 && yy < P.h_in && xx >= 0 && xx < P.w_in;   // also false for padding columns
          const float x = P.in[ok ? ibase[t] + off[j] : ibase[t]];
          vn[t][j] = ok ? x : 0.0f;
        }
      }
    }
  };
  load_step(0);
  for (int s = 0; s < STEM_STEPS; ++s) {
    PartsH b[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) b[t] = split8h(vn[t], mult);
    if (s + 1 < STEM_STEPS) load_step(s + 1);
    lds_u32x4_cptr a = (lds_u32x4_cptr)(size_t)(buf0 + (unsigned)(2 * s) * H16_UNIT_BYTES) + lane;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f16x8 ah = __builtin_bit_cast(f16x8, a[m * 128]), al = __builtin_bit_cast(f16x8, a[m * 128 + 64]);
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t][m] = mfma16h(ah, b[t].lo, acc[t][m]);
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t][m] = mfma16h(al, b[t].hi, acc[t][m]);
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t][m] = mfma16h(ah, b[t].hi, acc[t][m]);
    }
  }
  const float cm = pow2i(-(P.ew + eg));
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (!live[t]) continue;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        P.out[obase[t] + (long long)(32 * m + 8 * (r >> 2) + 4 * hl + (r & 3)) * hw_out] = acc[t][m][r] * cm;
  }
}

extern "C" int64_t mnerf_conv_stem_wstream_floats(void) { return STEM_W_BYTES / 4; }

extern "C" int mnerf_conv_stem(const float* wstream, int32_t ew, const float* in, const float* in_absmax, float* out,
                               int32_t n_img, int32_t h_in, int32_t w_in, void* stream) {
  const char* who = "mnerf_conv_stem";
  MNERF_REQUIRE(n_img >= 0 && h_in >= 1 && w_in >= 1, MNERF_E_RANGE, "%s: n_img=%d h_in=%d w_in=%d", who, n_img, h_in, w_in);
  MNERF_REQUIRE((long long)3 * h_in * w_in <= 0x7fffffffLL, MNERF_E_RANGE, "%s: image too large", who);
  if (n_img == 0) return MNERF_OK;
  MNERF_REQUIRE(wstream && in && in_absmax && out, MNERF_E_NULL, "%s: NULL buffer", who);
  MNERF_REQUIRE(mnerf_aligned16(wstream), MNERF_E_ALIGN, "%s: wstream must be 16-byte aligned", who);
  StemParams p;
  p.in = in;
  p.wstream = wstream;
  p.out = out;
  p.in_absmax = in_absmax;
  p.n_img = n_img;
  p.h_in = h_in;
  p.w_in = w_in;
  p.h_out = (h_in + 6 - 7) / 2 + 1;
  p.w_out = (w_in + 6 - 7) / 2 + 1;
  p.ew = ew;
  const long long n_pix = (long long)n_img * p.h_out * p.w_out;
  const size_t lds = STEM_W_BYTES + 3 * 16 * STEM_STEPS * sizeof(int);
  static std::atomic<unsigned long long> attr{0};
  if (mnerf_once_per_device(attr))
    (void)hipFuncSetAttribute((const void*)conv_stem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(CONV_NW * 64), lds, (hipStream_t)stream, p);
  return mnerf_check_launch(who);
}
