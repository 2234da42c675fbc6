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
//   data gradient   dX[ci][y][x] = sum_{tap, co} W[co][ci][tap] dY[co][(y + p - ky) / s][(x + p - kx) / s]   (and, with the other
//     gather, the training FORWARD in exact fp32: conv_gemm_kernel)
//     a wave owns 32 positions of one output row that share the x parity class (stride 2) x ALL input channels:
//     D[ci 32-block][position] += A[ci][co pair] B[co pair][position], A = the weights transposed to [tap][co][ci] (the caller
//     does that once per step: one permute), B = dY rows; per K-step (two output channels) one coalesced load of B, CIB of A,
//     CIB matrix instructions.  Taps whose (y + p - ky) is not a multiple of the stride or falls outside dY are skipped
//     (wave-uniform), columns outside dY contribute zeros.
//   weight gradient dW[co][ci][ky][kx] = sum_{n, yo, xo} dY[co][yo][xo] X[ci][s yo + ky - p][s xo + kx - p]
//     a wave owns (a chunk of dY rows, one 32-block of output channels, one 32-block of input channels) x ALL taps: D[co][ci]
//     per tap (nine accumulators at 3 x 3); the K dimension is positions, eight per step: lane (channel, k-half) reads FOUR
//     consecutive positions of its channel's dY row with one 16-byte load (the order of K inside a step is free as long as both
//     operands agree) and, per filter row, ONE window of its X row that covers all kx shifts; 4 x taps matrix instructions
//     consume them.  Partial sums per chunk go to a workspace, a second kernel adds the chunks in a fixed order
//     (bit-reproducible) and writes torch's [co][ci][ky][kx].
#include "common.hpp"
#include "split_f16.hpp"

typedef float cb_f16 __attribute__((ext_vector_type(16)));

struct ConvBwdGeom {
  int n, c_in, c_out, h, w, ho, wo, k, s, pad;
};

__device__ __forceinline__ cb_f16 cb_mfma(float a, float b, cb_f16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ============================================================================ data gradient / fp32 forward
// One kernel, two gathers.  It computes  dst[cm][row][col] = sum_{tap, ck} A[tap][ck][cm] src[ck][row'][col']  with
//   FWD = false (data gradient): cm = c_in, ck = c_out, src = dY, dst = dX; (row', col') = ((y + p - ky) / s, (x + p - kx) / s) where
//         that is an integer inside dY - a wave's positions share the x parity class, so the taps that contribute are wave-uniform;
//   FWD = true  (the training forward, exact fp32 like the gradients: no weight stream to re-pack after every optimizer step, no
//         operand gains): cm = c_out, ck = c_in (any count: the stem has 3), src = X, dst = Y; (row', col') = (s y + ky - p, s x + kx - p).
// NB blocks of 32 positions per wave: the weights (A) are read from L2 by every wave - 147 KB per 3x3 64 -> 64 filter - and one
// block per wave made that the bound (1.1 GB of weight reads per launch at the backbone's first stage: 64 TFLOP/s); two blocks
// halve it and bring the loads per matrix instruction from 1.5 to 1.
struct ConvGemm {
  int n, cm, ck, hs, ws, hd, wd, k, s, pad;  // src [n][ck][hs][ws], dst [n][cm][hd][wd]
};
template <int CIB, int NB, bool FWD, bool RAGGED>  // RAGGED: the source channel count is not a multiple of 8 (the stem's 3)
__global__ __launch_bounds__(256) void conv_gemm_kernel(const float* __restrict__ src, const float* __restrict__ wt, const float* __restrict__ bias,
                                                        float* __restrict__ dst, ConvGemm G, int nseg, long long n_waves) {
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (gw >= n_waves) return;
  const int nn = lane & 31, kk = lane >> 5;
  // wave -> (image, dst row y, x parity class px [data gradient], group of NB segments of 32 positions, group of CIB channel blocks)
  const int classes = FWD ? 1 : G.s;
  long long t = gw;
  const int cigs = G.cm / (32 * CIB);
  const int ci0 = (int)(t % cigs) * CIB * 32;
  t /= cigs;
  const int seg = (int)(t % nseg);
  t /= nseg;
  const int px = (int)(t % classes);
  t /= classes;
  const int y = (int)(t % G.hd);
  const int img = (int)(t / G.hd);
  cb_f16 acc[NB][CIB];
#pragma unroll
  for (int q = 0; q < NB; ++q)
#pragma unroll
    for (int c = 0; c < CIB; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[q][c][i] = 0.0f;
  const size_t plane_s = (size_t)G.hs * G.ws;
  const int ck8 = (G.ck + 7) & ~7;  // K runs over the source channels in trips of eight (two per matrix instruction)
  for (int ky = 0; ky < G.k; ++ky) {
    int ys;
    if (FWD) {
      ys = G.s * y + ky - G.pad;
      if (ys < 0 || ys >= G.hs) continue;  // (wave-uniform)
    } else {
      const int ty = y + G.pad - ky;
      if (ty < 0 || ty % G.s != 0) continue;
      ys = ty / G.s;
      if (ys >= G.hs) continue;
    }
    for (int kx = 0; kx < G.k; ++kx) {
      int shift = 0;
      if (!FWD) {
        const int tx0 = px + G.pad - kx;
        if (((tx0 % G.s) + G.s) % G.s != 0) continue;  // (wave-uniform: the parity class fixes which kx contribute)
        shift = tx0 >= 0 ? tx0 / G.s : -((-tx0) / G.s);  // source column = index in the class + shift
      }
      bool ok[NB];
      const float* brow[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int xq = (seg * NB + q) * 32 + nn;
        const int xs = FWD ? G.s * xq + kx - G.pad : xq + shift;
        ok[q] = xs >= 0 && xs < G.ws;
        brow[q] = src + ((size_t)img * G.ck + kk) * plane_s + (size_t)ys * G.ws + (ok[q] ? xs : 0);
      }
      const float* arow = wt + ((size_t)(ky * G.k + kx) * G.ck + kk) * G.cm + ci0 + nn;
      // four K-steps (eight source channels) per trip, two trips per loop turn, the next trip's loads issued in front of the
      // current trip's matrix instructions; channels beyond ck (the stem: 3) read channel 0 and count as zeros
      float b0[4][NB], a0[4][CIB], b1[4][NB], a1[4][CIB];
      auto fetch = [&](int c0, float (&b)[4][NB], float (&a)[4][CIB]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ch = c0 + 2 * u;  // (+ kk: in the row pointers)
          if constexpr (RAGGED) {
            const bool chok = ch + kk < G.ck;
            const int chs = chok ? ch : -kk;  // -> channel 0
#pragma unroll
            for (int q = 0; q < NB; ++q) {
              const float v = brow[q][(ptrdiff_t)chs * (ptrdiff_t)plane_s];
              b[u][q] = chok ? v : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < CIB; ++c) a[u][c] = arow[(ptrdiff_t)chs * G.cm + c * 32];
          } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) b[u][q] = brow[q][(size_t)ch * plane_s];
#pragma unroll
            for (int c = 0; c < CIB; ++c) a[u][c] = arow[(size_t)ch * G.cm + c * 32];
          }
        }
      };
      auto run = [&](const float (&b)[4][NB], const float (&a)[4][CIB]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < NB; ++q) {
            const float bu = ok[q] ? b[u][q] : 0.0f;
#pragma unroll
            for (int c = 0; c < CIB; ++c) acc[q][c] = cb_mfma(a[u][c], bu, acc[q][c]);
          }
      };
      fetch(0, b0, a0);
      for (int c0 = 0; c0 < ck8; c0 += 16) {
        const bool second = c0 + 8 < ck8;
        if (second) fetch(c0 + 8, b1, a1);
        run(b0, a0);
        if (c0 + 16 < ck8) fetch(c0 + 16, b0, a0);
        if (second) run(b1, a1);
      }
    }
  }
  // D[m = channel][n = position]: register i of lane (nn, kk) is row 8 (i / 4) + 4 kk + i % 4
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int xq = (seg * NB + q) * 32 + nn;
    const int x = FWD ? xq : G.s * xq + px;
    if (x < G.wd) {
      float* out = dst + ((size_t)img * G.cm * G.hd + y) * G.wd + x;
#pragma unroll
      for (int c = 0; c < CIB; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ch = ci0 + c * 32 + 8 * (i >> 2) + 4 * kk + (i & 3);
          out[(size_t)ch * G.hd * G.wd] = acc[q][c][i] + (FWD && bias ? bias[ch] : 0.0f);
        }
    }
  }
}

// ============================================================================ weight gradient
// NW consecutive floats of a row starting at column x0 (any alignment), zeros outside [0, limit)
template <int NW>
__device__ __forceinline__ void cb_window(const float* row, int x0, int limit, float (&w)[NW]) {
  if (x0 >= 0 && x0 + NW <= limit) {
#pragma unroll
    for (int i = 0; i + 4 <= NW; i += 4) __builtin_memcpy(&w[i], row + x0 + i, 16);  // (4-byte aligned 16-byte loads)
#pragma unroll
    for (int i = NW & ~3; i < NW; ++i) w[i] = row[x0 + i];
  } else {
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = x0 + i >= 0 && x0 + i < limit ? row[x0 + i] : 0.0f;
  }
}

// A wave owns (a chunk of dY rows, one 32-block of output channels, one 32-block of input channels) x ALL taps: the four dY
// positions a lane reads per step meet the K x K shifted windows of its X rows, which are read ONCE per row as a window of
// S * 3 + K consecutive floats and shifted in registers (v1 re-read X per kx and dY per ky: 0.29 KB of loads per matrix
// instruction; this form 0.15).  K = ksize, S = stride (compile time).
template <int K, int S>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                         ConvBwdGeom G, int chunks, int rows_per_chunk, long long n_waves) {
  constexpr int NW = S * 3 + K;  // columns S (x + j) + kx - pad for j = 0..3, kx = 0..K-1
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (gw >= n_waves) return;
  const int nn = lane & 31, kk = lane >> 5;
  const int cobs = G.c_out / 32, cibs = G.c_in / 32;
  long long t = gw;
  const int cob = (int)(t % cobs);
  t /= cobs;
  const int cib = (int)(t % cibs);
  const int chunk = (int)(t / cibs);
  cb_f16 acc[K][K];
#pragma unroll
  for (int p = 0; p < K; ++p)
#pragma unroll
    for (int q = 0; q < K; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[p][q][i] = 0.0f;
  const int row_end = min((chunk + 1) * rows_per_chunk, G.n * G.ho);
  for (int r = chunk * rows_per_chunk; r < row_end; ++r) {
    const int img = r / G.ho, yo = r - img * G.ho;
    const float* arow = dy + (((size_t)img * G.c_out + cob * 32 + nn) * G.ho + yo) * G.wo;
    const float* bplane = x + ((size_t)img * G.c_in + cib * 32 + nn) * G.h * G.w;
    bool rowok[K];
#pragma unroll
    for (int p = 0; p < K; ++p) {
      const int yi = S * yo + p - G.pad;
      rowok[p] = yi >= 0 && yi < G.h;  // (wave-uniform: the filter row looks at padding)
    }
    // the next step's operands are requested in front of the current step's matrix instructions (36 of them at 3 x 3: with the
    // accumulators of nine taps a SIMD holds ONE wave, nothing else hides the loads)
    float a[4], w[K][NW], an[4], wn[K][NW];
    auto fetch = [&](int x0, float (&a_)[4], float (&w_)[K][NW]) {
      const int xa = x0 + 4 * kk;
      cb_window<4>(arow, xa, G.wo, a_);
#pragma unroll
      for (int p = 0; p < K; ++p)
        if (rowok[p]) cb_window<NW>(bplane + (size_t)(S * yo + p - G.pad) * G.w, S * xa - G.pad, G.w, w_[p]);
    };
    fetch(0, a, w);
    for (int x0 = 0; x0 < G.wo; x0 += 8) {
      const bool more = x0 + 8 < G.wo;
      if (more) fetch(x0 + 8, an, wn);
#pragma unroll
      for (int j = 0; j < 4; ++j)  // (position outermost: consecutive matrix instructions never share an accumulator)
#pragma unroll
        for (int p = 0; p < K; ++p) {
          if (!rowok[p]) continue;
#pragma unroll
          for (int q = 0; q < K; ++q) acc[p][q] = cb_mfma(a[j], w[p][S * j + q], acc[p][q]);
        }
      if (more) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = an[j];
#pragma unroll
        for (int p = 0; p < K; ++p)
#pragma unroll
          for (int i = 0; i < NW; ++i) w[p][i] = wn[p][i];
      }
    }
  }
  // D[m = output channel][n = input channel] -> part[chunk][tap][co][ci]
#pragma unroll
  for (int p = 0; p < K; ++p)
#pragma unroll
    for (int q = 0; q < K; ++q) {
      float* dst = part + (((size_t)chunk * K * K + p * K + q) * G.c_out + cob * 32) * G.c_in + cib * 32 + nn;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = 8 * (i >> 2) + 4 * kk + (i & 3);
        dst[(size_t)m * G.c_in] = acc[p][q][i];
      }
    }
}

// The same decomposition on the 16-bit matrix pipe (round 6, second form): operands split into two fp16 terms with ONE power-of-two
// gain per tensor (from the absmax regions their producers fill, common.hpp), three products per MAC, fp32 accumulation - the
// arithmetic of the forward kernels (conv.hip).  A step is SIXTEEN positions (K = 16 per instruction: a lane holds eight
// consecutive positions of its channel's row = 32 contiguous bytes): 27 matrix instructions of 32 cycles do what 72 exact-f32
// instructions of 64 cycles did; the splits (~26 vector instructions per operand) are the price.  Partial sums are scaled back by
// 2^-(ex + ed) at the store.  2^-22 relative to the tensor's largest magnitude per element: gradients whose magnitude lies more
// than 22 binades below their tensor's maximum are lost (the forward has the same property).
template <int K, int S>
__global__ __launch_bounds__(256) void conv_wgrad16_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ x_absmax,
                                                           const float* __restrict__ dy_absmax, float* __restrict__ part, ConvBwdGeom G, int chunks,
                                                           int rows_per_chunk, long long n_waves) {
  constexpr int NW = S * 7 + K;  // columns S (x + j) + kx - pad for j = 0..7, kx = 0..K-1
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (gw >= n_waves) return;
  const int nn = lane & 31, kk = lane >> 5;
  const int cobs = G.c_out / 32, cibs = G.c_in / 32;
  long long t = gw;
  const int cob = (int)(t % cobs);
  t /= cobs;
  const int cib = (int)(t % cibs);
  const int chunk = (int)(t / cibs);
  const int ex = gain_exp(mnerf_absmax_read(x_absmax)), ed = gain_exp(mnerf_absmax_read(dy_absmax));
  const float gx = pow2i(ex), gd = pow2i(ed);
  f32x16 acc[K][K];
#pragma unroll
  for (int p = 0; p < K; ++p)
#pragma unroll
    for (int q = 0; q < K; ++q) acc[p][q] = (f32x16)(0.0f);
  const int row_end = min((chunk + 1) * rows_per_chunk, G.n * G.ho);
  for (int r = chunk * rows_per_chunk; r < row_end; ++r) {
    const int img = r / G.ho, yo = r - img * G.ho;
    const float* arow = dy + (((size_t)img * G.c_out + cob * 32 + nn) * G.ho + yo) * G.wo;
    const float* bplane = x + ((size_t)img * G.c_in + cib * 32 + nn) * G.h * G.w;
    bool rowok[K];
#pragma unroll
    for (int p = 0; p < K; ++p) {
      const int yi = S * yo + p - G.pad;
      rowok[p] = yi >= 0 && yi < G.h;  // (wave-uniform)
    }
    float a[8], w[K][NW], an[8], wn[K][NW];
    auto fetch = [&](int x0, float (&a_)[8], float (&w_)[K][NW]) {
      const int xa = x0 + 8 * kk;
      cb_window<8>(arow, xa, G.wo, a_);
#pragma unroll
      for (int p = 0; p < K; ++p)
        if (rowok[p]) cb_window<NW>(bplane + (size_t)(S * yo + p - G.pad) * G.w, S * xa - G.pad, G.w, w_[p]);
    };
    fetch(0, a, w);
    for (int x0 = 0; x0 < G.wo; x0 += 16) {
      const bool more = x0 + 16 < G.wo;
      if (more) fetch(x0 + 16, an, wn);
      const PartsH A = split8h(a, gd);
#pragma unroll
      for (int p = 0; p < K; ++p) {
        if (!rowok[p]) continue;
        if constexpr (K == 3 && S == 1) {
          // the ten values of the window are split ONCE (five fp16 pairs hi | lo); kx = 0 / 2 take pairs 0-3 / 1-4 as they are,
          // kx = 1 the pairs re-cut one element further (v_alignbit_b32): 38 vector instructions per row instead of 78
          unsigned H[5], L[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            const f32x2 ab = {w[p][2 * i] * gx, w[p][2 * i + 1] * gx};
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, f16x2));
            const f32x2 rr = {resid_lo(w[p][2 * i], gx, h), resid_hi(w[p][2 * i + 1], gx, h)};
            H[i] = h;
            L[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, f16x2));
          }
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            u32x4 bh, bl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (q == 1) {
                bh[i] = __builtin_amdgcn_alignbit(H[i + 1], H[i], 16);
                bl[i] = __builtin_amdgcn_alignbit(L[i + 1], L[i], 16);
              } else {
                bh[i] = H[i + q / 2];
                bl[i] = L[i + q / 2];
              }
            }
            const f16x8 Bh = __builtin_bit_cast(f16x8, bh), Bl = __builtin_bit_cast(f16x8, bl);
            acc[p][q] = mfma16h(A.lo, Bh, acc[p][q]);
            acc[p][q] = mfma16h(A.hi, Bl, acc[p][q]);
            acc[p][q] = mfma16h(A.hi, Bh, acc[p][q]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < K; ++q) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = w[p][S * j + q];
            const PartsH B = split8h(v8, gx);
            acc[p][q] = mfma16h(A.lo, B.hi, acc[p][q]);
            acc[p][q] = mfma16h(A.hi, B.lo, acc[p][q]);
            acc[p][q] = mfma16h(A.hi, B.hi, acc[p][q]);
          }
        }
      }
      if (more) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = an[j];
#pragma unroll
        for (int p = 0; p < K; ++p)
#pragma unroll
          for (int i = 0; i < NW; ++i) w[p][i] = wn[p][i];
      }
    }
  }
  const float unscale = pow2i(-(ex + ed));
#pragma unroll
  for (int p = 0; p < K; ++p)
#pragma unroll
    for (int q = 0; q < K; ++q) {
      float* dst = part + (((size_t)chunk * K * K + p * K + q) * G.c_out + cob * 32) * G.c_in + cib * 32 + nn;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = 8 * (i >> 2) + 4 * kk + (i & 3);
        dst[(size_t)m * G.c_in] = acc[p][q][i] * unscale;
      }
    }
}

// dW[co][ci][ky][kx] = sum over the chunks, in chunk order
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int chunks, int taps,
                                                                int c_out, int c_in) {
  const int per_chunk = taps * c_out * c_in;
  const int e = blockIdx.x * 256 + threadIdx.x;  // index into [tap][co][ci]
  if (e >= per_chunk) return;
  // four running sums (chunk index mod 4), added at the end: four independent load chains instead of one (the order is fixed)
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {
    s0 += part[(size_t)c * per_chunk + e], s1 += part[(size_t)(c + 1) * per_chunk + e];
    s2 += part[(size_t)(c + 2) * per_chunk + e], s3 += part[(size_t)(c + 3) * per_chunk + e];
  }
  for (; c < chunks; ++c) s0 += part[(size_t)c * per_chunk + e];
  const float s = (s0 + s1) + (s2 + s3);
  const int ci = e % c_in, co = (e / c_in) % c_out, tap = e / (c_in * c_out);
  dw[((size_t)co * c_in + ci) * taps + tap] = s;
}

// ============================================================================ the stem's weight gradient
// Conv2d(3, 64, 7, stride 2, padding 3) (backbone.py:45): three input channels would leave 29 of the 32 columns of D[co][ci] empty
// and 49 taps are 49 accumulators; here a COLUMN of D is (input channel, kx) - 21 of 32 used - and a wave keeps one accumulator
// per filter ROW: D_ky[co][(c, kx)] += dY[co][pos] X[c][2 yo + ky - 3][2 xo + kx - 3], the X values gathered per lane.
__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                              int n_img, int h, int w, int ho, int wo, int chunks, int rows_per_chunk,
                                                              long long n_waves) {
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (gw >= n_waves) return;
  const int nn = lane & 31, kk = lane >> 5;
  const int cob = (int)(gw & 1), chunk = (int)(gw >> 1);
  const int c = nn / 7, kx = nn - 7 * c;
  const bool col_ok = nn < 21;
  cb_f16 acc[7];
#pragma unroll
  for (int p = 0; p < 7; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.0f;
  const int row_end = min((chunk + 1) * rows_per_chunk, n_img * ho);
  for (int r = chunk * rows_per_chunk; r < row_end; ++r) {
    const int img = r / ho, yo = r - img * ho;
    const float* arow = dy + (((size_t)img * 64 + cob * 32 + nn) * ho + yo) * wo;
    const float* bplane = x + ((size_t)img * 3 + (col_ok ? c : 0)) * h * w;
    for (int x0 = 0; x0 < wo; x0 += 8) {
      const int xa = x0 + 4 * kk;
      float a[4];
      cb_window<4>(arow, xa, wo, a);
#pragma unroll
      for (int p = 0; p < 7; ++p) {
        const int yi = 2 * yo + p - 3;
        if (yi < 0 || yi >= h) continue;  // (wave-uniform)
        const float* brow = bplane + (size_t)yi * w;
        float b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int xi = 2 * (xa + j) + kx - 3;
          const bool ok = col_ok && xi >= 0 && xi < w && xa + j < wo;
          const float v = brow[ok ? xi : 0];
          b[j] = ok ? v : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p] = cb_mfma(a[j], b[j], acc[p]);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 7; ++p) {
    float* dst = part + (((size_t)chunk * 7 + p) * 64 + cob * 32) * 32 + nn;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[(size_t)(8 * (i >> 2) + 4 * kk + (i & 3)) * 32] = acc[p][i];
  }
}
// dW[co][c][ky][kx] = sum over the chunks of part[chunk][ky][co][7 c + kx]
__global__ __launch_bounds__(256) void conv_stem_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int chunks) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // index into dW: ((co * 3 + c) * 7 + ky) * 7 + kx
  if (e >= 64 * 147) return;
  const int kx = e % 7, ky = (e / 7) % 7, c = (e / 49) % 3, co = e / 147;
  const float* src = part + ((size_t)ky * 64 + co) * 32 + 7 * c + kx;
  const size_t stride = (size_t)7 * 64 * 32;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int ch = 0;
  for (; ch + 4 <= chunks; ch += 4) {
    s0 += src[(size_t)ch * stride], s1 += src[(size_t)(ch + 1) * stride];
    s2 += src[(size_t)(ch + 2) * stride], s3 += src[(size_t)(ch + 3) * stride];
  }
  for (; ch < chunks; ++ch) s0 += src[(size_t)ch * stride];
  dw[e] = (s0 + s1) + (s2 + s3);
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

// launch of conv_gemm_kernel: channel blocks per wave - all of them (the source loads are shared) unless that leaves the chip short of
// waves (the 64 x 80 maps of the last stage: 576 waves with four blocks each); two position blocks per wave where they fit
template <bool FWD>
static void cb_gemm_launch(const float* src, const float* wt, const float* bias, float* dst, const ConvGemm& G, hipStream_t st) {
  const int classes = FWD ? 1 : G.s;
  const int per_class = FWD ? G.wd : (G.wd + G.s - 1) / G.s;
  const int segs = (per_class + 31) / 32;
  const long long rows = (long long)G.n * G.hd * classes;
  int cib = G.cm / 32;
  while (cib > 1 && cib % 2 == 0 && rows * segs * (G.cm / 32 / cib) < 2048) cib /= 2;
  if (cib == 3 && rows * segs < 2048) cib = 1;
  const int nb = cib <= 2 && rows * ((segs + 1) / 2) * (G.cm / 32 / cib) >= 2048 ? 2 : 1;
  const int nseg = (segs + nb - 1) / nb;
  const long long waves = rows * nseg * (G.cm / 32 / cib);
  const dim3 grid((unsigned)((waves + 3) / 4));
#define CB_DG(C_, N_)                                                                                                           \
  do {                                                                                                                          \
    if (G.ck % 8 == 0)                                                                                                          \
      hipLaunchKernelGGL((conv_gemm_kernel<C_, N_, FWD, false>), grid, dim3(256), 0, st, src, wt, bias, dst, G, nseg, waves);   \
    else                                                                                                                        \
      hipLaunchKernelGGL((conv_gemm_kernel<C_, N_, FWD, true>), grid, dim3(256), 0, st, src, wt, bias, dst, G, nseg, waves);    \
  } while (0)
  if (cib == 1 && nb == 2) CB_DG(1, 2);
  else if (cib == 1) CB_DG(1, 1);
  else if (cib == 2 && nb == 2) CB_DG(2, 2);
  else if (cib == 2) CB_DG(2, 1);
  else if (cib == 3) CB_DG(3, 1);
  else CB_DG(4, 1);
#undef CB_DG
}

extern "C" int mnerf_conv2d_backward_data(const float* dy, const float* w_tap_major, float* dx, int32_t n_img, int32_t c_in, int32_t c_out,
                                          int32_t h_in, int32_t w_in, int32_t ksize, int32_t stride, void* stream) {
  const char* who = "mnerf_conv2d_backward_data";
  ConvBwdGeom B;
  if (const int rc = cb_geom(who, B, n_img, c_in, c_out, h_in, w_in, ksize, stride)) return rc;
  if (n_img == 0) return MNERF_OK;
  MNERF_REQUIRE(dy && w_tap_major && dx, MNERF_E_NULL, "%s: NULL buffer", who);
  ConvGemm G;
  G.n = n_img, G.cm = c_in, G.ck = c_out, G.hs = B.ho, G.ws = B.wo, G.hd = h_in, G.wd = w_in, G.k = ksize, G.s = stride, G.pad = B.pad;
  cb_gemm_launch<false>(dy, w_tap_major, nullptr, dx, G, (hipStream_t)stream);
  return mnerf_check_launch(who);
}

extern "C" int mnerf_conv2d_forward_f32(const float* x, const float* w_tap_major, const float* bias, float* y, int32_t n_img, int32_t c_in,
                                        int32_t c_out, int32_t h_in, int32_t w_in, int32_t ksize, int32_t stride, void* stream) {
  const char* who = "mnerf_conv2d_forward_f32";
  MNERF_REQUIRE(n_img >= 0 && h_in >= 1 && w_in >= 1, MNERF_E_RANGE, "%s: n=%d h=%d w=%d", who, n_img, h_in, w_in);
  MNERF_REQUIRE(c_in >= 1 && c_in <= 128 && c_out >= 32 && c_out <= 128 && c_out % 32 == 0, MNERF_E_UNSUPPORTED,
                "%s: channels %d -> %d (any input count up to 128, output multiples of 32 up to 128 are built)", who, c_in, c_out);
  MNERF_REQUIRE((ksize == 1 || ksize == 3 || ksize == 7) && (stride == 1 || stride == 2), MNERF_E_UNSUPPORTED, "%s: ksize=%d stride=%d", who,
                ksize, stride);
  if (n_img == 0) return MNERF_OK;
  MNERF_REQUIRE(x && w_tap_major && y, MNERF_E_NULL, "%s: NULL buffer", who);
  ConvGemm G;
  G.n = n_img, G.cm = c_out, G.ck = c_in, G.hs = h_in, G.ws = w_in, G.k = ksize, G.s = stride, G.pad = ksize / 2;
  G.hd = (h_in + 2 * G.pad - ksize) / stride + 1;
  G.wd = (w_in + 2 * G.pad - ksize) / stride + 1;
  cb_gemm_launch<true>(x, w_tap_major, bias, y, G, (hipStream_t)stream);
  return mnerf_check_launch(who);
}

// chunks of dY rows: enough waves for the chip, partial sums of at most ~32 MiB
static void cb_chunks(const ConvBwdGeom& G, int& chunks, int& rpc) {
  const long long rows = (long long)G.n * G.ho;
  const long long per_chunk_bytes = (long long)G.k * G.k * G.c_out * G.c_in * 4;
  long long want = 4096 / ((long long)(G.c_out / 32) * (G.c_in / 32));
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

static int cb_wgrad_impl(const char* who, const float* x, const float* dy, const float* x_absmax, const float* dy_absmax, float* dw,
                         void* workspace, size_t workspace_bytes, int32_t n_img, int32_t c_in, int32_t c_out, int32_t h_in, int32_t w_in,
                         int32_t ksize, int32_t stride, void* stream);

extern "C" int mnerf_conv2d_backward_weight(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int32_t n_img,
                                            int32_t c_in, int32_t c_out, int32_t h_in, int32_t w_in, int32_t ksize, int32_t stride,
                                            void* stream) {
  return cb_wgrad_impl("mnerf_conv2d_backward_weight", x, dy, nullptr, nullptr, dw, workspace, workspace_bytes, n_img, c_in, c_out, h_in, w_in,
                       ksize, stride, stream);
}

extern "C" int mnerf_conv2d_backward_weight_f16x3(const float* x, const float* dy, const float* x_absmax, const float* dy_absmax, float* dw,
                                                  void* workspace, size_t workspace_bytes, int32_t n_img, int32_t c_in, int32_t c_out,
                                                  int32_t h_in, int32_t w_in, int32_t ksize, int32_t stride, void* stream) {
  const char* who = "mnerf_conv2d_backward_weight_f16x3";
  MNERF_REQUIRE(x_absmax && dy_absmax, MNERF_E_NULL, "%s: absmax region is NULL", who);
  return cb_wgrad_impl(who, x, dy, x_absmax, dy_absmax, dw, workspace, workspace_bytes, n_img, c_in, c_out, h_in, w_in, ksize, stride, stream);
}

static int cb_wgrad_impl(const char* who, const float* x, const float* dy, const float* x_absmax, const float* dy_absmax, float* dw,
                         void* workspace, size_t workspace_bytes, int32_t n_img, int32_t c_in, int32_t c_out, int32_t h_in, int32_t w_in,
                         int32_t ksize, int32_t stride, void* stream) {
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
  const long long waves = (long long)chunks * (G.c_in / 32) * (G.c_out / 32);
  const dim3 grid((unsigned)((waves + 3) / 4));
#define CB_WG(K_, S_)                                                                                                                       \
  do {                                                                                                                                      \
    if (x_absmax)                                                                                                                           \
      hipLaunchKernelGGL((conv_wgrad16_kernel<K_, S_>), grid, dim3(256), 0, st, x, dy, x_absmax, dy_absmax, part, G, chunks, rpc, waves);   \
    else                                                                                                                                    \
      hipLaunchKernelGGL((conv_wgrad_kernel<K_, S_>), grid, dim3(256), 0, st, x, dy, part, G, chunks, rpc, waves);                          \
  } while (0)
  if (ksize == 3 && stride == 1) CB_WG(3, 1);
  else if (ksize == 3) CB_WG(3, 2);
  else if (stride == 1) CB_WG(1, 1);
  else CB_WG(1, 2);
#undef CB_WG
  const int per_chunk = G.k * G.k * G.c_out * G.c_in;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((per_chunk + 255) / 256)), dim3(256), 0, st, part, dw, chunks, G.k * G.k, G.c_out,
                     G.c_in);
  return mnerf_check_launch(who);
}

static void cb_stem_chunks(int n_img, int ho, int& chunks, int& rpc) {
  const long long rows = (long long)n_img * ho;
  long long want = rows < 384 ? rows : 384;  // (two waves per chunk; the reduction reads one value per chunk and output)
  if (want < 1) want = 1;
  rpc = (int)((rows + want - 1) / want);
  chunks = (int)((rows + rpc - 1) / rpc);
}

extern "C" size_t mnerf_conv_stem_backward_weight_workspace_bytes(int32_t n_img, int32_t h_in, int32_t w_in) {
  if (n_img <= 0 || h_in < 1 || w_in < 1) return 0;
  int chunks, rpc;
  cb_stem_chunks(n_img, (h_in - 1) / 2 + 1, chunks, rpc);
  return (size_t)chunks * 7 * 64 * 32 * sizeof(float);
}

extern "C" int mnerf_conv_stem_backward_weight(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int32_t n_img,
                                               int32_t h_in, int32_t w_in, void* stream) {
  const char* who = "mnerf_conv_stem_backward_weight";
  MNERF_REQUIRE(n_img >= 0 && h_in >= 1 && w_in >= 1, MNERF_E_RANGE, "%s: n=%d h=%d w=%d", who, n_img, h_in, w_in);
  MNERF_REQUIRE(dw, MNERF_E_NULL, "%s: dw is NULL", who);
  hipStream_t st = (hipStream_t)stream;
  if (n_img == 0) {
    const hipError_t e = hipMemsetAsync(dw, 0, (size_t)64 * 147 * sizeof(float), st);
    return e == hipSuccess ? MNERF_OK : (int)e;
  }
  MNERF_REQUIRE(x && dy && workspace, MNERF_E_NULL, "%s: NULL buffer", who);
  const int ho = (h_in - 1) / 2 + 1, wo = (w_in - 1) / 2 + 1;
  int chunks, rpc;
  cb_stem_chunks(n_img, ho, chunks, rpc);
  const size_t need = (size_t)chunks * 7 * 64 * 32 * sizeof(float);
  MNERF_REQUIRE(workspace_bytes >= need, MNERF_E_RANGE, "%s: workspace %zu bytes < %zu", who, workspace_bytes, need);
  float* part = reinterpret_cast<float*>(workspace);
  const long long waves = 2ll * chunks;
  hipLaunchKernelGGL(conv_stem_wgrad_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, dy, part, n_img, h_in, w_in, ho, wo, chunks,
                     rpc, waves);
  hipLaunchKernelGGL(conv_stem_wgrad_reduce_kernel, dim3((64 * 147 + 255) / 256), dim3(256), 0, st, part, dw, chunks);
  return mnerf_check_launch(who);
}
