// q | k | v projections of a GMFlow transformer layer as ONE kernel (gfx950).
//
// Replaces (paths relative to /root/reference/models/gmflow):
//   transformer.py:147-151  query = q_proj(source); key = k_proj(target); value = v_proj(target)
//   transformer.py:317-335  concat1 = cat(feature1, feature0): the "target" of the batched [f0; f1] sequences is the
//                           same tensor with its two halves swapped - read here through an index (kv_swap), never built
// Three library GEMM launches (14.7 us each at 30,720 tokens: launch- and latency-bound) and, for the cross-attention
// layers, a torch.cat per block before.  Same machinery as the encoder block kernel (encoder_block.hip): transposed
// chain Y^T[out, token] = W . X^T on v_mfma_f32_32x32x16_f16 with fp32-grade split-fp16 operands and one power-of-two
// gain per token; a wave owns 32 tokens and splits a token's 128 features ONCE for the projections that share them;
// the three weight matrices stream as six 32 KiB segments through a 2 x 32 KiB LDS double buffer.  Two waves per
// workgroup: 480 workgroups at 3 views, two per CU.
#include "wa_common.hpp"

#define QKV_NW 2
#define QKV_SEG_FLOATS (32 * 256)  // 4 K16-steps x 4 blocks x [hi | lo] x 1 KiB
#define QKV_C 128

struct QkvParams {
  const float* xq;
  const float* xkv;
  float* out[3];
  const float* wstream;
  int n_tokens, seq_len, n_seq, kv_swap;
  int ew[3];
};

// the lane's half of every K16-step of token row `row`: features 16 t + 8 hl + j, split with the token's own gain
__device__ __forceinline__ int qkv_load_split(const float* __restrict__ row, PartsH (&xp)[8]) {
  float x[64];
  float amax = 0.0f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(row + 16 * t), b = *reinterpret_cast<const float4*>(row + 16 * t + 4);
    x[8 * t + 0] = a.x; x[8 * t + 1] = a.y; x[8 * t + 2] = a.z; x[8 * t + 3] = a.w;
    x[8 * t + 4] = b.x; x[8 * t + 5] = b.y; x[8 * t + 6] = b.z; x[8 * t + 7] = b.w;
  }
#pragma unroll
  for (int i = 0; i < 64; ++i) amax = fmaxf(amax, fabsf(x[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
  const int eg = gain_exp(amax);
  const float mult = pow2i(eg);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float v8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = x[8 * t + j];
    xp[t] = split8h(v8, mult);
  }
  return eg;
}

__global__ __launch_bounds__(QKV_NW * 64, 2) void qkv_kernel(QkvParams P) {
  extern __shared__ __attribute__((aligned(16))) float qkv_smem[];
  const unsigned buf0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)qkv_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hl = lane >> 5;

  auto stage = [&](int seg) {
    const float* src = P.wstream + (size_t)seg * QKV_SEG_FLOATS + lane * 4;
    const unsigned dst = buf0 + (unsigned)(seg & 1) * (QKV_SEG_FLOATS * 4u);
    glds_segment(src, dst, QKV_SEG_FLOATS / 256, wave, QKV_NW);
  };
  stage(0);

  const int tok_raw = (blockIdx.x * QKV_NW + wave) * 32 + n;
  const bool tok_ok = tok_raw < P.n_tokens;
  const int tok = tok_ok ? tok_raw : P.n_tokens - 1;
  // the key / value source of a token: the same position in the sequence half a batch away (cross attention between
  // the two members of every pair), or the token itself
  int tok_kv = tok;
  if (P.kv_swap) {
    const int b = tok / P.seq_len, i = tok - b * P.seq_len;
    int bs = b + (P.n_seq >> 1);
    if (bs >= P.n_seq) bs -= P.n_seq;
    tok_kv = bs * P.seq_len + i;
  }
  const bool same_rows = !P.kv_swap && P.xkv == P.xq;

  PartsH xp[8];
  int eg = qkv_load_split(P.xq + (size_t)tok * QKV_C + 8 * hl, xp);
  segment_wait();
  __syncthreads();

  int seg = 0;
#pragma unroll 1
  for (int proj = 0; proj < 3; ++proj) {
    if (proj == 1 && !same_rows) eg = qkv_load_split(P.xkv + (size_t)tok_kv * QKV_C + 8 * hl, xp);
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
#pragma unroll
    for (int half = 0; half < 2; ++half, ++seg) {
      if (seg + 1 < 6) stage(seg + 1);
      ksteps_presplit<4, 4>(acc, buf0 + (unsigned)(seg & 1) * (QKV_SEG_FLOATS * 4u), lane, xp + 4 * half);
      segment_wait();
      __syncthreads();
    }
    if (tok_ok) {
      const float cm = pow2i(-(P.ew[proj] + eg));
      float* dst = P.out[proj] + (size_t)tok * QKV_C + 4 * hl;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // q, v: accumulator row order (channel 32 m + 8 g + 4 hl + q); k: the rows of Wk are packed so that registers
          // 8 t .. 8 t + 7 hold channels 16 t + 8 hl + j (the attention's operand order, qkv_images_kernel below)
          const int c0 = proj == 1 ? 16 * (2 * m + (g >> 1)) + 4 * (g & 1) + 4 * hl : 32 * m + 8 * g;
          *reinterpret_cast<float4*>(dst + c0) =
              make_float4(acc[m][4 * g] * cm, acc[m][4 * g + 1] * cm, acc[m][4 * g + 2] * cm, acc[m][4 * g + 3] * cm);
        }
    }
  }
}

// ---------------------------------------------------------------- q | k | v with K and V written as attention operands
// The window attention (window_attention.hip) consumes K and V as split-fp16 A-operand images per 32-key tile of a
// window (wa_common.hpp).  A wave here owns exactly one such tile - the 32 tokens are gathered in window-local order
// through win_token() - and writes the images itself, so the k / v tensors and the operand pre-pass disappear:
//   K  transposed chain Y^T = Wk X^T with the rows of Wk packed so that accumulator registers 8t..8t+7 of lane
//      (key, half) ARE the eight channels of K16-step t of S^T = K Q^T: split8h straight out of the accumulator;
//   V  the same fragments of Wv used as the B operand and the token features as the A operand (Y = X Wv^T): the
//      accumulator then holds keys in registers and channels across lanes - the layout of the A operand of
//      O^T += V^T P^T - again split8h straight out of the accumulator;
//   one power-of-two gain per tile and matrix from the wave's maximum; q is stored as fp32 rows by token.
struct QkvImgParams {
  const float* xq;
  const float* xkv;
  float* q;
  u32x4* img;
  int* rec;
  const float* wstream;
  WinGeom G;
  int n_batch, n_tiles, total_tiles, kv_swap;
  int ew[3];
};

// acc[m] += X . W^T for block m: the token features as the A operand, the weight fragments as the B operand
template <int NS>
__device__ __forceinline__ void ksteps_presplit_swapped(f32x16 (&acc)[4], unsigned base_lds, int lane, const PartsH* x) {
  lds_u32x4_cptr a = (lds_u32x4_cptr)(size_t)base_lds + lane;
  u32x4 ch = a[0], cl = a[64];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = u * 4 + m;
      const int nx = (i + 1 < NS * 4) ? (i + 1) * 128 : i * 128;
      const u32x4 nh = a[nx], nl = a[nx + 64];
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 wh = __builtin_bit_cast(f16x8, ch), wl = __builtin_bit_cast(f16x8, cl);
      acc[m] = mfma16h(x[u].lo, wh, acc[m]);  // the same three products in the same order as ksteps_presplit
      acc[m] = mfma16h(x[u].hi, wl, acc[m]);
      acc[m] = mfma16h(x[u].hi, wh, acc[m]);
      __builtin_amdgcn_sched_barrier(0);
      ch = nh;
      cl = nl;
    }
  }
}

__global__ __launch_bounds__(QKV_NW * 64, 2) void qkv_images_kernel(QkvImgParams P) {
  extern __shared__ __attribute__((aligned(16))) float qkv_smem[];
  const unsigned buf0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)qkv_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hl = lane >> 5;
  auto stage = [&](int seg) {
    const float* src = P.wstream + (size_t)seg * QKV_SEG_FLOATS + lane * 4;
    const unsigned dst = buf0 + (unsigned)(seg & 1) * (QKV_SEG_FLOATS * 4u);
    glds_segment(src, dst, QKV_SEG_FLOATS / 256, wave, QKV_NW);
  };
  stage(0);

  const WinGeom& G = P.G;
  const int t_raw = blockIdx.x * QKV_NW + wave;
  const bool tile_ok = t_raw < P.total_tiles;             // a surplus wave repeats the last tile and stores nothing
  const int tile = tile_ok ? t_raw : P.total_tiles - 1;
  const int gwin = tile / P.n_tiles, kt = tile - gwin * P.n_tiles;
  const int n_win = G.splits * G.splits;
  const int b = gwin / n_win, win = gwin - b * n_win;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const int li = kt * WA_KT + n;
  const bool key_ok = li < G.Lw;
  int region_unused;
  const int tok = win_token(G, wy, wx, key_ok ? li : G.Lw - 1, region_unused);
  int b_kv = b;
  if (P.kv_swap) {
    b_kv = b + (P.n_batch >> 1);
    if (b_kv >= P.n_batch) b_kv -= P.n_batch;
  }
  const size_t hw = (size_t)G.h * G.w;
  const bool same_rows = !P.kv_swap && P.xkv == P.xq;

  PartsH xp[8];
  int eg = qkv_load_split(P.xq + ((size_t)b * hw + tok) * QKV_C + 8 * hl, xp);
  segment_wait();
  __syncthreads();

  u32x4* img = P.img + (size_t)tile * (WA_IMG_BYTES / 16) + lane;
  int ek = 0, ev = 0;
  int seg = 0;
#pragma unroll 1
  for (int proj = 0; proj < 3; ++proj) {
    if (proj == 1 && !same_rows) eg = qkv_load_split(P.xkv + ((size_t)b_kv * hw + tok) * QKV_C + 8 * hl, xp);
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
#pragma unroll
    for (int half = 0; half < 2; ++half, ++seg) {
      if (seg + 1 < 6) stage(seg + 1);
      const unsigned cur = buf0 + (unsigned)(seg & 1) * (QKV_SEG_FLOATS * 4u);
      if (proj == 2) ksteps_presplit_swapped<4>(acc, cur, lane, xp + 4 * half);
      else ksteps_presplit<4, 4>(acc, cur, lane, xp + 4 * half);
      segment_wait();
      __syncthreads();
    }
    if (proj == 0) {
      if (tile_ok && key_ok) {
        const float cm = pow2i(-(P.ew[0] + eg));
        float* dst = P.q + ((size_t)b * hw + tok) * QKV_C + 4 * hl;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(dst + 32 * m + 8 * g) =
                make_float4(acc[m][4 * g] * cm, acc[m][4 * g + 1] * cm, acc[m][4 * g + 2] * cm, acc[m][4 * g + 3] * cm);
      }
      continue;
    }
    // K (proj 1): lane (key, half), registers 8t..8t+7 = channels 16t + 8 half + j;  V (proj 2): lane (channel 32m + n
    // of block m, half), registers 8t..8t+7 of block m = keys key(t, half, j).  In both cases the per-token (K) or
    // per-key-row (V) operand gain of the projection is undone first: the accumulator of lane (n, half) carries
    // 2^(ew + eg(n)) for K (its own token) - for V the rows are OTHER tokens' gains, see below.
    if (proj == 1) {
      const float cm = pow2i(-(P.ew[1] + eg));
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] *= cm;
    } else {
      // row i of the V accumulator is key i = (r & 3) + 8 (r >> 2) + 4 half, scaled by THAT key's operand gain:
      // fetch the gain exponent of key i from the lane that owns it (lane i of the lower half-wave)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * hl;
        const int eg_key = __shfl(eg, key, 64);
        const float cm = pow2i(-(P.ew[2] + eg_key));
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][r] *= cm;
      }
    }
    float tmax = 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, fabsf(acc[m][r]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, off, 64));
    const int et = gain_exp(tmax);
    const float mt = pow2i(et);
    if (proj == 1) {
      ek = et;
      if (tile_ok) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = acc[t >> 1][8 * (t & 1) + j];
          const PartsH kp = split8h(v8, mt);
          img[(2 * t) * 64] = __builtin_bit_cast(u32x4, kp.hi);
          img[(2 * t + 1) * 64] = __builtin_bit_cast(u32x4, kp.lo);
        }
      }
    } else {
      ev = et;
      if (tile_ok) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = acc[m][8 * t + j];
            const PartsH vp = split8h(v8, mt);
            img[(WA_IMG_VOFF / 64 + 2 * (4 * t + m)) * 64] = __builtin_bit_cast(u32x4, vp.hi);
            img[(WA_IMG_VOFF / 64 + 2 * (4 * t + m) + 1) * 64] = __builtin_bit_cast(u32x4, vp.lo);
          }
      }
    }
  }
  if (tile_ok) {
    int* rec = P.rec + WA_REC_INTS * (size_t)tile;
    wa_store_regions(G, wy, wx, kt, lane, rec);
    if (lane == 0) {
      rec[0] = ek;
      rec[1] = ev;
    }
  }
}

extern "C" int mnerf_qkv_window_images(const float* wstream, const int32_t* ew, const float* x_q, const float* x_kv,
                                       int32_t kv_swap, float* q, void* workspace, size_t workspace_bytes, int32_t batch,
                                       int32_t h, int32_t w, int32_t num_splits, int32_t shifted, void* stream) {
  const char* who = "mnerf_qkv_window_images";
  WinGeom G;
  int do_shift;
  if (const int rc = wa_geometry(who, batch, h, w, num_splits, shifted, G, do_shift)) return rc;
  MNERF_REQUIRE(!kv_swap || batch % 2 == 0, MNERF_E_RANGE, "%s: kv_swap needs an even number of sequences, got %d", who, batch);
  if (batch == 0) return MNERF_OK;
  MNERF_REQUIRE(wstream && ew && x_q && x_kv && q, MNERF_E_NULL, "%s: NULL buffer", who);
  MNERF_REQUIRE(mnerf_aligned16(wstream) && mnerf_aligned16(x_q) && mnerf_aligned16(x_kv) && mnerf_aligned16(q), MNERF_E_ALIGN,
                "%s: buffers must be 16-byte aligned", who);
  const size_t need = wa_workspace_bytes(batch, h, w, num_splits);
  MNERF_REQUIRE(workspace && mnerf_aligned16(workspace), MNERF_E_ALIGN, "%s: workspace NULL or not 16-byte aligned", who);
  MNERF_REQUIRE(workspace_bytes >= need, MNERF_E_RANGE, "%s: workspace %zu bytes < %zu", who, workspace_bytes, need);
  QkvImgParams p;
  p.xq = x_q;
  p.xkv = x_kv;
  p.q = q;
  p.wstream = wstream;
  p.G = G;
  p.n_batch = batch;
  p.n_tiles = (G.Lw + WA_KT - 1) / WA_KT;
  p.total_tiles = batch * num_splits * num_splits * p.n_tiles;
  p.kv_swap = kv_swap ? 1 : 0;
  p.img = reinterpret_cast<u32x4*>(workspace);
  p.rec = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + (size_t)p.total_tiles * WA_IMG_BYTES);
  for (int i = 0; i < 3; ++i) p.ew[i] = ew[i];
  const size_t lds = 2 * QKV_SEG_FLOATS * sizeof(float);
  static std::atomic<unsigned long long> attr{0};
  if (mnerf_once_per_device(attr))
    (void)hipFuncSetAttribute((const void*)qkv_images_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(qkv_images_kernel, dim3((p.total_tiles + QKV_NW - 1) / QKV_NW), dim3(QKV_NW * 64), lds, (hipStream_t)stream, p);
  return mnerf_check_launch(who);
}

extern "C" int64_t mnerf_qkv_wstream_floats(void) { return 6 * (int64_t)QKV_SEG_FLOATS; }

extern "C" int mnerf_qkv_projection(const float* wstream, const int32_t* ew, const float* x_q, const float* x_kv,
                                    int32_t kv_swap, float* q, float* k, float* v, int32_t n_seq, int32_t seq_len,
                                    void* stream) {
  const char* who = "mnerf_qkv_projection";
  MNERF_REQUIRE(n_seq >= 0 && seq_len >= 1 && (long long)n_seq * seq_len <= 0x7fffffffLL, MNERF_E_RANGE,
                "%s: n_seq=%d seq_len=%d", who, n_seq, seq_len);
  MNERF_REQUIRE(!kv_swap || n_seq % 2 == 0, MNERF_E_RANGE, "%s: kv_swap needs an even number of sequences, got %d", who, n_seq);
  if (n_seq == 0) return MNERF_OK;
  MNERF_REQUIRE(wstream && ew && x_q && x_kv && q && k && v, MNERF_E_NULL, "%s: NULL buffer", who);
  MNERF_REQUIRE(mnerf_aligned16(wstream) && mnerf_aligned16(x_q) && mnerf_aligned16(x_kv) && mnerf_aligned16(q) &&
                    mnerf_aligned16(k) && mnerf_aligned16(v),
                MNERF_E_ALIGN, "%s: buffers must be 16-byte aligned", who);
  QkvParams p;
  p.xq = x_q;
  p.xkv = x_kv;
  p.out[0] = q;
  p.out[1] = k;
  p.out[2] = v;
  p.wstream = wstream;
  p.n_tokens = n_seq * seq_len;
  p.seq_len = seq_len;
  p.n_seq = n_seq;
  p.kv_swap = kv_swap ? 1 : 0;
  for (int i = 0; i < 3; ++i) p.ew[i] = ew[i];
  const size_t lds = 2 * QKV_SEG_FLOATS * sizeof(float);
  static std::atomic<unsigned long long> attr{0};
  if (mnerf_once_per_device(attr))
    (void)hipFuncSetAttribute((const void*)qkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int per_wg = 32 * QKV_NW;
  hipLaunchKernelGGL(qkv_kernel, dim3((p.n_tokens + per_wg - 1) / per_wg), dim3(QKV_NW * 64), lds, (hipStream_t)stream, p);
  return mnerf_check_launch(who);
}
