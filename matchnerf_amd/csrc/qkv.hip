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
#include "split_f16.hpp"

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
    for (int p = wave; p < QKV_SEG_FLOATS / 256; p += QKV_NW)
      glds16(src + p * 256, __builtin_amdgcn_readfirstlane(dst + (unsigned)p * 1024u));
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
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(dst + 32 * m + 8 * g) =
              make_float4(acc[m][4 * g] * cm, acc[m][4 * g + 1] * cm, acc[m][4 * g + 2] * cm, acc[m][4 * g + 3] * cm);
    }
  }
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
