// K7 — everything of a GMFlow transformer layer that follows the window attention, as ONE kernel:
//
//   message = norm1(merge(attn))                                     (self-attention layer: done)
//   message = norm2(mlp.2(GELU(mlp.0(cat[source, message]))))        (cross-attention + FFN layer)
//   out     = source + message
//
// Replaces TransformerLayer.forward after the attention (/root/reference/models/gmflow/transformer.py:176-185):
// merge Linear 128->128, LayerNorm, torch.cat, Linear 256->1024, exact-erf GELU, Linear 1024->128, LayerNorm and
// the residual add — 9 eager ops / 4 library GEMM + elementwise launches per layer in the reference.
//
// Same machinery as the fused decoder (decoder.hip): every Linear is evaluated transposed, Y^T[out, token] =
// W . X^T, on v_mfma_f32_32x32x16_f16 with fp32-grade split-fp16 operands (split_f16.hpp); a wave owns 32 tokens,
// the accumulator layout of one stage is the operand layout of the next (weight columns permuted on the host:
// matchnerf_amd/gmflow.py, pack_encoder_block), LayerNorm is an in-lane + one cross-half reduction, and the packed
// weights stream through a ring of 4 x 32 KiB LDS buffers by LDS-DMA (three segments in flight: the problem is small
// — 30,720 tokens at 3 views are 960 wave-tiles for 1,024 SIMDs, one wave per SIMD — so a segment's 48 matrix
// instructions are far too short to hide the copy of the next one), one barrier per segment.
// The 1024 hidden units are produced and consumed 128 at a time (hidden chunk c: 16 K16-steps of mlp.0, GELU,
// 8 K16-steps of mlp.2 accumulated into the 128 outputs), so the [tokens, 1024] activation never exists; the
// per-token operand gain of the second Linear can grow from chunk to chunk, and the output accumulator is rescaled
// by the (exact, power-of-two) ratio whenever it does.
#include "split_f16.hpp"

#define EB_SEG_FLOATS (32 * 256)  // one weight segment: 4 K16-steps x 4 blocks x [hi | lo] x 1 KiB = 32 KiB
#define EB_C 128                  // d_model
#define EB_HIDDEN 1024            // 2 * d_model * ffn_dim_expansion
#define EB_CHUNKS (EB_HIDDEN / 128)
#define EB_LN_EPS 1e-5f
#define EB_NBUF 4                 // LDS ring: segment s lives in buffer s % 4, segments s+1 .. s+3 are in flight

struct EbParams {
  const float* attn;
  const float* source;
  float* out;
  const float* wstream;
  const float* ln;  // [4][128]: norm1 weight | norm1 bias | norm2 weight | norm2 bias
  int n_tokens;
  int ffn;
  int ew_merge, ew_w1, ew_w2;
  // training forward (template SAVE): what the layer's backward would otherwise re-evaluate with three GEMMs -
  float* save_m1;  // [n_tokens][128]  merge's output before norm1
  float* save_z1;  // [n_tokens][1024] mlp.0's output before the GELU       (FFN layers)
  float* save_m2;  // [n_tokens][128]  mlp.2's output before norm2          (FFN layers)
};

template <int NW>
__device__ __forceinline__ void eb_prefetch(const float* __restrict__ wstream, int seg, int n_seg, unsigned base,
                                            int wave, int lane) {
  if (seg >= n_seg) return;
  const float* src = wstream + (size_t)seg * EB_SEG_FLOATS + lane * 4;
  glds_segment(src, base, EB_SEG_FLOATS / 256, wave, NW);
}

// LayerNorm over the 128 features of a token held in accumulator layout: register r of block m of lane (n, half) is
// feature 32 m + (r & 3) + 8 (r >> 2) + 4 half of token n.  v <- (v - mean) * rstd * w + b.
__device__ __forceinline__ void eb_layer_norm(f32x16 (&v)[4], const float* ln_w, const float* ln_b, int hl) {
  float s = 0.0f;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[m][r];
  s += __shfl_xor(s, 32, 64);
  const float mean = s * (1.0f / EB_C);
  float q = 0.0f;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = v[m][r] - mean;
      q += d * d;
    }
  q += __shfl_xor(q, 32, 64);
  const float rstd = 1.0f / sqrtf(q * (1.0f / EB_C) + EB_LN_EPS);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 w4 = *reinterpret_cast<const float4*>(ln_w + 32 * m + 8 * g + 4 * hl);
      const float4 b4 = *reinterpret_cast<const float4*>(ln_b + 32 * m + 8 * g + 4 * hl);
      v[m][4 * g + 0] = (v[m][4 * g + 0] - mean) * rstd * w4.x + b4.x;
      v[m][4 * g + 1] = (v[m][4 * g + 1] - mean) * rstd * w4.y + b4.y;
      v[m][4 * g + 2] = (v[m][4 * g + 2] - mean) * rstd * w4.z + b4.z;
      v[m][4 * g + 3] = (v[m][4 * g + 3] - mean) * rstd * w4.w + b4.w;
    }
}

// nn.GELU() (exact): 0.5 x (1 + erf(x / sqrt 2)).  The kernel evaluates it 512 times per lane and FFN layer, and the
// library erff (~50 VALU instructions, two code paths) made the kernel VALU-bound: 27.8 k VALU against 2.5 k matrix
// instructions per wave (PMC, profiles/r2_07_encoder_pmc_summary.txt).  erf here is Abramowitz & Stegun 7.1.26,
//   erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),  t = 1 / (1 + 0.3275911 z),  z >= 0,
// |error| <= 1.5e-7 ABSOLUTE - which is what GELU needs: erf enters as 1 + erf, so the error of the result is
// <= 0.75e-7 |x|, a relative 1.5e-7 wherever GELU is not itself vanishing.  16 instructions, no branch.
__device__ __forceinline__ float eb_gelu(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = 1.061405429f;
  p = __builtin_fmaf(p, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float erfc_z = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);  // 1 - erf(|x| / sqrt 2)
  // 1 + erf(x / sqrt 2) = 2 - erfc_z for x >= 0, erfc_z for x < 0
  return 0.5f * x * (x >= 0.0f ? 2.0f - erfc_z : erfc_z);
}

// SAVE (round 6, the training forward of an FFN layer; the inference instance's code is unchanged): the pre-GELU hidden
// activations and mlp.2's output are written out for mnerf_encoder_layer_backward_saved.  A chunk's 128 hidden units leave as 16
// float4 per lane (registers 4 g .. 4 g + 3 of block m are four consecutive units).  Ordinary stores count in vmcnt next to the
// weight requests and need not retire in order with them, so the partial waits of eb_wait_next are not safe while stores are in
// flight: the segment that follows a chunk's stores ends with a full wait (one drained prefetch per chunk: +7 % on this instance).
template <int NW, bool SAVE = false>
__global__ __launch_bounds__(NW * 64, 1) void encoder_block_kernel(EbParams P) {
  extern __shared__ __attribute__((aligned(16))) float eb_smem[];
  const unsigned wbuf0_lds = __builtin_amdgcn_groupstaticsize();
  float* ln_lds = eb_smem + EB_NBUF * EB_SEG_FLOATS;  // [4][128]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hl = lane >> 5;
  const int n_seg = P.ffn ? 2 + EB_CHUNKS * 6 : 2;

  for (int i = tid; i < 4 * EB_C; i += NW * 64) ln_lds[i] = P.ln[i];
  int seg = 0;
#pragma unroll
  for (int i = 0; i < EB_NBUF - 1; ++i)
    eb_prefetch<NW>(P.wstream, i, n_seg, wbuf0_lds + (unsigned)i * EB_SEG_FLOATS * 4u, wave, lane);

  const int tok_raw = (blockIdx.x * NW + wave) * 32 + n;
  const bool tok_ok = tok_raw < P.n_tokens;
  const int tok = tok_ok ? tok_raw : P.n_tokens - 1;
  const float* arow = P.attn + (size_t)tok * EB_C + 8 * hl;    // this lane's half of a K16-step: features 16 t + 8 hl + j
  const float* srow = P.source + (size_t)tok * EB_C + 8 * hl;

  // A wave issues EB_SEG_FLOATS / 256 / NW = 8 DMA pieces per segment, and memory operations complete in order: with
  // k later segments in flight, "at most 8 k operations outstanding" means the next segment has landed (other loads
  // issued in between can only make the wait longer, never shorter).
  constexpr int PIECES = EB_SEG_FLOATS / 256 / NW;
  static_assert(PIECES * (EB_NBUF - 2) < 64, "vmcnt is a 6-bit counter");
  auto eb_wait_next = [&](int s) {  // segment s + 1 complete (this wave's pieces)
    const int last = n_seg - 1 < s + EB_NBUF - 1 ? n_seg - 1 : s + EB_NBUF - 1;  // newest segment issued so far
    const int ahead = last - (s + 1);
    if (ahead >= 2)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
    else if (ahead == 1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
#define EB_CUR (wbuf0_lds + (unsigned)(seg & (EB_NBUF - 1)) * EB_SEG_FLOATS * 4u)
#define EB_BEGIN()                                                                                                   \
  eb_prefetch<NW>(P.wstream, seg + EB_NBUF - 1, n_seg,                                                               \
                  wbuf0_lds + (unsigned)((seg + EB_NBUF - 1) & (EB_NBUF - 1)) * EB_SEG_FLOATS * 4u, wave, lane)
#define EB_END()        \
  do {                  \
    eb_wait_next(seg);  \
    __syncthreads();    \
    ++seg;              \
  } while (0)
#define EB_END_DRAIN()                                   \
  do {                                                   \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
    __syncthreads();                                     \
    ++seg;                                               \
  } while (0)

  // ---------------------------------------------------------------- merge: 128 -> 128 on the attention output
  f32x16 m1[4];
  {
    float a_in[64];
    float amax = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 lo4 = *reinterpret_cast<const float4*>(arow + 16 * t);
      const float4 hi4 = *reinterpret_cast<const float4*>(arow + 16 * t + 4);
      a_in[8 * t + 0] = lo4.x; a_in[8 * t + 1] = lo4.y; a_in[8 * t + 2] = lo4.z; a_in[8 * t + 3] = lo4.w;
      a_in[8 * t + 4] = hi4.x; a_in[8 * t + 5] = hi4.y; a_in[8 * t + 6] = hi4.z; a_in[8 * t + 7] = hi4.w;
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) amax = fmaxf(amax, fabsf(a_in[i]));
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    const int em = gain_exp(amax);
    const float mult = pow2i(em);
#pragma unroll
    for (int m = 0; m < 4; ++m) m1[m] = (f32x16)(0.0f);
    eb_wait_next(-1);
    __syncthreads();  // weight segment 0 and the LayerNorm parameters are in LDS
#pragma unroll
    for (int sgi = 0; sgi < 2; ++sgi) {
      EB_BEGIN();
      float v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = a_in[32 * sgi + i];
      ksteps_h<4, 4>(m1, EB_CUR, lane, v, mult);
      EB_END();
    }
    const float cm = pow2i(-(P.ew_merge + em));
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) m1[m][r] *= cm;
  }
  if constexpr (SAVE) {
    // (no weight request is waited for by count between here and the first FFN segment's end other than through EB_END's partial
    // wait: the stores go out BEHIND the drain below)
    if (tok_ok) {
      float* mp = P.save_m1 + (size_t)tok * EB_C + 4 * hl;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(mp + 32 * m + 8 * g) = make_float4(m1[m][4 * g + 0], m1[m][4 * g + 1], m1[m][4 * g + 2], m1[m][4 * g + 3]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores and weight requests do not retire in order with each other
  }
  eb_layer_norm(m1, ln_lds, ln_lds + EB_C, hl);

  f32x16 y[4];
  if (!P.ffn) {
#pragma unroll
    for (int m = 0; m < 4; ++m) y[m] = m1[m];
  } else {
    // ---------------------------------------------------------------- FFN on cat[source, message]
    // one true gain for both operand sets of mlp.0 (they share the accumulator), from the larger of the two maxima
    // the token's 128 source features in K16-step order, loaded ONCE: an ordinary global load inside the chunk loop
    // would make hipcc drain every weight DMA in flight in front of its first use (it counts only its own loads),
    // i.e. undo the three-segment prefetch
    float s_in[64];
    float smax = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 lo4 = *reinterpret_cast<const float4*>(srow + 16 * t);
      const float4 hi4 = *reinterpret_cast<const float4*>(srow + 16 * t + 4);
      s_in[8 * t + 0] = lo4.x; s_in[8 * t + 1] = lo4.y; s_in[8 * t + 2] = lo4.z; s_in[8 * t + 3] = lo4.w;
      s_in[8 * t + 4] = hi4.x; s_in[8 * t + 5] = hi4.y; s_in[8 * t + 6] = hi4.z; s_in[8 * t + 7] = hi4.w;
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) smax = fmaxf(smax, fabsf(s_in[i]));
    const float cmax = fmaxf(fmaxf(smax, __shfl_xor(smax, 32, 64)), sample_absmax<4>(m1));
    const int eg1 = gain_exp(cmax);
    const float mult1 = pow2i(eg1);
    const float c1 = pow2i(-(P.ew_w1 + eg1));
    // both operand sets of mlp.0, split once for all eight hidden chunks: steps 0..7 source, 8..15 message
    PartsH xp[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = s_in[8 * t + j];
      xp[t] = split8h(v8, mult1);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = m1[t >> 1][8 * (t & 1) + j];
      xp[8 + t] = split8h(v8, mult1);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) y[m] = (f32x16)(0.0f);
    int eg2 = 127;  // running operand-gain exponent of mlp.2 (the accumulator y holds 2^(ew_w2 + eg2) * true)
    for (int c = 0; c < EB_CHUNKS; ++c) {
      f32x16 hd[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) hd[m] = (f32x16)(0.0f);
      // mlp.0, hidden chunk c: four segments of four K16-steps over cat[source, message]
#pragma unroll
      for (int sgi = 0; sgi < 4; ++sgi) {
        EB_BEGIN();
        ksteps_presplit<4, 4>(hd, EB_CUR, lane, xp + 4 * sgi);
        EB_END();
      }
      // GELU; the chunk's largest activation sets (or lowers) the operand gain of mlp.2
      if constexpr (SAVE) {
        if (tok_ok) {
          float* zp = P.save_z1 + (size_t)tok * EB_HIDDEN + 128 * c + 4 * hl;
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(zp + 32 * m + 8 * g) =
                  make_float4(hd[m][4 * g + 0] * c1, hd[m][4 * g + 1] * c1, hd[m][4 * g + 2] * c1, hd[m][4 * g + 3] * c1);
        }
      }
      float gmax = 0.0f;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float g = eb_gelu(hd[m][r] * c1);
          hd[m][r] = g;
          gmax = fmaxf(gmax, fabsf(g));
        }
      gmax = fmaxf(gmax, __shfl_xor(gmax, 32, 64));
      const int egc = gain_exp(gmax);
      if (egc < eg2) {  // larger activations than any chunk before: rescale what has been accumulated (exact)
        if (c > 0) {
          const float resc = pow2i(egc - eg2);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[m][r] *= resc;
        }
        eg2 = egc;
      }
      const float mult2 = pow2i(eg2);
      // mlp.2: the 128 hidden units of this chunk into the 128 outputs
#pragma unroll
      for (int sgi = 0; sgi < 2; ++sgi) {
        EB_BEGIN();
        kblock_h<4>(y, EB_CUR, lane, hd[2 * sgi], mult2);
        kblock_h<4>(y, EB_CUR + 8 * H16_UNIT_BYTES, lane, hd[2 * sgi + 1], mult2);
        if (SAVE && sgi == 0) EB_END_DRAIN();  // (the chunk's stores are in flight)
        else EB_END();
      }
    }
    const float c2 = pow2i(-(P.ew_w2 + eg2));
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[m][r] *= c2;
    if constexpr (SAVE) {
      if (tok_ok) {
        float* mp = P.save_m2 + (size_t)tok * EB_C + 4 * hl;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(mp + 32 * m + 8 * g) = make_float4(y[m][4 * g + 0], y[m][4 * g + 1], y[m][4 * g + 2], y[m][4 * g + 3]);
      }
    }
    eb_layer_norm(y, ln_lds + 2 * EB_C, ln_lds + 3 * EB_C, hl);
  }
#undef EB_CUR
#undef EB_BEGIN
#undef EB_END
#undef EB_END_DRAIN

  // ---------------------------------------------------------------- out = source + message (accumulator layout: 16 float4)
  if (tok_ok) {
    const float* sp = P.source + (size_t)tok * EB_C + 4 * hl;
    float* op = P.out + (size_t)tok * EB_C + 4 * hl;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 s4 = *reinterpret_cast<const float4*>(sp + 32 * m + 8 * g);
        *reinterpret_cast<float4*>(op + 32 * m + 8 * g) =
            make_float4(s4.x + y[m][4 * g + 0], s4.y + y[m][4 * g + 1], s4.z + y[m][4 * g + 2], s4.w + y[m][4 * g + 3]);
      }
  }
}

extern "C" int64_t mnerf_encoder_block_wstream_floats(int32_t ffn) {
  return (int64_t)(ffn ? 2 + EB_CHUNKS * 6 : 2) * EB_SEG_FLOATS;
}

static int encoder_block_impl(const mnerf_encoder_layer* blk, const float* attn, const float* source, float* out, float* save_m1,
                              float* save_z1, float* save_m2, int32_t n_tokens, void* stream) {
  MNERF_REQUIRE(blk, MNERF_E_NULL, "mnerf_encoder_block: blk is NULL");
  MNERF_REQUIRE(n_tokens >= 0, MNERF_E_RANGE, "mnerf_encoder_block: n_tokens=%d", n_tokens);
  if (n_tokens == 0) return MNERF_OK;
  MNERF_REQUIRE(blk->wstream && blk->ln && attn && source && out, MNERF_E_NULL, "mnerf_encoder_block: NULL buffer");
  MNERF_REQUIRE(mnerf_aligned16(blk->wstream) && mnerf_aligned16(attn) && mnerf_aligned16(source) && mnerf_aligned16(out) &&
                    mnerf_aligned16(blk->ln),
                MNERF_E_ALIGN, "mnerf_encoder_block: buffers must be 16-byte aligned");
  MNERF_REQUIRE(blk->wstream_floats == mnerf_encoder_block_wstream_floats(blk->ffn), MNERF_E_RANGE,
                "mnerf_encoder_block: wstream has %lld floats, expected %lld", (long long)blk->wstream_floats,
                (long long)mnerf_encoder_block_wstream_floats(blk->ffn));
  EbParams p;
  p.attn = attn;
  p.source = source;
  p.out = out;
  p.wstream = blk->wstream;
  p.ln = blk->ln;
  p.n_tokens = n_tokens;
  p.ffn = blk->ffn ? 1 : 0;
  p.ew_merge = blk->ew_merge;
  p.ew_w1 = blk->ew_w1;
  p.ew_w2 = blk->ew_w2;
  p.save_m1 = save_m1;
  p.save_z1 = save_z1;
  p.save_m2 = save_m2;
  constexpr int NW = 4;
  const size_t lds = ((size_t)EB_NBUF * EB_SEG_FLOATS + 4 * EB_C) * sizeof(float);
  static std::atomic<unsigned long long> attr_set{0};
  if (mnerf_once_per_device(attr_set))
    (void)hipFuncSetAttribute((const void*)encoder_block_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = (n_tokens + NW * 32 - 1) / (NW * 32);
  if (save_m1) {
    static std::atomic<unsigned long long> attr_save{0};
    if (mnerf_once_per_device(attr_save))
      (void)hipFuncSetAttribute((const void*)encoder_block_kernel<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((encoder_block_kernel<NW, true>), dim3(grid), dim3(NW * 64), lds, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(encoder_block_kernel<NW>, dim3(grid), dim3(NW * 64), lds, (hipStream_t)stream, p);
  }
  return mnerf_check_launch("mnerf_encoder_block");
}

extern "C" int mnerf_encoder_block(const mnerf_encoder_layer* blk, const float* attn, const float* source, float* out,
                                   int32_t n_tokens, void* stream) {
  return encoder_block_impl(blk, attn, source, out, nullptr, nullptr, nullptr, n_tokens, stream);
}

// Training forward: the same result bit for bit, and the pre-norm activations written out for mnerf_encoder_layer_backward_saved:
// merge's output before norm1 (m1 [n_tokens][128]) and, for a layer with an FFN, mlp.0's output before the GELU (z1
// [n_tokens][1024]) and mlp.2's output before norm2 (m2 [n_tokens][128]; both NULL for a layer without).
extern "C" int mnerf_encoder_block_save(const mnerf_encoder_layer* blk, const float* attn, const float* source, float* out,
                                        float* m1, float* z1, float* m2, int32_t n_tokens, void* stream) {
  MNERF_REQUIRE(blk, MNERF_E_NULL, "mnerf_encoder_block_save: blk is NULL");
  if (n_tokens == 0) return MNERF_OK;
  MNERF_REQUIRE(m1 && mnerf_aligned16(m1), MNERF_E_NULL, "mnerf_encoder_block_save: m1 NULL or not 16-byte aligned");
  MNERF_REQUIRE(!blk->ffn || (z1 && m2 && mnerf_aligned16(z1) && mnerf_aligned16(m2)), MNERF_E_NULL,
                "mnerf_encoder_block_save: z1 / m2 NULL or not 16-byte aligned");
  return encoder_block_impl(blk, attn, source, out, m1, blk->ffn ? z1 : nullptr, blk->ffn ? m2 : nullptr, n_tokens, stream);
}
