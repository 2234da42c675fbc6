// K3+K4+K5 — conditional radiance MLP + per-ray transformer + alpha compositing as ONE
// ray-chunk kernel for gfx950 (fp32, exact-f32 MFMA).
//
// Replaces, per chunk of rays (paths relative to /root/reference):
//   models/matchnerf.py:118-132        NDC warp w.r.t. source view 0, view-dir rotation
//   models/rfdecoder/cond_nerf.py:52-100   CondNeRF.forward (posenc, FiLM-modulated MLP, heads)
//   models/rfdecoder/ray_transformer.py:14-26, 49-79   4-head attention along the ray + LN
//   models/rfdecoder/nerf.py:101-124   NeRF.composite
// The reference runs these as ~60 eager ops that materialise [R,S,128] activations per layer
// (134 MB each at R=4096,S=64) and a [R,4,S,S] score tensor (268 MB).  Here a workgroup owns
// a tile of TILE = 32*NW samples (whole rays), activations never leave registers, the
// ray-attention K/V and per-sample (rgb,sigma) live in LDS, and only 5 floats per ray are
// written to HBM.
//
// ---- MFMA formulation (the part that is specific to CDNA) --------------------------------
// Every Linear is evaluated TRANSPOSED:  Y^T[out, sample] = W[out, in] . H^T[in, sample]
// with v_mfma_f32_32x32x2_f32:  A = W tile (32 outs x 2 ins), B = H^T (2 ins x 32 samples).
// A wave owns 32 samples (N = lane&31) and all 128 outputs (4 M-blocks -> 4 x 16 accumulator
// VGPRs).  The C/D layout puts output row (r&3)+8*(r>>2)+4*(lane>>5) of block m in register
// r of lane (n, half) — which is exactly the B-operand layout the NEXT layer needs if its
// K-steps are taken in the order "register r of block m": lower half-wave supplies input
// feature f_lo(m,r), upper half supplies f_hi(m,r) = f_lo + 4.  Since a dot product does not
// care about the order of its terms, the host packs each weight matrix with its columns
// permuted to that order (matchnerf_amd/cond_nerf.py:pack_decoder), and the whole 6-layer
// MLP + heads chains accumulator -> operand with NO transpose, shuffle or LDS round trip.
// Biases ride along as one extra K-step whose B operand is the constant (1 | 0); the FiLM
// multiplier (pts_bias(cond), cond_nerf.py:62) is itself computed by an MFMA stage and kept
// in 64 VGPRs; the epilogue of a layer is one v_mul + v_max per accumulator register.
//
// Weights: 130k floats (521 KB) cannot live in LDS, so the packed A-fragment stream is cut
// into segments of <= 33 KiB that every wave consumes in the same order; segment i+1 is
// DMA'd global->LDS (global_load_lds_dwordx4, no VGPRs) into the other half of a double
// buffer while segment i feeds the MFMAs; one workgroup barrier per segment.
// With NW=4 a workgroup needs 76 KiB of LDS and <=256 VGPRs, so two workgroups share a CU
// (2 waves/SIMD) and de-synchronise: one's VALU phases (posenc, attention, compositing)
// overlap the other's MFMA phases.
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SEG_CAP_FLOATS (33 * 256)  // one LDS weight buffer: 33 KiB
#define MAX_SEGS 40
#define SMALL_FIXED 1360  // floats of `small` before the optional ray-posenc table

struct DecSched {
  int n_seg;
  int film_steps, enc_steps;
  int seg_off[MAX_SEGS];     // float offset of the segment in wstream (multiple of 256)
  int seg_floats[MAX_SEGS];  // padded to a multiple of 256 floats (1 KiB DMA pieces)
  int seg_steps[MAX_SEGS];
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- weight-segment DMA: each wave copies 1 KiB pieces, LDS dest = uniform base + lane*16
template <int NW>
__device__ __forceinline__ void prefetch_segment(const float* __restrict__ wstream,
                                                 const DecSched& sch, int seg, float* lds_buf,
                                                 int wave, int lane) {
  if (seg >= sch.n_seg) return;
  const float* src = wstream + sch.seg_off[seg];
  const int pieces = sch.seg_floats[seg] >> 8;
  for (int p = wave; p < pieces; p += NW) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(src + p * 256 + lane * 4),
        (__attribute__((address_space(3))) void*)(lds_buf + p * 256), 16, 0, 0);
  }
}

// one K-step against 4 / 2 / 1 M-blocks; A fragments laid out [step][lane][nmb]
__device__ __forceinline__ void step4(f32x16 (&acc)[4], const float* seg, int step, int lane,
                                      float b) {
  const float4 a = reinterpret_cast<const float4*>(seg)[step * 64 + lane];
  acc[0] = mfma(a.x, b, acc[0]);
  acc[1] = mfma(a.y, b, acc[1]);
  acc[2] = mfma(a.z, b, acc[2]);
  acc[3] = mfma(a.w, b, acc[3]);
}
__device__ __forceinline__ void step2(f32x16 (&acc)[2], const float* seg, int step, int lane,
                                      float b) {
  const float2 a = reinterpret_cast<const float2*>(seg)[step * 64 + lane];
  acc[0] = mfma(a.x, b, acc[0]);
  acc[1] = mfma(a.y, b, acc[1]);
}
__device__ __forceinline__ void step1(f32x16& acc, const float* seg, int step, int lane, float b) {
  acc = mfma(seg[step * 64 + lane], b, acc);
}

// 32 K-steps fed from two 16-register accumulator blocks of the previous layer
template <int NMB>
__device__ __forceinline__ void steps_from_regs(f32x16 (&acc)[NMB], const float* seg, int step0,
                                                int lane, const f32x16& h0, const f32x16& h1);
template <>
__device__ __forceinline__ void steps_from_regs<4>(f32x16 (&acc)[4], const float* seg, int step0,
                                                   int lane, const f32x16& h0, const f32x16& h1) {
#pragma unroll
  for (int r = 0; r < 16; ++r) step4(acc, seg, step0 + r, lane, h0[r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) step4(acc, seg, step0 + 16 + r, lane, h1[r]);
}
template <>
__device__ __forceinline__ void steps_from_regs<2>(f32x16 (&acc)[2], const float* seg, int step0,
                                                   int lane, const f32x16& h0, const f32x16& h1) {
#pragma unroll
  for (int r = 0; r < 16; ++r) step2(acc, seg, step0 + r, lane, h0[r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) step2(acc, seg, step0 + 16 + r, lane, h1[r]);
}
template <>
__device__ __forceinline__ void steps_from_regs<1>(f32x16 (&acc)[1], const float* seg, int step0,
                                                   int lane, const f32x16& h0, const f32x16& h1) {
#pragma unroll
  for (int r = 0; r < 16; ++r) step1(acc[0], seg, step0 + r, lane, h0[r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) step1(acc[0], seg, step0 + 16 + r, lane, h1[r]);
}

// B operand of positional-encoding step t for this lane (cond_nerf.py:108-116 legacy /
// nerf.py:126-133 non-legacy; the packer maps weight columns accordingly):
//   t < 3L : arg = x_{t%3} * 2^{t/3} (* pi)  ->  lower half sin(arg), upper half cos(arg)
//   t = 3L : (x | y)      t = 3L+1 : (z | 1)   [the 1 multiplies the packed bias column]
__device__ __forceinline__ float enc_operand(int t, int L3, int hl, float x, float y, float z,
                                             float freq_mul) {
  if (t < L3) {
    const int l = t / 3, c = t - 3 * l;
    const float xc = (c == 0) ? x : ((c == 1) ? y : z);
    const float arg = xc * (ldexpf(1.0f, l) * freq_mul);
    float s, co;
    sincosf(arg, &s, &co);
    return hl ? co : s;
  }
  if (t == L3) return hl ? y : x;
  return hl ? 1.0f : z;
}

template <int NW>
struct Smem {
  static constexpr int TILE = NW * 32;
  static constexpr int W_FLOATS = 2 * SEG_CAP_FLOATS;
  static constexpr int A_FLOATS = TILE * 16;
  static constexpr int RS_FLOATS = TILE * 4;
  // attention scratch (K/V + small params): aliased on weight buffer 1 for NW=4 (22 KiB <= 33),
  // dedicated for NW=8 (1 workgroup per CU anyway)
  static constexpr int KV_FLOATS = TILE * 4 * 8;
  static constexpr int ATT_FLOATS = KV_FLOATS + SMALL_FIXED;
  static constexpr bool ATT_ALIASED = ATT_FLOATS <= SEG_CAP_FLOATS;
  static constexpr int TOTAL_FLOATS = W_FLOATS + A_FLOATS + RS_FLOATS + (ATT_ALIASED ? 0 : ATT_FLOATS);
};

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void decoder_kernel(
    mnerf_decoder D, DecSched sch, mnerf_view view0, mnerf_rays R, int Sp,
    const float* __restrict__ cond, float* __restrict__ out_rgb, float* __restrict__ out_depth,
    float* __restrict__ out_opacity, float* __restrict__ dbg_rgb_s, float* __restrict__ dbg_sigma) {
  using SM = Smem<NW>;
  constexpr int TILE = SM::TILE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wbuf0 = smem;
  float* wbuf1 = smem + SEG_CAP_FLOATS;
  float* a_lds = smem + SM::W_FLOATS;                 // [TILE][16]  alpha features
  float* rs_lds = a_lds + SM::A_FLOATS;               // [TILE][4]   rgb.xyz, sigma.w
  float* att = SM::ATT_ALIASED ? wbuf1 : (rs_lds + SM::RS_FLOATS);
  float* kv_lds = att;                                // [rays][4 heads][Sp][8]  (k0..3, v0..3)
  float* sm_lds = att + SM::KV_FLOATS;                // copy of small[0:SMALL_FIXED]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int n = lane & 31, hl = lane >> 5;
  const int S = R.n_samples;
  const int rays_per_tile = TILE / Sp;
  const int n_tiles = (R.n_rays + rays_per_tile - 1) / rays_per_tile;
  const int L3 = 3 * D.L_3D;
  const int CS = D.cond_stride;
  const float freq_mul = R.legacy_coord ? 1.0f : 3.14159265358979323846f;
  const float wm1 = (float)(R.width - 1), hm1 = (float)(R.height - 1);

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // ------------------------------------------------------------ per-lane sample identity
    const int s_local = wave * 32 + n;
    const int ray_t = s_local / Sp;                 // ray within the tile
    const int jp = s_local - ray_t * Sp;            // padded sample slot
    const int ray_raw = tile * rays_per_tile + ray_t;
    const bool ray_ok = ray_raw < R.n_rays;
    const int ray = ray_ok ? ray_raw : (R.n_rays - 1);
    const int j = jp < S ? jp : (S - 1);            // padded slots recompute the last sample
    const size_t gs = (size_t)ray * S + j;          // global sample index

    const RayGeom g = make_ray(R, ray);
    const float dpt = sample_depth(R, ray, j);
    float x, y, z;
    {
      float wx_, wy_, wz_;
      ray_point(g, dpt, wx_, wy_, wz_);
      project(view0, wx_, wy_, wz_, wm1, hm1, x, y, z);
    }
    // view direction in the frame of source view 0 (matchnerf.py:129-131)
    const float rn = fmaxf(sqrtf(g.rx * g.rx + g.ry * g.ry + g.rz * g.rz), 1e-12f);
    const float ux = g.rx / rn, uy = g.ry / rn, uz = g.rz / rn;
    const float dx = ux * view0.extr[0] + uy * view0.extr[1] + uz * view0.extr[2];
    const float dy = ux * view0.extr[4] + uy * view0.extr[5] + uz * view0.extr[6];
    const float dz = ux * view0.extr[8] + uy * view0.extr[9] + uz * view0.extr[10];

    int seg = 0;   // running segment index; segment k lives in buffer (k & 1)
    prefetch_segment<NW>(D.wstream, sch, 0, wbuf0, wave, lane);
    __syncthreads();

#define CUR_BUF ((seg & 1) ? wbuf1 : wbuf0)
#define NXT_BUF ((seg & 1) ? wbuf0 : wbuf1)
#define SEG_BEGIN() prefetch_segment<NW>(D.wstream, sch, seg + 1, NXT_BUF, wave, lane)
#define SEG_END() \
  do {            \
    __syncthreads(); \
    ++seg;        \
  } while (0)

    // ------------------------------------------------------------ FiLM = pts_bias(cond)
    f32x16 film[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) film[m] = (f32x16)(0.0f);
    {
      const float* crow = cond + gs * CS + (size_t)hl * sch.film_steps;  // this half's inputs
      int done = 0;
      while (done < sch.film_steps) {
        const int ns = sch.seg_steps[seg];
        SEG_BEGIN();
        const float* wseg = CUR_BUF;
        for (int t0 = 0; t0 < ns; t0 += 4) {
          const float4 c4 = *reinterpret_cast<const float4*>(crow + done + t0);
          step4(film, wseg, t0 + 0, lane, c4.x);
          step4(film, wseg, t0 + 1, lane, c4.y);
          step4(film, wseg, t0 + 2, lane, c4.z);
          step4(film, wseg, t0 + 3, lane, c4.w);
        }
        done += ns;
        SEG_END();
      }
    }

    // ------------------------------------------------------------ layer 0: enc -> 128
    f32x16 acc[4], h[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
    {
      int done = 0;
      while (done < sch.enc_steps) {
        const int ns = sch.seg_steps[seg];
        SEG_BEGIN();
        const float* wseg = CUR_BUF;
        for (int t = 0; t < ns; ++t)
          step4(acc, wseg, t, lane, enc_operand(done + t, L3, hl, x, y, z, freq_mul));
        done += ns;
        SEG_END();
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f);

    // ------------------------------------------------------------ layers 1..4: 128 -> 128
    for (int layer = 1; layer <= 4; ++layer) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
      SEG_BEGIN();
      steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[0], h[1]);
      SEG_END();
      SEG_BEGIN();
      steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[2], h[3]);
      step4(acc, CUR_BUF, 32, lane, hl ? 0.0f : 1.0f);  // bias column
      SEG_END();
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f);
    }

    // ------------------------------------------------------------ layer 5: [enc, h] -> 128
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
    {
      int done = 0;
      while (done < sch.enc_steps) {
        const int ns = sch.seg_steps[seg];
        SEG_BEGIN();
        const float* wseg = CUR_BUF;
        for (int t = 0; t < ns; ++t)
          step4(acc, wseg, t, lane, enc_operand(done + t, L3, hl, x, y, z, freq_mul));
        done += ns;
        SEG_END();
      }
    }
    SEG_BEGIN();
    steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[0], h[1]);
    SEG_END();
    SEG_BEGIN();
    steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[2], h[3]);
    SEG_END();
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[m][r] = fmaxf(acc[m][r] * film[m][r], 0.0f);

    // ------------------------------------------------------------ alpha head: 128 -> 16
    {
      f32x16 al[1];
      al[0] = (f32x16)(0.0f);
      SEG_BEGIN();
      const float* wseg = CUR_BUF;
      steps_from_regs<1>(al, wseg, 0, lane, h[0], h[1]);
      steps_from_regs<1>(al, wseg, 32, lane, h[2], h[3]);
      step1(al[0], wseg, 64, lane, hl ? 0.0f : 1.0f);
      // rows 0..15 <-> registers 0..7: feature o = (r&3) + 8*(r>>2) + 4*hl
      float av[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float t = al[0][r];
        t = D.raytrans_elu ? (t > 0.0f ? t : (expf(t) - 1.0f)) : fmaxf(t, 0.0f);
        av[r] = t;
      }
      if (D.raytrans_posenc) {
        const float* tab = D.small_ + SMALL_FIXED + (size_t)j * 16;
#pragma unroll
        for (int r = 0; r < 8; ++r) av[r] += tab[(r & 3) + 8 * (r >> 2) + 4 * hl];
      }
      float4* dst = reinterpret_cast<float4*>(a_lds + s_local * 16);
      dst[hl] = make_float4(av[0], av[1], av[2], av[3]);          // features 0-3 | 4-7
      dst[2 + hl] = make_float4(av[4], av[5], av[6], av[7]);      // features 8-11 | 12-15
      SEG_END();
    }

    // ------------------------------------------------------------ feature_linear: 128 -> 128
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x16)(0.0f);
    SEG_BEGIN();
    steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[0], h[1]);
    SEG_END();
    SEG_BEGIN();
    steps_from_regs<4>(acc, CUR_BUF, 0, lane, h[2], h[3]);
    step4(acc, CUR_BUF, 32, lane, hl ? 0.0f : 1.0f);
    SEG_END();

    // ------------------------------------------------------------ views_linear: [feat, dir] -> 64
    f32x16 hv[2];
    hv[0] = (f32x16)(0.0f);
    hv[1] = (f32x16)(0.0f);
    {
      SEG_BEGIN();
      const float* wseg = CUR_BUF;
      steps_from_regs<2>(hv, wseg, 0, lane, acc[0], acc[1]);
      steps_from_regs<2>(hv, wseg, 32, lane, acc[2], acc[3]);
      step2(hv, wseg, 64, lane, hl ? dy : dx);
      step2(hv, wseg, 65, lane, hl ? 1.0f : dz);
      SEG_END();
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) hv[m][r] = fmaxf(hv[m][r], 0.0f);

    // ------------------------------------------------------------ rgb_linear: 64 -> 3, sigmoid
    {
      f32x16 c3[1];
      c3[0] = (f32x16)(0.0f);
      SEG_BEGIN();  // no-op past the last segment
      const float* wseg = CUR_BUF;
      steps_from_regs<1>(c3, wseg, 0, lane, hv[0], hv[1]);
      step1(c3[0], wseg, 32, lane, hl ? 0.0f : 1.0f);
      if (hl == 0) {
        const float cr = 1.0f / (1.0f + expf(-c3[0][0]));
        const float cg = 1.0f / (1.0f + expf(-c3[0][1]));
        const float cb = 1.0f / (1.0f + expf(-c3[0][2]));
        rs_lds[s_local * 4 + 0] = cr;
        rs_lds[s_local * 4 + 1] = cg;
        rs_lds[s_local * 4 + 2] = cb;
        if (dbg_rgb_s && ray_ok && jp < S) {
          dbg_rgb_s[gs * 3 + 0] = cr;
          dbg_rgb_s[gs * 3 + 1] = cg;
          dbg_rgb_s[gs * 3 + 2] = cb;
        }
      }
      SEG_END();  // all waves are done with the weight buffers (att may alias wbuf1)
    }
#undef CUR_BUF
#undef NXT_BUF
#undef SEG_BEGIN
#undef SEG_END

    // ============================================================ ray transformer (K4)
    for (int i = tid; i < SMALL_FIXED; i += NW * 64) sm_lds[i] = D.small_[i];
    float a16[16];
    {
      const float4* src = reinterpret_cast<const float4*>(a_lds + s_local * 16);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 t = src[q4];
        a16[q4 * 4 + 0] = t.x;
        a16[q4 * 4 + 1] = t.y;
        a16[q4 * 4 + 2] = t.z;
        a16[q4 * 4 + 3] = t.w;
      }
    }
    // number of source views that see this sample = sum of the visibility masks, the last V
    // entries of the conditioning vector (cond_nerf.py:79-80)
    float n_valid = 0.0f;
    {
      const float* crow = cond + gs * CS + (D.cond_dim - D.n_views);
      for (int v = 0; v < D.n_views; ++v) n_valid += crow[v];
    }
    const bool q_valid = n_valid > 1.0f;
    __syncthreads();

    float qv[8], ov[8];
    {
      float kk[8], vv[8];
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) {
        const int row = 8 * hl + idx;
        const float4* wq = reinterpret_cast<const float4*>(sm_lds + row * 16);
        const float4* wk = reinterpret_cast<const float4*>(sm_lds + 256 + row * 16);
        const float4* wv = reinterpret_cast<const float4*>(sm_lds + 512 + row * 16);
        float sq = 0.f, sk = 0.f, sv = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 a = wq[q4], b = wk[q4], c = wv[q4];
          sq += a.x * a16[q4 * 4] + a.y * a16[q4 * 4 + 1] + a.z * a16[q4 * 4 + 2] + a.w * a16[q4 * 4 + 3];
          sk += b.x * a16[q4 * 4] + b.y * a16[q4 * 4 + 1] + b.z * a16[q4 * 4 + 2] + b.w * a16[q4 * 4 + 3];
          sv += c.x * a16[q4 * 4] + c.y * a16[q4 * 4 + 1] + c.z * a16[q4 * 4 + 2] + c.w * a16[q4 * 4 + 3];
        }
        qv[idx] = q_valid ? sq * 0.5f : 0.0f;  // temperature sqrt(d_k) = 2; masked query row -> uniform
        kk[idx] = sk;
        vv[idx] = sv;
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float4* dst = reinterpret_cast<float4*>(kv_lds + ((size_t)(ray_t * 4 + 2 * hl + hh) * Sp + jp) * 8);
        dst[0] = make_float4(kk[hh * 4], kk[hh * 4 + 1], kk[hh * 4 + 2], kk[hh * 4 + 3]);
        dst[1] = make_float4(vv[hh * 4], vv[hh * 4 + 1], vv[hh * 4 + 2], vv[hh * 4 + 3]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const float4* base = reinterpret_cast<const float4*>(kv_lds + (size_t)(ray_t * 4 + 2 * hl + hh) * Sp * 8);
      const float q0 = qv[hh * 4], q1 = qv[hh * 4 + 1], q2 = qv[hh * 4 + 2], q3 = qv[hh * 4 + 3];
      float mx = -3.0e38f;
      for (int jj = 0; jj < S; ++jj) {
        const float4 k4 = base[jj * 2];
        mx = fmaxf(mx, q0 * k4.x + q1 * k4.y + q2 * k4.z + q3 * k4.w);
      }
      float l = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
      for (int jj = 0; jj < S; ++jj) {
        const float4 k4 = base[jj * 2], v4 = base[jj * 2 + 1];
        const float p = expf((q0 * k4.x + q1 * k4.y + q2 * k4.z + q3 * k4.w) - mx);
        l += p;
        o0 += p * v4.x;
        o1 += p * v4.y;
        o2 += p * v4.z;
        o3 += p * v4.w;
      }
      const float il = 1.0f / l;
      ov[hh * 4] = o0 * il;
      ov[hh * 4 + 1] = o1 * il;
      ov[hh * 4 + 2] = o2 * il;
      ov[hh * 4 + 3] = o3 * il;
    }
    // fc (16x16, no bias) split over the two half-waves, + residual, LayerNorm(eps 1e-6)
    float xr[16];
    {
      float mean = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float4* wf = reinterpret_cast<const float4*>(sm_lds + 768 + c * 16 + 8 * hl);
        const float4 w0 = wf[0], w1 = wf[1];
        float part = w0.x * ov[0] + w0.y * ov[1] + w0.z * ov[2] + w0.w * ov[3] + w1.x * ov[4] +
                     w1.y * ov[5] + w1.z * ov[6] + w1.w * ov[7];
        part += __shfl_xor(part, 32, 64);
        xr[c] = part + a16[c];
        mean += xr[c];
      }
      mean *= (1.0f / 16.0f);
      float var = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float dlt = xr[c] - mean;
        var += dlt * dlt;
      }
      const float rstd = 1.0f / sqrtf(var * (1.0f / 16.0f) + 1e-6f);
#pragma unroll
      for (int c = 0; c < 16; ++c) xr[c] = (xr[c] - mean) * rstd * sm_lds[1024 + c] + sm_lds[1040 + c];
    }
    // out_alpha_linear: 16 -> 16 (act) -> 1 (ReLU)   (cond_nerf.py:33-36, 84)
    float sigma = sm_lds[1344];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4* w1 = reinterpret_cast<const float4*>(sm_lds + 1056 + c * 16);
      float t = sm_lds[1312 + c];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 w = w1[q4];
        t += w.x * xr[q4 * 4] + w.y * xr[q4 * 4 + 1] + w.z * xr[q4 * 4 + 2] + w.w * xr[q4 * 4 + 3];
      }
      t = D.raytrans_elu ? (t > 0.0f ? t : (expf(t) - 1.0f)) : fmaxf(t, 0.0f);
      sigma += sm_lds[1328 + c] * t;
    }
    sigma = fmaxf(sigma, 0.0f);
    if (D.density_maskfill && n_valid < 1.0f) sigma = 0.0f;
    if (hl == 0) {
      rs_lds[s_local * 4 + 3] = sigma;
      if (dbg_sigma && ray_ok && jp < S) dbg_sigma[gs] = sigma;
    }
    __syncthreads();

    // ============================================================ compositing (K5)
    for (int rt = wave; rt < rays_per_tile; rt += NW) {
      const int rr = tile * rays_per_tile + rt;
      if (rr >= R.n_rays) continue;  // wave-uniform
      float rlen = 1.0f;
      if (!D.wo_render_interval) {
        const RayGeom gg = make_ray(R, rr);
        rlen = sqrtf(gg.rx * gg.rx + gg.ry * gg.ry + gg.rz * gg.rz);
      }
      float carry = 0.f, ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, ao = 0.f;
      for (int j0 = 0; j0 < S; j0 += 64) {
        const int jj = j0 + lane;
        const bool ok = jj < S;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        float dd = 0.f;
        if (ok) {
          c = reinterpret_cast<const float4*>(rs_lds)[rt * Sp + jj];
          dd = sample_depth(R, rr, jj);
          if (!D.wo_render_interval) {
            const float intv = (jj + 1 < S) ? (sample_depth(R, rr, jj + 1) - dd) : 1e10f;
            c.w = c.w * (intv * rlen);
          }
        }
        // exclusive prefix of sigma*delta: scan the lane-shifted values (see composite.hip)
        float incl = __shfl_up(c.w, 1, 64);
        if (lane == 0) incl = 0.0f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const float t = __shfl_up(incl, off, 64);
          if (lane >= off) incl += t;
        }
        const float excl = carry + incl;
        const float w = ok ? expf(-excl) * (1.0f - expf(-c.w)) : 0.0f;
        ar += w * c.x;
        ag += w * c.y;
        ab += w * c.z;
        ad += w * dd;
        ao += w;
        carry = __shfl(excl + c.w, 63, 64);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        ar += __shfl_xor(ar, off, 64);
        ag += __shfl_xor(ag, off, 64);
        ab += __shfl_xor(ab, off, 64);
        ad += __shfl_xor(ad, off, 64);
        ao += __shfl_xor(ao, off, 64);
      }
      if (lane == 0) {
        const float bg = D.setbg_opaque ? (1.0f - ao) : 0.0f;
        out_rgb[(size_t)rr * 3 + 0] = ar + bg;
        out_rgb[(size_t)rr * 3 + 1] = ag + bg;
        out_rgb[(size_t)rr * 3 + 2] = ab + bg;
        out_depth[rr] = ad;
        out_opacity[rr] = ao;
      }
    }
    __syncthreads();  // LDS is recycled by the next tile
  }
}

// ------------------------------------------------------------------ host side
// Segment schedule shared with the Python packer (matchnerf_amd/cond_nerf.py):
// stages (steps, M-blocks) in consumption order; a stage is cut into ceil(T/cap) segments,
// the first ones get floor(T/nseg) steps, the last one the rest; every segment is padded to
// a multiple of 256 floats.
static int build_schedule(const mnerf_decoder* D, DecSched* sch) {
  const int fs = D->cond_stride / 2, es = 3 * D->L_3D + 2;
  const int T[12] = {fs, es, 65, 65, 65, 65, es, 64, 65, 65, 66, 33};
  const int M[12] = {4, 4, 4, 4, 4, 4, 4, 4, 1, 4, 2, 1};
  int n = 0;
  long long off = 0;
  for (int st = 0; st < 12; ++st) {
    const int cap = SEG_CAP_FLOATS / (64 * M[st]);
    const int nseg = (T[st] + cap - 1) / cap;
    const int base = T[st] / nseg;
    for (int k = 0; k < nseg; ++k) {
      if (n >= MAX_SEGS) return -1;
      const int steps = (k == nseg - 1) ? (T[st] - base * (nseg - 1)) : base;
      const int fl = ((steps * 64 * M[st] + 255) / 256) * 256;
      if (fl > SEG_CAP_FLOATS) return -1;
      sch->seg_off[n] = (int)off;
      sch->seg_floats[n] = fl;
      sch->seg_steps[n] = steps;
      off += fl;
      ++n;
    }
  }
  sch->n_seg = n;
  sch->film_steps = fs;
  sch->enc_steps = es;
  return (int)off;
}

extern "C" int64_t mnerf_decoder_wstream_floats(int32_t cond_stride, int32_t L_3D) {
  mnerf_decoder d = {};
  d.cond_stride = cond_stride;
  d.L_3D = L_3D;
  DecSched s;
  return build_schedule(&d, &s);
}

static int pick_padded_samples(int S) {
  if (S <= 32) return 32;
  if (S <= 64) return 64;
  if (S <= 128) return 128;
  if (S <= 256) return 256;
  return -1;
}

extern "C" int mnerf_decoder_chunk(const mnerf_decoder* dec, const mnerf_view* view0,
                                   const mnerf_rays* rays, const float* cond, float* rgb,
                                   float* depth, float* opacity, float* dbg_rgb_s,
                                   float* dbg_sigma, void* stream) {
  MNERF_REQUIRE(dec && view0 && rays, MNERF_E_NULL, "mnerf_decoder_chunk: NULL argument struct");
  MNERF_REQUIRE(dec->wstream && dec->small_ && cond && rgb && depth && opacity, MNERF_E_NULL,
                "mnerf_decoder_chunk: NULL buffer");
  MNERF_REQUIRE(mnerf_aligned16(dec->wstream) && mnerf_aligned16(cond), MNERF_E_ALIGN,
                "mnerf_decoder_chunk: wstream / cond must be 16-byte aligned");
  MNERF_REQUIRE(dec->L_3D >= 0 && dec->L_3D <= 16, MNERF_E_RANGE, "mnerf_decoder_chunk: L_3D=%d",
                dec->L_3D);
  MNERF_REQUIRE(dec->cond_stride % 8 == 0 && dec->cond_stride >= dec->cond_dim + 1 &&
                    dec->cond_stride <= 64,
                MNERF_E_RANGE,
                "mnerf_decoder_chunk: cond_stride=%d (cond_dim=%d) must be a multiple of 8 in (cond_dim, 64]",
                dec->cond_stride, dec->cond_dim);
  MNERF_REQUIRE(dec->n_views >= 1 && dec->n_views * 4 < dec->cond_dim, MNERF_E_RANGE,
                "mnerf_decoder_chunk: n_views=%d inconsistent with cond_dim=%d", dec->n_views,
                dec->cond_dim);
  MNERF_REQUIRE(rays->n_rays >= 0 && rays->n_samples >= 1, MNERF_E_RANGE,
                "mnerf_decoder_chunk: n_rays=%d S=%d", rays->n_rays, rays->n_samples);
  MNERF_REQUIRE(rays->legacy_coord == 0 || rays->n_samples >= 2, MNERF_E_RANGE,
                "mnerf_decoder_chunk: legacy depth sampling needs S >= 2");
  const int Sp = pick_padded_samples(rays->n_samples);
  MNERF_REQUIRE(Sp > 0, MNERF_E_UNSUPPORTED,
                "mnerf_decoder_chunk: sample_intvs=%d > 256 is not supported by the fused kernel",
                rays->n_samples);
  DecSched sch;
  const int total = build_schedule(dec, &sch);
  MNERF_REQUIRE(total > 0, MNERF_E_RANGE, "mnerf_decoder_chunk: cannot schedule weight stream");
  MNERF_REQUIRE(dec->wstream_floats == total, MNERF_E_RANGE,
                "mnerf_decoder_chunk: wstream has %lld floats, schedule expects %d",
                (long long)dec->wstream_floats, total);
  if (rays->n_rays == 0) return MNERF_OK;
  hipStream_t st = (hipStream_t)stream;
  if (Sp <= 128) {
    constexpr int NW = 4;
    const int rpt = (NW * 32) / Sp;
    int tiles = (rays->n_rays + rpt - 1) / rpt;
    const int grid = tiles < 2048 ? tiles : 2048;
    const size_t lds = Smem<NW>::TOTAL_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)decoder_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL(decoder_kernel<NW>, dim3(grid), dim3(NW * 64), lds, st, *dec, sch, *view0, *rays,
                       Sp, cond, rgb, depth, opacity, dbg_rgb_s, dbg_sigma);
  } else {
    constexpr int NW = 8;
    int tiles = rays->n_rays;
    const int grid = tiles < 1024 ? tiles : 1024;
    const size_t lds = Smem<NW>::TOTAL_FLOATS * sizeof(float);
    static bool attr_set8 = false;
    if (!attr_set8) {
      (void)hipFuncSetAttribute((const void*)decoder_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set8 = true;
    }
    hipLaunchKernelGGL(decoder_kernel<NW>, dim3(grid), dim3(NW * 64), lds, st, *dec, sch, *view0, *rays,
                       Sp, cond, rgb, depth, opacity, dbg_rgb_s, dbg_sigma);
  }
  return mnerf_check_launch("mnerf_decoder_chunk");
}
