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
#include <stdlib.h>

#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SEG_CAP_FLOATS (33 * 256)  // one LDS weight buffer: 33 KiB
#define MAX_SEGS 40
#define SMALL_FIXED 1360  // floats of `small` before the optional ray-posenc table

#ifdef MNERF_TIMELINE
// debug build (tools/exp/timeline.py): per-wave s_memtime stamps at phase boundaries
#define TL_POINTS 20
#define TL_STAMP(k)                                                                     \
  do {                                                                                  \
    if (sch.tl && lane == 0 && tl_slot >= 0 && tl_tile < 4)                             \
      sch.tl[(((size_t)tl_slot * 4 + tl_tile) * NW + wave) * TL_POINTS + (k)] =         \
          __builtin_amdgcn_s_memtime();                                                 \
  } while (0)
#else
#define TL_STAMP(k) do {} while (0)
#endif

struct DecSched {
#ifdef MNERF_TIMELINE
  unsigned long long* tl;
#endif
  int stagger_sleeps;  // one-time start delay (x s_sleep 127) of the 2nd resident workgroup of a CU
  int stagger_mode;    // which workgroups wait: 0 odd HW wave slot, 1 upper half of grid, 2 (b>>3)&1, 3 all
  int n_seg;
  int film_steps, enc_steps;
  int seg_off[MAX_SEGS];     // float offset of the segment in wstream (multiple of 256)
  int seg_floats[MAX_SEGS];  // padded to a multiple of 256 floats (1 KiB DMA pieces)
  int seg_steps[MAX_SEGS];
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- weight-segment DMA: each wave copies 1 KiB pieces, LDS dest = uniform base + lane*16.
// The LDS-DMA is issued from inline asm on purpose: hipcc tracks a builtin global_load_lds as a
// pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the NEXT ds_read — which would
// drain the prefetch of segment i+1 before segment i has issued a single MFMA (seen in the
// ISA: the DMA was fully exposed 18x per tile).  An asm load is invisible to that bookkeeping
// (cdna_hip_programming.md 5.7); segment_wait() below is the one place that waits for it,
// right before the barrier that publishes the buffer.  M0 (LDS base) is written and restored
// inside the same statement.
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte_addr)
      : "memory");
}

__device__ __forceinline__ void segment_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int NW>
__device__ __forceinline__ void prefetch_segment(const float* __restrict__ wstream,
                                                 const DecSched& sch, int seg, float* lds_buf,
                                                 int wave, int lane) {
  if (seg >= sch.n_seg) return;
  const float* src = wstream + sch.seg_off[seg] + lane * 4;
  const int pieces = sch.seg_floats[seg] >> 8;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_buf;
  for (int p = wave; p < pieces; p += NW)
    glds16(src + p * 256, __builtin_amdgcn_readfirstlane(base + (unsigned)p * 1024u));
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
  // A fragments are double-buffered in registers: the ds_read_b128 of step t+1 is issued before
  // the four MFMAs of step t, so its LDS latency hides under 256 cycles of matrix work.
  const float4* a4 = reinterpret_cast<const float4*>(seg) + step0 * 64 + lane;
  float4 cur = a4[0];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const float4 nxt = a4[(r + 1) * 64];  // r == 31 reads the following fragment (bias step or pad)
    __builtin_amdgcn_sched_barrier(0);    // keep the prefetch ABOVE this step's MFMAs
    const float b = r < 16 ? h0[r & 15] : h1[r & 15];
    acc[0] = mfma(cur.x, b, acc[0]);
    acc[1] = mfma(cur.y, b, acc[1]);
    acc[2] = mfma(cur.z, b, acc[2]);
    acc[3] = mfma(cur.w, b, acc[3]);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
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
    return sin_quarter(arg, hl);
  }
  if (t == L3) return hl ? y : x;
  return hl ? 1.0f : z;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16 independent 4x4 outer products per wave: D[r](lane l) += A(lane 4*(l/4)+r) * B(lane l)
// (layout verified on MI355X, tools/exp/mfma4x4.hip)
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

template <int NW, int SP>
struct Smem {
  static constexpr int TILE = NW * 32;
  static constexpr int W_FLOATS = 2 * SEG_CAP_FLOATS;
  static constexpr int A_FLOATS = TILE * 16;
  static constexpr int RS_FLOATS = TILE * 4;
  // Ray-attention scratch lives on top of the (by then idle) weight double buffer.
  //   MFMA form (SP <= 128): K [rays][4][SP][4], V^T [rays][4][4][SP], Q [TILE][16], O [TILE][16]
  //   VALU form (SP  = 256): K|V interleaved [rays][4][SP][8]
  // followed by a copy of the small parameter block.
  static constexpr bool MFMA_ATT = SP <= 128;
  static constexpr int KV_FLOATS = MFMA_ATT ? TILE * 64 : TILE * 32;
  static constexpr int ATT_FLOATS = KV_FLOATS + SMALL_FIXED;
  static constexpr bool ATT_ALIASED = ATT_FLOATS <= W_FLOATS;
  static constexpr int TOTAL_FLOATS = W_FLOATS + A_FLOATS + RS_FLOATS + (ATT_ALIASED ? 0 : ATT_FLOATS);
};

template <int NW, int SP>
__global__ __launch_bounds__(NW * 64, 2) void decoder_kernel(
    mnerf_decoder D, DecSched sch, mnerf_view view0, mnerf_rays R,
    const float* __restrict__ cond, float* __restrict__ out_rgb, float* __restrict__ out_depth,
    float* __restrict__ out_opacity, float* __restrict__ dbg_rgb_s, float* __restrict__ dbg_sigma) {
  constexpr int Sp = SP;
  using SM = Smem<NW, SP>;
  constexpr int TILE = SM::TILE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wbuf0 = smem;
  float* wbuf1 = smem + SEG_CAP_FLOATS;
  float* a_lds = smem + SM::W_FLOATS;                 // [TILE][16]  alpha features
  float* rs_lds = a_lds + SM::A_FLOATS;               // [TILE][4]   rgb.xyz, sigma.w
  float* att = SM::ATT_ALIASED ? smem : (rs_lds + SM::RS_FLOATS);
  float* kv_lds = att;                                // VALU form: [rays][4 heads][Sp][8] (k0..3, v0..3)
  float* k_lds = att;                                 // MFMA form
  float* vt_lds = att + TILE * 16;
  float* q_lds = att + TILE * 32;
  float* o_lds = att + TILE * 48;
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

  // Phase stagger.  With <=256 VGPRs and 76 KiB of LDS two workgroups share a CU (one wave of
  // each per SIMD).  Launched together they run IN PHASE: their VALU-only phases (prologue,
  // ray attention, compositing) coincide and the matrix pipe idles for both (measured: MFMA busy
  // 69.6 % = 2M/(2M+V)).  The grid is persistent (2 workgroups per CU), and the workgroup whose
  // waves sit in the odd hardware wave slot waits half a tile ONCE, so that from then on one
  // workgroup's VALU phases overlap the other's MFMA phases.  Speed only: any placement is correct.
  if (sch.stagger_sleeps > 0) {
    const unsigned hw_id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11));  // HW_ID.WAVE_ID
    bool late = (hw_id & 1u) != 0;
    if (sch.stagger_mode == 1) late = blockIdx.x >= (gridDim.x >> 1);
    if (sch.stagger_mode == 2) late = ((blockIdx.x >> 3) & 1) != 0;
    if (sch.stagger_mode == 3) late = true;
    if (late)
      for (int i = 0; i < sch.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  if (sch.stagger_mode >= 10) {  // experiment: static priority split between the two co-resident WGs
    const unsigned hw_id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11));
    const bool slot_odd = __builtin_amdgcn_readfirstlane(hw_id & 1u) != 0;
    if (sch.stagger_mode == 10) { if (slot_odd) __builtin_amdgcn_s_setprio(3); }
    if (sch.stagger_mode == 11) { if (!slot_odd) __builtin_amdgcn_s_setprio(3); }
    if (sch.stagger_mode == 12) { if (slot_odd) { __builtin_amdgcn_s_setprio(3); for (int i = 0; i < 16; ++i) __builtin_amdgcn_s_sleep(127); } }
  }

#ifdef MNERF_TIMELINE
  int tl_tile = -1;
  // record blocks [0,64) and their presumed CU partners [256,320)
  const int tl_slot = blockIdx.x < 64 ? (int)blockIdx.x : ((blockIdx.x >= 256 && blockIdx.x < 320) ? (int)blockIdx.x - 192 : -1);
#endif
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#ifdef MNERF_TIMELINE
    ++tl_tile;
    if (sch.tl && lane == 0 && tl_slot >= 0 && tl_tile < 4)
      sch.tl[(((size_t)tl_slot * 4 + tl_tile) * NW + wave) * TL_POINTS + 19] =
          __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_ID
#endif
    TL_STAMP(0);
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

    // global loads of this tile issued first, so that their latency hides under the prologue:
    // this half-wave's FiLM inputs and the visibility-mask sum (cond_nerf.py:79-80)
    float4 cpre[8];
    {
      const float4* crow4 = reinterpret_cast<const float4*>(cond + gs * CS + (size_t)hl * sch.film_steps);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        cpre[i] = (4 * i < sch.film_steps) ? crow4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float n_valid = 0.0f;
    {
      const float* mrow = cond + gs * CS + (D.cond_dim - D.n_views);
      for (int v = 0; v < D.n_views; ++v) n_valid += mrow[v];
    }
    const bool q_valid = n_valid > 1.0f;

    int seg = 0;   // running segment index; segment k lives in buffer (k & 1)
    prefetch_segment<NW>(D.wstream, sch, 0, wbuf0, wave, lane);
    segment_wait();
    __syncthreads();

#define CUR_BUF ((seg & 1) ? wbuf1 : wbuf0)
#define NXT_BUF ((seg & 1) ? wbuf0 : wbuf1)
#define SEG_BEGIN() prefetch_segment<NW>(D.wstream, sch, seg + 1, NXT_BUF, wave, lane)
#define SEG_END()     \
  do {                \
    segment_wait();   \
    __syncthreads();  \
    ++seg;            \
  } while (0)

    TL_STAMP(1);
    // ------------------------------------------------------------ FiLM = pts_bias(cond)
    f32x16 film[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) film[m] = (f32x16)(0.0f);
    {
      // cond_stride <= 64 => film_steps <= 32 => one segment; inputs preloaded in cpre[]
      const int ns = sch.seg_steps[seg];
      SEG_BEGIN();
      const float* wseg = CUR_BUF;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (4 * i < ns) {
          step4(film, wseg, 4 * i + 0, lane, cpre[i].x);
          step4(film, wseg, 4 * i + 1, lane, cpre[i].y);
          step4(film, wseg, 4 * i + 2, lane, cpre[i].z);
          step4(film, wseg, 4 * i + 3, lane, cpre[i].w);
        }
      }
      SEG_END();
    }

    TL_STAMP(2);
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

    TL_STAMP(3);
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

    TL_STAMP(4);
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

    TL_STAMP(5);
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

    TL_STAMP(6);
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

    TL_STAMP(7);
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

    TL_STAMP(8);
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

    TL_STAMP(9);
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
    __syncthreads();

    float ov16[16];  // attention output of this lane's sample, all 4 heads (head-major)
    if constexpr (SM::MFMA_ATT) {
      // ---- q/k/v projections (16 -> 16 each) of this lane's sample for its two heads
      {
        float qq[8], kk[8], vv[8];
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
          qq[idx] = q_valid ? sq * 0.5f : 0.0f;  // temperature sqrt(d_k) = 2; masked query row -> uniform
          kk[idx] = sk;
          vv[idx] = sv;
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int head = 2 * hl + hh;
          *reinterpret_cast<float4*>(k_lds + ((ray_t * 4 + head) * Sp + jp) * 4) =
              make_float4(kk[hh * 4], kk[hh * 4 + 1], kk[hh * 4 + 2], kk[hh * 4 + 3]);
          float* vcol = vt_lds + (ray_t * 4 + head) * 4 * Sp + jp;
          vcol[0] = vv[hh * 4];
          vcol[Sp] = vv[hh * 4 + 1];
          vcol[2 * Sp] = vv[hh * 4 + 2];
          vcol[3 * Sp] = vv[hh * 4 + 3];
          *reinterpret_cast<float4*>(q_lds + s_local * 16 + head * 4) =
              make_float4(qq[hh * 4], qq[hh * 4 + 1], qq[hh * 4 + 2], qq[hh * 4 + 3]);
        }
      }
      __syncthreads();
      TL_STAMP(10);
      // ---- attention proper on the matrix pipe: lane = query.  v_mfma_f32_4x4x1_16b runs 16
      // independent 4x4 outer products per instruction, D[r](lane) += A(lane 4*(l/4)+r) B(lane):
      //   scores of 4 keys  s4[r] += K[k0+r][d] * Q[query][d]     (A = K row l%4, 4 steps over d)
      //   output            o4[d] += V[key][d] * P[query][key]    (A = V^T row l%4, 1 step per key)
      // Scores / probabilities of all S keys stay in VGPRs (the MLP's registers are dead here).
      int a_ray, a_hp, a_jq;
      if constexpr (SP >= 64) {
        constexpr int CH = SP / 64;
        int idx = wave;
        const int chunk = idx % CH;
        idx /= CH;
        a_hp = idx & 1;
        a_ray = idx >> 1;
        a_jq = chunk * 64 + lane;
      } else {
        a_ray = wave * (64 / (2 * SP)) + 0;  // SP == 32: one ray per wave, head pairs in the half-waves
        a_hp = lane >> 5;
        a_jq = lane & 31;
      }
      const int s_q = a_ray * Sp + a_jq;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int head = 2 * a_hp + hh;
        const float4 q4 = *reinterpret_cast<const float4*>(q_lds + s_q * 16 + head * 4);
        const float* kb = k_lds + ((a_ray * 4 + head) * Sp + (lane & 3)) * 4;
        f32x4 sc[SP / 4];
#pragma unroll
        for (int g = 0; g < SP / 4; ++g) {
          const float4 kk = *reinterpret_cast<const float4*>(kb + g * 16);
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
          t = mfma4(kk.x, q4.x, t);
          t = mfma4(kk.y, q4.y, t);
          t = mfma4(kk.z, q4.z, t);
          t = mfma4(kk.w, q4.w, t);
          sc[g] = t;
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int g = 0; g < SP / 4; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = (4 * g + r < S) ? sc[g][r] : -3.0e38f;  // padded key slots
            sc[g][r] = v;
            mx = fmaxf(mx, v);
          }
        float lsum = 0.f;
#pragma unroll
        for (int g = 0; g < SP / 4; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __expf(sc[g][r] - mx);
            sc[g][r] = pr;
            lsum += pr;
          }
        const float* vb = vt_lds + ((a_ray * 4 + head) * 4 + (lane & 3)) * Sp;
        f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < SP / 4; g += 2) {
          const float4 va = *reinterpret_cast<const float4*>(vb + 4 * g);
          const float4 vc = *reinterpret_cast<const float4*>(vb + 4 * g + 4);
          oa = mfma4(va.x, sc[g][0], oa);
          ob = mfma4(vc.x, sc[g + 1][0], ob);
          oa = mfma4(va.y, sc[g][1], oa);
          ob = mfma4(vc.y, sc[g + 1][1], ob);
          oa = mfma4(va.z, sc[g][2], oa);
          ob = mfma4(vc.z, sc[g + 1][2], ob);
          oa = mfma4(va.w, sc[g][3], oa);
          ob = mfma4(vc.w, sc[g + 1][3], ob);
        }
        const float il = 1.0f / lsum;
        *reinterpret_cast<float4*>(o_lds + s_q * 16 + head * 4) =
            make_float4((oa[0] + ob[0]) * il, (oa[1] + ob[1]) * il, (oa[2] + ob[2]) * il, (oa[3] + ob[3]) * il);
      }
      __syncthreads();
      {
        const float4* src = reinterpret_cast<const float4*>(o_lds + s_local * 16);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 t = src[q4];
          ov16[q4 * 4 + 0] = t.x;
          ov16[q4 * 4 + 1] = t.y;
          ov16[q4 * 4 + 2] = t.z;
          ov16[q4 * 4 + 3] = t.w;
        }
      }
    } else {
      // ---- VALU form (S > 128): two lanes per sample, two heads each, K/V broadcast from LDS
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
          qv[idx] = q_valid ? sq * 0.5f : 0.0f;
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
      TL_STAMP(10);
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
          const float p = __expf((q0 * k4.x + q1 * k4.y + q2 * k4.z + q3 * k4.w) - mx);
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
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) {  // gather the partner half-wave's two heads
        const float other = __shfl_xor(ov[idx], 32, 64);
        ov16[idx] = hl ? other : ov[idx];
        ov16[8 + idx] = hl ? ov[idx] : other;
      }
    }
    TL_STAMP(11);
    // fc (16x16, no bias) + residual, LayerNorm(eps 1e-6)
    float xr[16];
    {
      float mean = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float4* wf = reinterpret_cast<const float4*>(sm_lds + 768 + c * 16);
        float part = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 w = wf[q4];
          part += w.x * ov16[q4 * 4] + w.y * ov16[q4 * 4 + 1] + w.z * ov16[q4 * 4 + 2] + w.w * ov16[q4 * 4 + 3];
        }
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

    TL_STAMP(12);
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
    TL_STAMP(13);
    // No barrier here: the next tile touches rs_lds / a_lds only after several segment barriers,
    // and every read of the attention scratch (aliased on the weight buffers that the next
    // tile's first DMA overwrites) completed before the barrier in front of the compositing.
    TL_STAMP(14);
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
#ifdef MNERF_TIMELINE
  sch->tl = nullptr;
  if (const char* e = getenv("MNERF_TIMELINE_PTR")) sch->tl = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  sch->stagger_sleeps = 16;  // ~130k cycles ~ half a tile (MNERF_DECODER_STAGGER overrides)
  if (const char* e = getenv("MNERF_DECODER_STAGGER")) sch->stagger_sleeps = atoi(e);
  sch->stagger_mode = 0;
  if (const char* e = getenv("MNERF_DECODER_STAGGER_MODE")) sch->stagger_mode = atoi(e);
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
  int resident = 512;  // persistent: 2 workgroups per CU x 256 CUs
  if (const char* e = getenv("MNERF_DECODER_GRID")) resident = atoi(e);
#define MNERF_LAUNCH_DECODER(NW_, SP_)                                                               \
  do {                                                                                               \
    const int rpt = (NW_ * 32) / SP_;                                                                \
    const int tiles = (rays->n_rays + rpt - 1) / rpt;                                                \
    const int grid = tiles < resident ? tiles : resident;                                            \
    const size_t lds = Smem<NW_, SP_>::TOTAL_FLOATS * sizeof(float);                                 \
    static bool attr_set = false;                                                                    \
    if (!attr_set) {                                                                                 \
      (void)hipFuncSetAttribute((const void*)decoder_kernel<NW_, SP_>,                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
      attr_set = true;                                                                               \
    }                                                                                                \
    hipLaunchKernelGGL((decoder_kernel<NW_, SP_>), dim3(grid), dim3(NW_ * 64), lds, st, *dec, sch,   \
                       *view0, *rays, cond, rgb, depth, opacity, dbg_rgb_s, dbg_sigma);              \
  } while (0)
  switch (Sp) {
    case 32: MNERF_LAUNCH_DECODER(4, 32); break;
    case 64: MNERF_LAUNCH_DECODER(4, 64); break;
    case 128: MNERF_LAUNCH_DECODER(4, 128); break;
    default: MNERF_LAUNCH_DECODER(8, 256); break;
  }
#undef MNERF_LAUNCH_DECODER
  return mnerf_check_launch("mnerf_decoder_chunk");
}
