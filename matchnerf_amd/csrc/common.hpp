// Shared host/device helpers for libmnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/mnerf.h"

// ------------------------------------------------------------------ host: error channel
void mnerf_set_error(const char* fmt, ...);

#define MNERF_REQUIRE(cond, code, ...)   \
  do {                                   \
    if (!(cond)) {                       \
      mnerf_set_error(__VA_ARGS__);      \
      return (code);                     \
    }                                    \
  } while (0)

static inline int mnerf_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mnerf_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return MNERF_OK;
}

static inline bool mnerf_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ------------------------------------------------------------------ host: tuning table, per-device set-up
// Debug / tuning knobs (MNERF_DECODER_GRID, MNERF_DECODER_STAGGER[_MODE], MNERF_CV_VARIANT, MNERF_CV_GRID,
// MNERF_WA_MIN4): read from the environment once at library load (api.cpp), constant afterwards.
struct mnerf_tuning {
  int decoder_grid, decoder_stagger, decoder_stagger_mode;
  int cv_variant, cv_grid;
  int cv_pair_block;  // MNERF_CV_PAIR_BLOCK: view pairs per launch of the many-view cost volume (0: all in one; default -1: 8 for the walk, all in one for the matrix form)
  int cv_mm;      // MNERF_CV_MM (default 1): the matrix form of the cost volume (cost_volume_mm.hip) where it applies (scene->feat_op given, contiguous pixels)
  int cv_mm_spw;  // MNERF_CV_MM_SPW (default 4): samples per wave and work item of the matrix form (an item = one 8x4-pixel tile x 4 x spw samples)
  int cv_uvpair;  // MNERF_CV_UVPAIR (default -1 = by LDS footprint): 1 / 0 force the per-pair projection scratch of the walk on / off
  int wa_min4;
  int wa_xcd;  // pre-split window attention: all query blocks of a window on one XCD (1) or launch order (0)
  int render_fused;  // MNERF_RENDER_FUSED (default 0): 1 = mnerf_render_chunk uses the one-launch form where it applies
  int decoder_pp;       // MNERF_DECODER_PP (default 1): the ping-pong form of the split-fp16 decoder where it applies
  int decoder_pp_grid;  // MNERF_DECODER_PP_GRID (default 256): its persistent grid, one 8-wave workgroup per CU
  int decoder_pp_max_s; // MNERF_DECODER_PP_MAX_S (default 256, values above 256 are clamped): largest padded sample count per ray that takes the ping-pong form
};
mnerf_tuning mnerf_tune();  // a snapshot (api.cpp: copied under the table lock)
// true exactly once per (mask, current HIP device): guards hipFuncSetAttribute, which is per device
bool mnerf_once_per_device(std::atomic<unsigned long long>& mask);
// argument checks of the scene / rays structs (cost_volume.hip)
int mnerf_scene_check(const mnerf_scene* sc, const mnerf_rays* rays, const char* who);
// fused ray-chunk form (decoder.hip), used by mnerf_render_chunk (render_chunk.hip)
bool mnerf_fused_render_applies(const mnerf_scene* sc, const mnerf_decoder* dec, const mnerf_rays* rays);
bool mnerf_cost_volume_takes_pose_table(const mnerf_scene* sc);                 // cost_volume.hip
bool mnerf_cost_volume_mm_applies(const mnerf_scene* sc, const mnerf_rays* rays);  // cost_volume_mm.hip
int mnerf_cost_volume_mm_launch(const mnerf_scene* sc, const mnerf_rays* rays, int cond_stride, float* cond, void* stream);
bool mnerf_decoder_takes_pose_table(const mnerf_decoder* dec, int n_samples);   // decoder.hip
int mnerf_fused_render_launch(const mnerf_scene* sc, const mnerf_decoder* dec, const mnerf_rays* rays, float* rgb,
                              float* depth, float* opacity, void* stream);

// ------------------------------------------------------------------ device: geometry
// The positional encoding multiplies the projected coordinate by up to 2^9 (x pi), so a
// 1-ulp difference in (u,v,z) becomes a 1e-4-class difference in sin/cos and, through random
// MLP weights, in RGB.  The reference's small matmuls ([N,4]@[4,3], [N,3]@[3,3]) evaluate on
// the CPU as a k-ordered FMA chain  acc = x0*w0; acc = fma(xk, wk, acc)  and its elementwise
// expressions round every op separately; tools/gen_golden.py fixtures confirm that this order
// reproduces the reference's `pts` and NDC coordinates BIT-FOR-BIT.  So the device code pins
// exactly that order: contraction is switched off in this block and every FMA is explicit.
#pragma clang fp contract(off)

__device__ __forceinline__ float dot3_chain(float x0, float x1, float x2, const float* w) {
  return __builtin_fmaf(x2, w[2], __builtin_fmaf(x1, w[1], x0 * w[0]));
}
// [x0 x1 x2 1] . w[0..3]
__device__ __forceinline__ float dot4h_chain(float x0, float x1, float x2, const float* w) {
  return __builtin_fmaf(x2, w[2], __builtin_fmaf(x1, w[1], x0 * w[0])) + w[3];
}

// One target ray rebuilt from its pixel index (misc/camera.py:255-278).
struct RayGeom {
  float cx, cy, cz;  // centre (camera position in world)
  float rx, ry, rz;  // un-normalised direction
};

// Pose table (mnerf_rays.pose_table): the per-launch target-camera constants of `Rt` replaced by those of pose `pose` = the one
// the launch-local ray `first_ray` belongs to.  Callers pass a wave-uniform ray (a tile's / a ray block's first one:
// rays_per_pose is a multiple of 64), so the 23 floats are scalar loads (from the scalar cache after a tile's first use).  The
// address is opaque to the compiler: a kernel that needs the constants at several places of a long tile loads them again at
// each place instead of keeping 23 more scalars alive across the tile.
__device__ __forceinline__ int pose_of_ray(const mnerf_rays& R, int first_ray) {
  return __builtin_amdgcn_readfirstlane((R.ray_begin + first_ray) / R.rays_per_pose);
}
__device__ __forceinline__ void rays_for_pose(mnerf_rays& Rt, const mnerf_rays& R, int pose) {
  const float* p = R.pose_table + (size_t)pose * MNERF_POSE_FLOATS;
  asm volatile("" : "+s"(p));
#pragma unroll
  for (int i = 0; i < 9; ++i) Rt.kinv[i] = p[i];
#pragma unroll
  for (int i = 0; i < 12; ++i) Rt.c2w[i] = p[9 + i];
  Rt.near_ = p[21];
  Rt.far_ = p[22];
  Rt.ray_begin = R.ray_begin - pose * R.rays_per_pose;  // make_ray: pixel = ray_begin + local ray = the pixel inside this pose's frame
}

__device__ __forceinline__ RayGeom make_ray(const mnerf_rays& R, int ray_local) {
  int pix = R.ray_idx ? R.ray_idx[ray_local] : (R.ray_begin + ray_local);
  int py = pix / R.width;
  int px = pix - py * R.width;
  float off = R.legacy_coord ? 0.0f : 0.5f;
  float x = (float)px + off, y = (float)py + off;
  // cam = [x y 1] @ Kinv^T   (img2cam, camera.py:221-222)
  float c0 = __builtin_fmaf(y, R.kinv[1], x * R.kinv[0]) + R.kinv[2];
  float c1 = __builtin_fmaf(y, R.kinv[4], x * R.kinv[3]) + R.kinv[5];
  float c2 = __builtin_fmaf(y, R.kinv[7], x * R.kinv[6]) + R.kinv[8];
  RayGeom g;
  g.cx = R.c2w[3];
  g.cy = R.c2w[7];
  g.cz = R.c2w[11];
  // ray = ([cam 1] @ c2w^T) - centre   (camera.py:273-276)
  g.rx = dot4h_chain(c0, c1, c2, R.c2w + 0) - g.cx;
  g.ry = dot4h_chain(c0, c1, c2, R.c2w + 4) - g.cy;
  g.rz = dot4h_chain(c0, c1, c2, R.c2w + 8) - g.cz;
  return g;
}

// depth of sample j on a ray (models/matchnerf.py:163-181); every op rounded on its own
__device__ __forceinline__ float sample_depth(const mnerf_rays& R, int ray_local, int j) {
  float shift = R.legacy_coord ? 0.0f : 0.5f;
  if (R.strat_u) shift = R.strat_u[(size_t)ray_local * R.n_samples + j];
  float denom = R.legacy_coord ? (float)(R.n_samples - 1) : (float)R.n_samples;
  float t = ((float)j + shift) / denom;
  float span = R.far_ - R.near_;
  float d = t * span;
  d = d + R.near_;
  if (R.depth_inverse) d = 1.0f / (d + 1e-8f);
  return d;
}

// x = c + d v  (camera.py:281-286): multiply and add are separate torch ops
__device__ __forceinline__ void ray_point(const RayGeom& g, float d, float& px, float& py, float& pz) {
  float tx = g.rx * d, ty = g.ry * d, tz = g.rz * d;
  px = g.cx + tx;
  py = g.cy + ty;
  pz = g.cz + tz;
}

// world point -> (u, v, z) of a source view (misc/camera.py:351-379)
__device__ __forceinline__ void project(const mnerf_view& V, float px, float py, float pz,
                                        float wm1, float hm1, float& u, float& v,
                                        float& z) {
  float c0 = dot4h_chain(px, py, pz, V.extr + 0);
  float c1 = dot4h_chain(px, py, pz, V.extr + 4);
  float c2 = dot4h_chain(px, py, pz, V.extr + 8);
  float q0 = dot3_chain(c0, c1, c2, V.intr + 0);
  float q1 = dot3_chain(c0, c1, c2, V.intr + 3);
  float q2 = dot3_chain(c0, c1, c2, V.intr + 6);
  u = q0 / q2;
  u = u / wm1;
  v = q1 / q2;
  v = v / hm1;
  float span = V.far_ - V.near_;
  z = (q2 - V.near_) / span;
}

// sin(arg) (quarter = 0) or cos(arg) (quarter = 1) for the positional encoding, |arg| up to a
// few thousand.  The argument is reduced in revolutions with a two-float 1/(2 pi) (product error
// term recovered by FMA — which is why contraction must stay OFF here: fusing `th - rint(th)`
// would count that term twice), cos is sin shifted by a quarter turn, and sin(2 pi r) on
// [-1/4, 1/4] is a degree-9 odd minimax polynomial.  Max abs error 3.0e-7 over |arg| <= 2000
// (tools/exp/sincos_poly.hip, measured on MI355X) versus 0.7e-7 for ocml sincosf at ~8x the
// instruction count; v_sin_f32 / v_cos_f32 alone are only good to 8e-5.
// (th, tl) = arg / (2 pi) as an unevaluated two-float sum
__device__ __forceinline__ void turns_two_float(float arg, float& th, float& tl) {
  const float C_HI = 0.15915494309189535f;
  const float C_LO = (float)(0.15915494309189535 - (double)0.15915494309189535f);
  th = arg * C_HI;
  tl = __builtin_fmaf(arg, C_LO, __builtin_fmaf(arg, C_HI, -th));
}
// sin (quarter = 0) / cos (quarter = 1) of 2 pi (th + tl).  Scaling (th, tl) by a power of two is exact, so the
// 2^l octaves of the positional encoding share ONE two-float reduction per coordinate: bit-identical to calling
// sin_quarter(2^l arg, .) for every octave, at two multiplications instead of a multiplication and three FMAs each.
__device__ __forceinline__ float sin_quarter_turns(float th, float tl, int quarter) {
  float r = th - rintf(th);
  r = r + (tl + (quarter ? 0.25f : 0.0f));
  r = r - rintf(r);
  float a = fabsf(r);
  r = (a > 0.25f) ? (copysignf(0.5f, r) - r) : r;
  float x2 = r * r;
  float p = 39.536705017089844f;
  p = __builtin_fmaf(p, x2, -76.5497817993164f);
  p = __builtin_fmaf(p, x2, 81.60100555419922f);
  p = __builtin_fmaf(p, x2, -41.34165573120117f);
  p = __builtin_fmaf(p, x2, 6.283185005187988f);
  return p * r;
}
__device__ __forceinline__ float sin_quarter(float arg, int quarter) {
  float th, tl;
  turns_two_float(arg, th, tl);
  return sin_quarter_turns(th, tl, quarter);
}
#pragma clang fp contract(fast)

// ---- "largest magnitude of a tensor" hand-over between a producer kernel and the split-fp16 convolution that reads
// the tensor (conv.hip): a region of MNERF_ABSMAX_SLOTS partial maxima, one 128-byte line apart, zeroed by the caller.
// Workgroup b of the producer merges its maximum into slot b % SLOTS with an atomic maximum (non-negative floats order
// as integers; one address would serialise thousands of atomics: 15-40 us per kernel measured), the consumer takes the
// maximum of the slots.
#ifdef __HIPCC__
__device__ __forceinline__ void mnerf_absmax_merge(float wave_or_block_max, float* region) {
  atomicMax(reinterpret_cast<int*>(region + (blockIdx.x % MNERF_ABSMAX_SLOTS) * MNERF_ABSMAX_STRIDE),
            __float_as_int(wave_or_block_max));
}
__device__ __forceinline__ float mnerf_absmax_read(const float* region) {  // whole wave; every lane gets the maximum
  float m = region[(threadIdx.x & (MNERF_ABSMAX_SLOTS - 1)) * MNERF_ABSMAX_STRIDE];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  return m;
}
#endif
