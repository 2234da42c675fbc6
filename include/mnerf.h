/*
 * mnerf.h — C ABI of libmnerf_hip.so: the MI355X (gfx950) kernels of the MatchNeRF per-ray
 * rendering hot path.
 *
 * The reference (donydchen/matchnerf) is pure Python/PyTorch and has no FFI of its own
 * (SURVEY.md §8b): the seam is the Python class `MatchNeRF` (models/matchnerf.py:13-325).
 * This header is the boundary the build introduces *beneath* that class.  Each entry point
 * replaces one chain of eager PyTorch ops in the reference (file:line given per function).
 * The Python host (matchnerf_amd/hip.py) binds it with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C: raw device pointers, explicit sizes, no torch / C++ types in any signature;
 *   - all tensors are fp32, row-major, contiguous unless a stride is given;
 *   - the caller allocates every buffer (inputs, outputs, scratch); functions only enqueue
 *     work on `stream` (a hipStream_t passed as void*; NULL = default stream): no
 *     allocation, no synchronisation => safe inside hipGraph capture;
 *   - small camera matrices travel BY VALUE inside the argument structs (kernel arguments),
 *     so no host->device copy is hidden in a call;
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative
 *     MNERF_E_* argument-check code.  Nothing throws across the ABI.  The message of the
 *     last failure on the calling thread is available from mnerf_last_error();
 *   - launches go to the calling thread's CURRENT HIP device (hipSetDevice / torch.cuda.device):
 *     every pointer and the stream must belong to it;
 *   - process-wide state: none that a caller can observe.  Debug / tuning knobs (MNERF_DECODER_GRID,
 *     MNERF_DECODER_STAGGER[_MODE], MNERF_CV_VARIANT, MNERF_CV_GRID, MNERF_WA_MIN4, MNERF_WA_XCD, MNERF_RENDER_FUSED) are read from the
 *     environment ONCE, when the library is loaded; the library keeps one bit per (kernel, device) to
 *     remember that the kernel's dynamic-LDS attribute has been raised on that device.  The full list of knobs: INTEGRATION.md 4.
 *
 * ABI history (mnerf_abi_version(); struct sizes from mnerf_struct_size()):
 *   v6  mnerf_debug_set_knob returns the status and the previous value separately
 *   v7  mnerf_rays gained pose_table / rays_per_pose (several target poses per launch; mnerf_render_takes_pose_table).
 *       Added under v7 without a layout change: mnerf_debug_gemm, mnerf_window_attention_presplit_stats,
 *       mnerf_window_attention_backward_stats.
 *   v8  no layout change: mnerf_decoder.wstream_format accepts MNERF_WSTREAM_F16X1 (one-product fp16 fast mode); pose tables at
 *       sample_intvs <= 128 (was 64); mnerf_debug_set_knob serialised with the launches' reads.
 *   v9  mnerf_scene gained feat_op (appended: the offsets of the older fields are unchanged, sizeof grows by 8): the split-fp16
 *       OPERAND IMAGE of the feature maps that the matrix form of the cost volume reads (mnerf_cost_volume_operands,
 *       mnerf_cost_volume_operand_bytes).  NULL keeps the segment walk.
 */
#ifndef MNERF_H_
#define MNERF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNERF_ABI_VERSION 9
#define MNERF_MAX_VIEWS 16
#define MNERF_FEAT_CH 128 /* channels of one pair-specific GMFlow feature map */
/* floats per sample of a `cond` buffer: sum(cos_n_group) + 4 n_views + 1, rounded up to a multiple of 8.
 * 16 views x ([8,8] groups) = 81 -> 88 fits the split weight-stream formats; the exact-f32 stream keeps its FiLM
 * inputs in registers and is limited to 64 (13 views with the shipped [2,8] groups). */
#define MNERF_COND_STRIDE_MAX 96
#define MNERF_COND_STRIDE_MAX_F32 64

enum {
  MNERF_OK = 0,
  MNERF_E_NULL = -1,        /* required pointer is NULL                         */
  MNERF_E_RANGE = -2,       /* a size / count is out of the supported range     */
  MNERF_E_UNSUPPORTED = -3, /* valid in the reference, not built in this kernel */
  MNERF_E_ALIGN = -4        /* pointer not aligned as required (16 B)           */
};

/* One source view (models/matchnerf.py:75-86 `ref_poses`). */
typedef struct mnerf_view {
  float extr[12]; /* world->camera [R|t], row-major 3x4  (batch.extrinsics[:, v, :3, :]) */
  float intr[9];  /* 3x3 intrinsics                       (batch.intrinsics[:, v])        */
  float near_, far_;
} mnerf_view;

/* The rays of one chunk of one target view.
 * Replaces camera.get_center_and_ray + sample_depth + get_3D_points_from_depth, which the
 * reference evaluates for the FULL image on every chunk (models/matchnerf.py:104-115,
 * misc/camera.py:255-286): here each ray is rebuilt in-kernel from the pixel index. */
typedef struct mnerf_rays {
  int32_t n_rays;         /* rays in this chunk                                               */
  int32_t n_samples;      /* S = opt.nerf.sample_intvs                                        */
  int32_t ray_begin;      /* first pixel index (row-major y*W+x) when ray_idx == NULL          */
  int32_t legacy_coord;   /* opt.nerf.legacy_coord: integer pixel centres, i/(S-1) depths      */
  int32_t depth_inverse;  /* opt.nerf.depth.param == "inverse"                                 */
  int32_t height, width;  /* image size of the views                                          */
  const int32_t* ray_idx; /* device, [n_rays] pixel indices (train / test-optim), or NULL      */
  const float* strat_u;   /* device, [n_rays, S] U[0,1) offsets (stratified train), or NULL    */
  float kinv[9];          /* inverse target intrinsics, fp32 (camera.py:221-222)              */
  float c2w[12];          /* target camera->world 3x4 (legacy: fp64 inverse cast to fp32,      */
                          /* camera.py:231-240; else [R^T | -R^T t], camera.py:36-42)          */
  float near_, far_;      /* target near / far (batch.near_fars[:, -1])                       */
  /* POSE TABLE (ABI v7; video paths of small frames, models/matchnerf.py:42-71): several target poses in ONE launch.
   * With pose_table != NULL the launch's rays are the concatenation of the poses' frames: global ray g = ray_begin + r
   * renders pixel g % rays_per_pose of pose g / rays_per_pose, and kinv / c2w / near_ / far_ above are ignored in favour
   * of row `pose` of the table: [kinv 9 | c2w 12 | near | far | pad] = MNERF_POSE_FLOATS fp32.  rays_per_pose must be a
   * multiple of 64 (so that a wavefront never straddles two poses); ray_idx and strat_u must be NULL.  Accepted by
   * mnerf_cost_volume (<= 5 views), mnerf_decoder_chunk (split-fp16 stream, S <= 128, <= 5 views) and mnerf_render_chunk
   * (staged form) where mnerf_render_takes_pose_table() returns 1; MNERF_E_UNSUPPORTED everywhere else. */
  const float* pose_table;
  int32_t rays_per_pose;
  int32_t pad_;
} mnerf_rays;
#define MNERF_POSE_FLOATS 24

/* Source-view data resident in HBM.  Feature maps are PAIR-MAJOR and CHANNEL-LAST:
 *   feat[s] : [n_pairs][2][fh[s]][fw[s]][128] fp32,  pair p=(a,b), a<b lexicographic,
 *             side 0 = feature of view a, side 1 = feature of view b after the GMFlow
 *             transformer ran on (a,b)  (models/gmflow/gmflow.py:47-67).
 * The reference keeps them as per-view channel chunks [V,(V-1)*128,h,w] (matchnerf.py:192-205)
 * and pairs chunk j of view i with chunk i of view j+1 (matchnerf.py:260-268) — the same data.
 *   images  : [n_views][H][W][4] fp32, RGB in [0,1] + one pad float (16-byte texels). */
typedef struct mnerf_scene {
  int32_t n_views;
  int32_t n_scales;       /* 2: raw 1/8 features + up-sampled 1/4 features                    */
  int32_t fh[2], fw[2];
  int32_t n_group[2];     /* opt.encoder.cos_n_group, each in {1,2,4,8}                        */
  const float* feat[2];
  const float* images;
  mnerf_view views[MNERF_MAX_VIEWS];
  /* ABI v9: operand image of feat[] for the MATRIX FORM of the cost volume (mnerf_cost_volume_operands below), or NULL.
   * With it, mnerf_cost_volume / mnerf_render_chunk interpolate the 128 channels of every (pair, scale) on the matrix pipe
   * (v_mfma_f32_32x32x16_f16, split-fp16 operands) for launches over contiguous pixels of one pose (ray_idx == NULL, no pose
   * table); every other launch, and a NULL pointer, takes the segment walk on feat[] - which must stay valid either way
   * (the backward kernels and the walk read it). */
  const void* feat_op;
} mnerf_scene;

/* Decoder parameters, pre-packed by the host (matchnerf_amd/cond_nerf.py):
 *   wstream : MFMA A-operand fragments of every Linear of CondNeRF in consumption order
 *             (layout in DESIGN.md §Decoder weight stream); 16-byte aligned.  Two formats:
 *             MNERF_WSTREAM_F32   fp32 fragments for v_mfma_f32_32x32x2_f32 (pack_wstream)
 *             MNERF_WSTREAM_BF16X3 each fp32 weight as three bf16 terms for
 *                                 v_mfma_f32_32x32x16_bf16, six product terms per MAC, fp32
 *                                 accumulate, fp32 bias fragments (pack_wstream16)
 *             MNERF_WSTREAM_F16X2 each fp32 weight as two fp16 terms of 2^ew W (one power-of-two
 *                                 scale per tensor) for v_mfma_f32_32x32x16_f16, three product terms
 *                                 per MAC, per-sample power-of-two activation gains chosen in the
 *                                 kernel, fp32 accumulate and biases (pack_wstream_h; default)
 *   small   : ray-transformer + density-head parameters (fp32, layout in DESIGN.md)
 * Architecture switches mirror opt.decoder.* / opt.nerf.* (configs/base.yaml:29-48). */
#define MNERF_WSTREAM_F32 0
#define MNERF_WSTREAM_BF16X3 1
#define MNERF_WSTREAM_F16X2 2
#define MNERF_WSTREAM_F16X1 3 /* ABI v8: the F16X2 stream read as plain fp16 weights (hi terms only), ONE product per MAC on fp16
                                * activations, fp32 accumulation: the reduced-precision FAST mode (RGB L-inf ~4e-3 against the
                                * fp32 path: outside the 1e-4 parity gate, never the default).  Ping-pong decoder only (<= 5
                                * source views, sample_intvs <= 128, no pose table); MNERF_E_UNSUPPORTED elsewhere. */

typedef struct mnerf_decoder {
  const float* wstream;
  int64_t wstream_floats;
  const float* small_;
  int32_t n_views;         /* V = opt.n_src_views                                               */
  int32_t cond_dim;        /* sum(cos_n_group) + 4*V (cond_nerf.py:18)                          */
  int32_t cond_stride;     /* floats per sample in `cond` buffers: multiple of 8, > cond_dim, <= MNERF_COND_STRIDE_MAX[_F32] */
  int32_t L_3D;            /* opt.decoder.posenc.L_3D (L_view must be 0)                        */
  int32_t raytrans_posenc; /* opt.decoder.raytrans_posenc                                       */
  int32_t raytrans_elu;    /* opt.decoder.raytrans_act == "ELU" (else ReLU)                     */
  int32_t density_maskfill;/* opt.decoder.density_maskfill                                      */
  int32_t wo_render_interval; /* opt.nerf.wo_render_interval                                    */
  int32_t setbg_opaque;    /* MatchNeRF.nerf_setbg_opaque                                       */
  int32_t wstream_format;  /* MNERF_WSTREAM_*                                                    */
} mnerf_decoder;

int mnerf_abi_version(void);
const char* mnerf_last_error(void);
/* sizeof() of the argument structs as compiled (0 view, 1 rays, 2 scene, 3 decoder, 4 encoder_layer, 5 conv; -1 else):
 * lets a foreign-language binding verify its struct mirrors before the first call. */
int64_t mnerf_struct_size(int32_t which);

/* a8-a10 — target rays, depth samples, world points and their (u,v,z) in one source view.
 * Replaces camera.get_center_and_ray (misc/camera.py:255-278), MatchNeRF.sample_depth
 * (models/matchnerf.py:163-181), get_3D_points_from_depth (camera.py:281-286) and
 * get_coord_ref_ndc (camera.py:351-379).  Any of pts [R,S,3], ndc [R,S,3], depth [R,S] may be
 * NULL.  Bit-exact against the reference's CPU path (same k-ordered FMA chains). */
int mnerf_ray_samples(const mnerf_rays* rays, const mnerf_view* view, float* pts, float* ndc,
                      float* depth, void* stream);

/* K5 — volume-rendering quadrature.  Replaces NeRF.composite
 * (models/rfdecoder/nerf.py:101-124).  rgb_s [R,S,3], sigma [R,S], depth_s [R,S],
 * ray_len [R] (|ray|, only read when wo_render_interval == 0; may be NULL otherwise)
 * -> rgb [R,3], depth [R], opacity [R] and, when non-NULL, prob [R,S] (the per-sample weights
 * T_j alpha_j that the reference returns as its fourth value, nerf.py:116,124). */
int mnerf_composite(int32_t n_rays, int32_t n_samples, const float* rgb_s, const float* sigma,
                    const float* depth_s, const float* ray_len, int32_t wo_render_interval,
                    int32_t setbg_opaque, float* rgb, float* depth, float* opacity, float* prob,
                    void* stream);

/* K1+K2 — epipolar feature sampling + group-cosine cost volume + colours + visibility mask.
 * Replaces MatchNeRF.query_cond_info (models/matchnerf.py:209-293), sample_features_by_grid
 * (models/gmflow/utils.py:131-134) and get_coord_ref_ndc (misc/camera.py:351-379).
 * cond [n_rays*S, cond_stride]: [cos@scale0 | cos@scale1 | rgb_v0.. | mask_v0.. | 1.0 | 0 pad]
 * (the order CondNeRF concatenates them in, cond_nerf.py:59; the trailing 1.0 feeds the
 * bias column of the packed FiLM layer). */
int mnerf_cost_volume(const mnerf_scene* scene, const mnerf_rays* rays, int32_t cond_stride,
                      float* cond, void* stream);

/* ABI v9 — operand image for the matrix form of K1+K2: the feature maps of scene->feat[] re-laid for the matrix pipe, once per
 * source set (two launches, ~0.1 ms at 3 views of 512x640; enqueue-only).  Every fp32 texel x becomes two fp16 terms of 2^e x
 * (one power-of-two gain per map, taken from the map's largest magnitude, so that the cosine - which is scale-free per map -
 * is unchanged); texels are stored in aligned 2-row x 4-column cells, channel-tile-major, so that the 16 texels of a 4 x 4
 * chunk are the K dimension of one v_mfma_f32_32x32x16_f16 A operand per 32 channels, read with fully coalesced 16-byte loads
 * (layout: matchnerf_amd/csrc/cost_volume_mm.hip).  `opnd`: caller-owned, 16-byte aligned, mnerf_cost_volume_operand_bytes(scene)
 * bytes (about the size of the fp32 maps); scene->feat_op is ignored by both functions.  Replaces nothing in the reference: it
 * is the layout change that lets sample_features_by_grid (models/gmflow/utils.py:131-134) run as a matrix product. */
int64_t mnerf_cost_volume_operand_bytes(const mnerf_scene* scene);
int mnerf_cost_volume_operands(const mnerf_scene* scene, void* opnd, void* stream);

/* K3+K4+K5 — conditional radiance MLP, per-ray transformer and compositing for one chunk.
 * Replaces CondNeRF.forward (models/rfdecoder/cond_nerf.py:52-100), MultiHeadAttention
 * (ray_transformer.py:29-79), the ref-view-0 NDC warp and view-direction rotation
 * (models/matchnerf.py:118-132) and NeRF.composite (nerf.py:101-124).
 * view0 = source view 0 (coordinate reference).  cond as produced by mnerf_cost_volume.
 * Outputs rgb [R,3], depth [R], opacity [R]; dbg_rgb_s [R,S,3] / dbg_sigma [R,S] receive the
 * per-sample decoder outputs when non-NULL (parity tests). */
/* expected wstream size in 4-byte words for a format (negative: cannot be scheduled) */
int64_t mnerf_decoder_wstream_floats(int32_t cond_dim, int32_t cond_stride, int32_t L_3D,
                                     int32_t wstream_format);
int mnerf_decoder_chunk(const mnerf_decoder* dec, const mnerf_view* view0, const mnerf_rays* rays,
                        const float* cond, float* rgb, float* depth, float* opacity,
                        float* dbg_rgb_s, float* dbg_sigma, void* stream);

/* K3+K4 with caller-supplied inputs — the literal CondNeRF.forward(opt, points_3D, ray_unit, cond_info)
 * (models/rfdecoder/cond_nerf.py:52-100): x_ndc [R,S,3] sample coordinates w.r.t. source view 0
 * (matchnerf.py:121-126), dir [R,S,3] unit view directions in that view's frame (matchnerf.py:129-132),
 * cond [R*S, cond_stride] = cat(feat_info, color_info, mask_info) (+ the constant 1 and zero padding of
 * mnerf_cost_volume's layout) -> rgb_s [R,S,3], sigma [R,S].  Same kernel as mnerf_decoder_chunk without
 * the in-kernel ray geometry and without compositing; `legacy_coord` selects the positional-encoding
 * frequencies (2^l vs 2^l pi) and must match the packing of dec->wstream. */
int mnerf_decoder_samples(const mnerf_decoder* dec, int32_t n_rays, int32_t n_samples,
                          int32_t legacy_coord, const float* x_ndc, const float* dir,
                          const float* cond, float* rgb_s, float* sigma, void* stream);

/* a7 — one full render chunk = cost volume + decoder + compositing
 * (MatchNeRF.render, models/matchnerf.py:88-143).  Two forms with identical results, bit for bit:
 *   mnerf_render_chunk        staged: mnerf_cost_volume -> `workspace` -> mnerf_decoder_chunk (two launches);
 *                             `workspace` must hold mnerf_render_workspace_bytes(n_rays, n_samples, cond_stride)
 *                             bytes.  The default because it is the faster one on MI355X.
 *   mnerf_render_chunk_fused  ONE launch: the ray-chunk kernel produces the conditioning rows of each tile in LDS
 *                             and consumes them there (no HBM hand-off, no workspace).  Available where
 *                             mnerf_render_chunk_is_fused() returns 1: split-fp16 weight stream, S <= 128, at most
 *                             5 views / 32 conditioning inputs, cos_n_group entries >= 2; MNERF_E_UNSUPPORTED
 *                             otherwise.  MNERF_RENDER_FUSED=1 (read at load) makes mnerf_render_chunk take this
 *                             form wherever it is available (`workspace` may then be NULL). */
int64_t mnerf_render_workspace_bytes(int32_t n_rays, int32_t n_samples, int32_t cond_stride);
int32_t mnerf_render_chunk_is_fused(const mnerf_scene* scene, const mnerf_decoder* dec, const mnerf_rays* rays);
/* 1 if mnerf_render_chunk accepts a mnerf_rays.pose_table for this scene / decoder / sample count / frame size (rays_per_pose
 * = H*W): the kernel instances that take a table are the ones of the shipped shape (<= 5 source views, split-fp16 stream,
 * sample_intvs <= 128, H*W a multiple of 64).  A caller renders pose by pose otherwise. */
int32_t mnerf_render_takes_pose_table(const mnerf_scene* scene, const mnerf_decoder* dec, int32_t n_samples,
                                      int32_t rays_per_pose);
int mnerf_render_chunk(const mnerf_scene* scene, const mnerf_decoder* dec, const mnerf_rays* rays,
                       void* workspace, float* rgb, float* depth, float* opacity, void* stream);
int mnerf_render_chunk_fused(const mnerf_scene* scene, const mnerf_decoder* dec, const mnerf_rays* rays,
                             float* rgb, float* depth, float* opacity, void* stream);

/* Backward kernels of the ray chunk (training through the HIP path; reference: autograd through the eager chain,
 * coach.py:215-243).
 * K5 backward — given d(rgb [R,3], depth [R] or NULL, opacity [R] or NULL) and the forward's per-sample inputs,
 * writes g_rgb_s [R,S,3] and g_sigma [R,S] (NeRF.composite, nerf.py:101-124). */
int mnerf_composite_backward(int32_t n_rays, int32_t n_samples, const float* rgb_s, const float* sigma,
                             const float* depth_s, const float* ray_len, int32_t wo_render_interval,
                             int32_t setbg_opaque, const float* g_rgb, const float* g_depth,
                             const float* g_opacity, float* g_rgb_s, float* g_sigma, void* stream);
/* K1+K2 backward — g_cond [n_rays*S, cond_stride] (gradient of mnerf_cost_volume's rows; only the cosine
 * entries are read) is scattered into the feature-map gradients g_feat0 / g_feat1 (layouts of scene->feat[0] /
 * feat[1]; ACCUMULATED with atomic adds: the caller zero-fills them; g_feat1 may be NULL when n_scales == 1).
 * The forward interpolation is recomputed from `scene` and `rays` (query_cond_info, matchnerf.py:209-293). */
int mnerf_cost_volume_backward(const mnerf_scene* scene, const mnerf_rays* rays, int32_t cond_stride,
                               const float* g_cond, float* g_feat0, float* g_feat1, void* stream);

/* Test / diagnosis hook: set one tuning knob after load (the lower-case name of its MNERF_* environment variable without the
 * prefix: "decoder_pp", "cv_variant", ...).  Returns MNERF_OK and the previous value through *old_value (may be NULL), or
 * MNERF_E_RANGE + a message for an unknown name (ABI v6: the status no longer shares the return value with the old value,
 * whose legitimate range includes -1).  The table is the library's only mutable process-wide state besides the thread-local
 * error string; this hook is its only writer.  Writes and the launches' reads are serialised by a lock and every launch works
 * on a copy of the table, so a launch on another thread sees the value from before or after the call, never a torn table;
 * kernels already enqueued are not affected. */
int mnerf_debug_set_knob(const char* name, int value, int* old_value);

/* K3+K4 backward — gradients of the conditional MLP + ray transformer (CondNeRF.forward, cond_nerf.py:52-100;
 * MultiHeadAttention.forward, ray_transformer.py:29-79; what `loss.backward()` does to them in coach.py:215-243).
 * The parameters are given in torch's own layouts (Linear.weight [out,in], fp32), indexed by MNERF_DT_*; `g[k]` receives the
 * gradient of `w[k]`, ACCUMULATED (the caller zero-fills or carries over), NULL = not wanted. */
enum {
  MNERF_DT_PTS_W0 = 0,   /* pts_linears.i.weight = 2 i, .bias = 2 i + 1, i = 0..5 */
  MNERF_DT_BIAS_W = 12,  /* pts_bias (the FiLM multiplier) */
  MNERF_DT_BIAS_B,
  MNERF_DT_ALPHA_W,      /* alpha_linear.0 */
  MNERF_DT_ALPHA_B,
  MNERF_DT_WQ,           /* ray_attention.w_qs / w_ks / w_vs / fc (.weight, no bias) */
  MNERF_DT_WK,
  MNERF_DT_WV,
  MNERF_DT_FC,
  MNERF_DT_LN_W,         /* ray_attention.layer_norm */
  MNERF_DT_LN_B,
  MNERF_DT_OA0_W,        /* out_alpha_linear.0 */
  MNERF_DT_OA0_B,
  MNERF_DT_OA2_W,        /* out_alpha_linear.2 */
  MNERF_DT_OA2_B,
  MNERF_DT_FEAT_W,       /* feature_linear */
  MNERF_DT_FEAT_B,
  MNERF_DT_VIEWS_W,      /* views_linears.0 : [64, 128 + 3] */
  MNERF_DT_VIEWS_B,
  MNERF_DT_RGB_W,        /* rgb_linear */
  MNERF_DT_RGB_B,
  MNERF_DEC_TENSORS
};
typedef struct mnerf_decoder_train {
  int32_t n_views;          /* source views: the last n_views of the cond_dim conditioning columns are the visibility masks */
  int32_t cond_dim;         /* conditioning columns the FiLM Linear reads */
  int32_t n_trunk;          /* opt.decoder.net_depth (6) */
  int32_t net_width;        /* opt.decoder.net_width (128) */
  int32_t skip_layer;       /* the trunk layer after which the encoding is concatenated again (opt.decoder.skip[0]); -1: none */
  int32_t L_3D;             /* octaves of the positional encoding */
  int32_t legacy_coord;     /* opt.nerf.legacy_coord: frequency 2^l and [l][xyz] order instead of pi 2^l and [xyz][sin|cos][l] */
  int32_t raytrans_elu;     /* opt.decoder.raytrans_act == 'ELU' (else ReLU) */
  int32_t raytrans_posenc;  /* add raytrans_table [S,16] to the alpha features */
  int32_t density_maskfill;
  const float* raytrans_table;
  const float* w[MNERF_DEC_TENSORS];
  float* g[MNERF_DEC_TENSORS];
} mnerf_decoder_train;
/* x_ndc [N,3] sample coordinates and dirs [n_rays,3] unit view directions in source view 0's frame (what mnerf_ray_samples and the
 * host compute for the forward), cond [N, cond_stride] the rows mnerf_cost_volume wrote, g_rgb_s [N,3] / g_sigma [N] from
 * mnerf_composite_backward (N = n_rays * n_samples, n_samples <= 256).  g_cond [N, cond_stride] (or NULL) receives the gradient
 * of the first cond_dim columns of the rows (other columns untouched).  workspace: mnerf_decoder_backward_workspace_bytes(). */
int64_t mnerf_decoder_backward_workspace_bytes(int32_t n_rays, int32_t n_samples);
int mnerf_decoder_backward(const mnerf_decoder_train* dec, int32_t n_rays, int32_t n_samples, const float* x_ndc,
                           const float* dirs, const float* cond, int32_t cond_stride, const float* g_rgb_s,
                           const float* g_sigma, float* g_cond, void* workspace, void* stream);

/* K6 — GMFlow single-head (shifted-)window attention, flash style (no score matrix).
 * Replaces single_head_split_window_attention / single_head_full_attention and the
 * shift-mask tensor (models/gmflow/transformer.py:8-16, 19-43, 46-105).
 * q,k,v,out [batch, h*w, 128]; num_splits >= 1 (1 = full attention); `shifted` applies the
 * swin roll by half a window with wrap-region masking (-100 added across regions).
 * `math`: MNERF_WA_SPLIT_BF16 (fp32-grade products from three bf16 terms per operand, six products per MAC on the
 * bf16 matrix cores), MNERF_WA_SPLIT_F16 (two range-managed fp16 terms,
 * three products; gains per query and per 32-key K / V tile) or MNERF_WA_EXACT_F32 (v_mfma_f32_32x32x2_f32). */
#define MNERF_WA_SPLIT_BF16 0
#define MNERF_WA_EXACT_F32 1
#define MNERF_WA_SPLIT_F16 2
int mnerf_window_attention(const float* q, const float* k, const float* v, float* out,
                           int32_t batch, int32_t h, int32_t w, int32_t num_splits,
                           int32_t shifted, int32_t math, void* stream);

/* K6 with the K / V matrix operands prepared once per call (the host's default): a first small kernel turns every
 * 32-key tile of every window into split-fp16 operand fragments + one power-of-two gain per tile and matrix in
 * `workspace`, the attention kernel streams those images through LDS (arithmetic = MNERF_WA_SPLIT_F16; its waves
 * no longer convert the window's K and V once per 32 queries).  Same arguments and semantics as above;
 * `workspace`: device, 16-byte aligned, at least mnerf_window_attention_workspace_bytes(batch, h, w, num_splits)
 * bytes (= the size of k and v together + 32 bytes per tile), scratch only - nothing is kept between calls. */
size_t mnerf_window_attention_workspace_bytes(int32_t batch, int32_t h, int32_t w, int32_t num_splits);
int mnerf_window_attention_presplit(const float* q, const float* k, const float* v, float* out,
                                    int32_t batch, int32_t h, int32_t w, int32_t num_splits, int32_t shifted,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* q | k | v projections of a GMFlow transformer layer in one launch (models/gmflow/transformer.py:147-151):
 *   q = Wq x_q,  k = Wk x_kv',  v = Wv x_kv'      x_q, x_kv, q, k, v: [n_seq, seq_len, 128] fp32 tokens
 * kv_swap: x_kv' is x_kv with its two batch halves exchanged (sequence b reads sequence (b + n_seq/2) mod n_seq): the
 * "target" of the batched pair members (transformer.py:317-335) without building the concatenated tensor.
 * wstream: split-fp16 A-operand fragments of [Wq | Wk | Wv] (matchnerf_amd/gmflow.py: pack_qkv;
 * mnerf_qkv_wstream_floats() words, device), ew: the three weight-scale exponents (HOST array of 3). */
int64_t mnerf_qkv_wstream_floats(void);
int mnerf_qkv_projection(const float* wstream, const int32_t* ew, const float* x_q, const float* x_kv, int32_t kv_swap,
                         float* q, float* k, float* v, int32_t n_seq, int32_t seq_len, void* stream);

/* The same projections with K and V written directly as the window attention's operand images (no k / v tensors, no
 * operand pre-pass): q [batch, h*w, 128] fp32 + `workspace` in the format mnerf_window_attention_images consumes
 * (mnerf_window_attention_workspace_bytes(batch, h, w, num_splits) bytes).  The geometry arguments must be the ones the
 * attention call will get. */
int mnerf_qkv_window_images(const float* wstream, const int32_t* ew, const float* x_q, const float* x_kv, int32_t kv_swap,
                            float* q, void* workspace, size_t workspace_bytes, int32_t batch, int32_t h, int32_t w,
                            int32_t num_splits, int32_t shifted, void* stream);
/* K6 on a workspace filled by mnerf_qkv_window_images (or by a previous mnerf_window_attention_presplit call) */
int mnerf_window_attention_images(const float* q, float* out, int32_t batch, int32_t h, int32_t w, int32_t num_splits,
                                  int32_t shifted, const void* workspace, size_t workspace_bytes, void* stream);

/* K6 backward — gradients of the (shifted-)window attention (what `loss.backward()` does to
 * models/gmflow/transformer.py:46-105 in coach.py:215-243), flash style: the [windows, L_w, L_w] score tensor is never
 * materialised.  q, k, v, out (the forward's result), g_out (gradient of `out`): [batch, h*w, 128] fp32; g_q, g_k, g_v
 * (same shape) are OVERWRITTEN.  fp32-grade matrix products (MNERF_WA_BWD_MATH = "f16x3", the default: two fp16 terms per operand with
 * power-of-two gains from the tensors' maxima, three term products; "bf16x6": three bf16 terms, six term products; "f32": the
 * exact-f32 matrix instruction), deterministic (the only atomics are the maxima of the f16x3 form's gains).  `out` must be the
 * forward's result for these q, k, v (the f16x3 gains bound <g_out, out> by the maxima of g_out and v).  `workspace`: at least
 * mnerf_window_attention_backward_workspace_bytes(batch, h, w) bytes (four floats per token: row maximum, row sum of
 * exponentials, <g_out, out>, |g_out|^2; + the four operand maxima). */
int64_t mnerf_window_attention_backward_workspace_bytes(int32_t batch, int32_t h, int32_t w);
int mnerf_window_attention_backward(const float* q, const float* k, const float* v, const float* out, const float* g_out,
                                    float* g_q, float* g_k, float* g_v, int32_t batch, int32_t h, int32_t w,
                                    int32_t num_splits, int32_t shifted, void* workspace, size_t workspace_bytes,
                                    void* stream);
/* Training pair: the forward that also publishes the softmax's row statistics, row_stats = [2][batch * h*w] fp32 (running
 * maximum | sum of exponentials, log2 domain, token order), and the backward that reads them instead of recomputing them in a
 * first pass over all keys.  Same results as the pair above up to the last bits of the statistics. */
int mnerf_window_attention_presplit_stats(const float* q, const float* k, const float* v, float* out, float* row_stats,
                                          int32_t batch, int32_t h, int32_t w, int32_t num_splits, int32_t shifted,
                                          void* workspace, size_t workspace_bytes, void* stream);
int mnerf_window_attention_backward_stats(const float* q, const float* k, const float* v, const float* out, const float* g_out,
                                          const float* row_stats, float* g_q, float* g_k, float* g_v, int32_t batch, int32_t h,
                                          int32_t w, int32_t num_splits, int32_t shifted, void* workspace,
                                          size_t workspace_bytes, void* stream);

/* InstanceNorm2d (no affine, biased variance, as torch.nn.functional.instance_norm) of an NCHW tensor fused with what
 * follows it in the GMFlow backbone (models/gmflow/backbone.py:27-35, 101-103):
 *   v = (x - mean_plane) / sqrt(var_plane + eps);  if relu_inner: v = max(v, 0);
 *   if residual: v += residual;                     if relu_outer: v = max(v, 0)
 * x, residual (or NULL), out: [planes = N*C][plane_size = H*W] fp32; out may alias x.  One workgroup per plane.
 * out_absmax: absmax region (MNERF_ABSMAX_FLOATS floats, see mnerf_conv2d) or NULL; max |out| is merged into it (the
 * operand scale of the convolution that reads `out`). */
int mnerf_instance_norm(const float* x, const float* residual, float* out, int64_t planes, int64_t plane_size,
                        float eps, int32_t relu_inner, int32_t relu_outer, float* out_absmax, void* stream);

/* Backward of out = [relu](InstanceNorm(x)) (training path; what autograd runs for F.instance_norm + F.relu in
 * models/gmflow/backbone.py:27-35, 101-103): dx from x (the norm's INPUT, the statistics are re-derived) and dy.  relu: the forward
 * applied a ReLU to the normalised value.  x, dy, dx: [planes][plane_size] fp32; dx may alias dy. */
int mnerf_instance_norm_backward(const float* x, const float* dy, float* dx, int64_t planes, int64_t plane_size, float eps,
                                 int32_t relu, void* stream);

/* F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) of the up-sampler's training path (superres.py:37) and its
 * backward.  in [planes][h][w] -> out [planes][2h][2w] (+ add, same shape as out, or NULL); dout [planes][2h][2w] -> din. */
int mnerf_upsample_bilinear2x(const float* in, const float* add, float* out, int64_t planes, int32_t h, int32_t w, void* stream);
int mnerf_upsample_bilinear2x_backward(const float* dout, float* din, int64_t planes, int32_t h, int32_t w, void* stream);

/* Convolutions of the GMFlow backbone / up-sampler (models/gmflow/backbone.py:6-122, superres.py:5-38) as implicit
 * GEMMs with fp32-grade split-fp16 products (matchnerf_amd/csrc/conv.hip).  Built: c_in a multiple of 32, c_out 64 /
 * 96 / 128, 1x1 and 3x3 filters with padding ksize/2, stride 1 / 2.
 *   wstream     : A-operand fragments of 2^ew * W, K16-step s = (tap, 16 input channels), unit (s, 32-row block) =
 *                 [hi | lo] x 64 lanes x 8 fp16 (matchnerf_amd/gmflow.py: pack_conv); mnerf_conv_wstream_floats() words
 *   bias        : [c_out] or NULL;  leaky_slope: LeakyReLU slope applied to the result (1 = none)
 * mnerf_conv2d: in [n_img, c_in, h_in, w_in] (or [n_img, h_in, w_in, c_in] with in_channels_last), optionally read
 * through a nearest 2x up-sampling (upsample2x); out [n_img, c_out, h_out, w_out] NCHW.
 *   in_absmax   : absmax region (MNERF_ABSMAX_FLOATS floats) whose maximum is >= max |in| (operands are scaled by ONE
 *                 power of two taken from it; a value that is too small overflows fp16) - filled by the producer of
 *                 `in`: mnerf_instance_norm, mnerf_conv2d (out_absmax) or mnerf_absmax
 *   add_bilinear2x : NULL, or [n_img, c_out, h_out/2, w_out/2] NCHW whose bilinear 2x up-sampling (align_corners=False,
 *                 as F.interpolate) is added to the result (superres.py:37: right = up(right) + conv(left))
 *   out_layout  : MNERF_CONV_OUT_NCHW, _CHANNEL_LAST (the transformer's tokens) or _PAIR_MAJOR (image i of the batched
 *                 pair members [a-sides; b-sides] goes to (pair i mod n_img/2, side i div n_img/2))
 *   add_channel_last : NULL, or a [h_out*w_out][c_out] tile added to every image of a channel-last result (the window
 *                 position embedding, gmflow/utils.py:68-88, fused into the backbone's last convolution)
 *   out_absmax  : absmax region or NULL; max |out| is merged into it with atomic maxima (zero it first). */
#define MNERF_ABSMAX_SLOTS 64   /* an "absmax region" = SLOTS partial maxima, STRIDE floats apart: 2048 floats, zeroed */
#define MNERF_ABSMAX_STRIDE 32  /* by the caller; producers merge with atomic maxima, the convolution reduces the slots */
#define MNERF_ABSMAX_FLOATS (MNERF_ABSMAX_SLOTS * MNERF_ABSMAX_STRIDE)
typedef struct mnerf_conv {
  const float* wstream;
  int64_t wstream_floats;
  const float* bias;
  int32_t c_in, c_out, ksize, stride;
  int32_t ew;
  float leaky_slope;
} mnerf_conv;
int64_t mnerf_conv_wstream_floats(int32_t c_in, int32_t c_out, int32_t ksize);
#define MNERF_CONV_OUT_NCHW 0
#define MNERF_CONV_OUT_CHANNEL_LAST 1 /* tokens [n_img][h_out][w_out][c_out] */
#define MNERF_CONV_OUT_PAIR_MAJOR 2   /* the cost volume's feature layout [n_img/2][2][h_out][w_out][c_out] */
int mnerf_conv2d(const mnerf_conv* cv, const float* in, int32_t in_channels_last, int32_t upsample2x,
                 const float* in_absmax, const float* add_bilinear2x, const float* add_channel_last, float* out,
                 int32_t out_layout, float* out_absmax, int32_t n_img, int32_t h_in, int32_t w_in, void* stream);
/* The backbone's stem: Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (models/gmflow/backbone.py:45, 101), same
 * arithmetic as mnerf_conv2d.  wstream: fragments of the [64, 147 -> 160] matrix (k = 3 (7 ky + kx) + c;
 * matchnerf_amd/gmflow.py: pack_conv_stem; mnerf_conv_stem_wstream_floats() words).  in [n_img,3,h_in,w_in] ->
 * out [n_img,64,(h_in-1)/2+1,(w_in-1)/2+1] NCHW. */
int64_t mnerf_conv_stem_wstream_floats(void);
int mnerf_conv_stem(const float* wstream, int32_t ew, const float* in, const float* in_absmax, float* out,
                    int32_t n_img, int32_t h_in, int32_t w_in, void* stream);
/* max |x| of n floats merged into the absmax region `out` (atomic maxima; zero it first) */
int mnerf_absmax(const float* x, int64_t n, float* out, void* stream);

/* Backward of Conv2d(c_in, c_out, ksize, stride, padding = ksize / 2) for the training path of the GMFlow backbone / up-sampler
 * (what autograd runs for models/gmflow/backbone.py:6-122, superres.py:5-38 under coach.py:215-243); exact-fp32 products on
 * v_mfma_f32_32x32x2_f32, NCHW fp32 (matchnerf_amd/csrc/conv_backward.hip).  Built: c_in, c_out in {32, 64, 96, 128}, ksize 1 / 3,
 * stride 1 / 2.  h_in, w_in: the convolution's INPUT size; dy is [n_img, c_out, h_out, w_out] with h_out = (h_in + 2 (ksize/2) -
 * ksize) / stride + 1.
 *   mnerf_conv2d_backward_data  : dx [n_img, c_in, h_in, w_in] = conv_transpose(dy, W).  w_tap_major: the weight permuted to
 *                                 [ky][kx][c_out][c_in] (torch: weight.permute(2, 3, 0, 1).contiguous()).  dx is overwritten.
 *   mnerf_conv2d_backward_weight: dw [c_out, c_in, ksize, ksize] (torch's layout, overwritten) = sum over images and positions;
 *                                 partial sums of row chunks go through `workspace` (mnerf_conv2d_backward_weight_workspace_bytes)
 *                                 and are added in chunk order: bit-reproducible. */
int mnerf_conv2d_backward_data(const float* dy, const float* w_tap_major, float* dx, int32_t n_img, int32_t c_in, int32_t c_out,
                               int32_t h_in, int32_t w_in, int32_t ksize, int32_t stride, void* stream);
size_t mnerf_conv2d_backward_weight_workspace_bytes(int32_t n_img, int32_t c_in, int32_t c_out, int32_t h_in, int32_t w_in,
                                                    int32_t ksize, int32_t stride);
int mnerf_conv2d_backward_weight(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int32_t n_img,
                                 int32_t c_in, int32_t c_out, int32_t h_in, int32_t w_in, int32_t ksize, int32_t stride, void* stream);
/* The weight gradient on the 16-bit matrix pipe: operands as two fp16 terms with one power-of-two gain per tensor, three products per
 * MAC, fp32 accumulation (the forward kernels' arithmetic).  x_absmax / dy_absmax: absmax regions (MNERF_ABSMAX_FLOATS floats, see
 * mnerf_conv2d) holding max |x| and max |dy| - filled by mnerf_absmax or the tensors' producers.  Same workspace. */
int mnerf_conv2d_backward_weight_f16x3(const float* x, const float* dy, const float* x_absmax, const float* dy_absmax, float* dw,
                                       void* workspace, size_t workspace_bytes, int32_t n_img, int32_t c_in, int32_t c_out, int32_t h_in,
                                       int32_t w_in, int32_t ksize, int32_t stride, void* stream);
/* The training FORWARD of the same convolutions in exact fp32 (the gradients' arithmetic; no weight stream to re-pack after every
 * optimizer step): y [n_img, c_out, h_out, w_out] = conv(x, W) + bias.  w_tap_major: the weight permuted to [ky][kx][c_in][c_out]
 * (torch: weight.permute(2, 3, 1, 0).contiguous()); bias [c_out] or NULL.  c_in: any count up to 128 (the stem: 3), c_out a
 * multiple of 32 up to 128, ksize 1 / 3 / 7, stride 1 / 2. */
int mnerf_conv2d_forward_f32(const float* x, const float* w_tap_major, const float* bias, float* y, int32_t n_img, int32_t c_in,
                             int32_t c_out, int32_t h_in, int32_t w_in, int32_t ksize, int32_t stride, void* stream);
/* Weight gradient of the stem Conv2d(3, 64, 7, stride 2, padding 3) (backbone.py:45): dw [64, 3, 7, 7]; x [n_img, 3, h_in, w_in],
 * dy [n_img, 64, (h_in - 1) / 2 + 1, (w_in - 1) / 2 + 1].  (Its data gradient is the gradient of the images: never needed.) */
size_t mnerf_conv_stem_backward_weight_workspace_bytes(int32_t n_img, int32_t h_in, int32_t w_in);
int mnerf_conv_stem_backward_weight(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int32_t n_img,
                                    int32_t h_in, int32_t w_in, void* stream);

/* K7 — what follows the window attention inside one GMFlow transformer layer, as one kernel
 * (TransformerLayer.forward, models/gmflow/transformer.py:176-185):
 *   message = norm1(merge(attn));  [ffn:] message = norm2(mlp.2(GELU(mlp.0(cat[source, message]))));  out = source + message
 * attn, source, out: [n_tokens, 128] fp32 (out may alias neither input).  The three bias-free Linears travel as one
 * split-fp16 MFMA A-fragment stream (matchnerf_amd/gmflow.py: pack_encoder_block; layout in DESIGN.md section 5):
 * 32 KiB segments = 4 K16-steps x 4 row blocks x [hi | lo]; merge (2 segments), then per 128-unit hidden chunk
 * mlp.0 (4 segments: source features, then the message in accumulator order) and mlp.2 (2 segments).
 * ew_*: exponent of the power-of-two scale each weight tensor was packed with; ln: [4][128] = norm1 weight | norm1
 * bias | norm2 weight | norm2 bias (norm2 ignored when ffn == 0).  fp32-grade results (same arithmetic as the
 * decoder's MNERF_WSTREAM_F16X2 path). */
typedef struct mnerf_encoder_layer {
  const float* wstream;
  int64_t wstream_floats;  /* mnerf_encoder_block_wstream_floats(ffn) */
  const float* ln;
  int32_t ffn;             /* 0: self-attention layer (no FFN), 1: cross-attention + FFN layer */
  int32_t ew_merge, ew_w1, ew_w2;
} mnerf_encoder_layer;
int64_t mnerf_encoder_block_wstream_floats(int32_t ffn);
int mnerf_encoder_block(const mnerf_encoder_layer* blk, const float* attn, const float* source, float* out,
                        int32_t n_tokens, void* stream);

/* Backward of a transformer layer around the window attention (what `loss.backward()` does to
 * models/gmflow/transformer.py:108-185 in coach.py:215-243; round 4).  Parameters in torch's layouts, gradients ACCUMULATED
 * into the g_* tensors (a NULL g_* skips that parameter).  Matrix products with I, J >= 128 run fp32-grade in split-bf16 (three bf16 terms per operand,
 * six products per MAC on the bf16 MFMA) by default, exact f32 (v_mfma_f32_32x32x2_f32) with MNERF_GEMM_MATH=f32 or for smaller products
 * (csrc/gemm_f32.hpp). */
typedef struct mnerf_encoder_layer_train {
  int32_t ffn;            /* 0: self-attention layer (merge + norm1 only), 1: + cat / mlp / norm2 */
  int32_t pad_;
  const float* w_merge;   /* merge.weight [128,128] */
  const float* ln1_w;     /* norm1.weight / bias [128] */
  const float* ln1_b;
  const float* w_mlp0;    /* mlp.0.weight [1024,256] (columns: source | message) */
  const float* w_mlp2;    /* mlp.2.weight [128,1024] */
  const float* ln2_w;
  const float* ln2_b;
  float* g_w_merge;
  float* g_ln1_w;
  float* g_ln1_b;
  float* g_w_mlp0;
  float* g_w_mlp2;
  float* g_ln2_w;
  float* g_ln2_b;
} mnerf_encoder_layer_train;
/* attn (the attention's output), source (the layer's input), g_out (gradient of the layer's output): [n_tokens,128] ->
 * g_attn, g_source [n_tokens,128] (OVERWRITTEN; g_source includes the residual path).  The chain of mnerf_encoder_block is
 * re-evaluated in fp32 with its pre-activations kept in `workspace` (mnerf_encoder_layer_backward_workspace_bytes:
 * 3 970 floats per token). */
int64_t mnerf_encoder_layer_backward_workspace_bytes(int32_t n_tokens);
int mnerf_encoder_layer_backward(const mnerf_encoder_layer_train* layer, const float* attn, const float* source,
                                 const float* g_out, float* g_attn, float* g_source, int32_t n_tokens, void* workspace,
                                 void* stream);
/* Training pair (round 6): the forward that also writes the pre-norm activations - merge's output before norm1 (m1 [n_tokens,128])
 * and, for a layer with an FFN, mlp.0's output before the GELU (z1 [n_tokens,1024]) and mlp.2's output before norm2 (m2
 * [n_tokens,128]; NULL without an FFN); `out` equal to mnerf_encoder_block's bit for bit -, and the backward that reads them instead
 * of re-evaluating them (four of an FFN layer's eleven matrix products).  Same workspace. */
int mnerf_encoder_block_save(const mnerf_encoder_layer* blk, const float* attn, const float* source, float* out, float* m1,
                             float* z1, float* m2, int32_t n_tokens, void* stream);
int mnerf_encoder_layer_backward_saved(const mnerf_encoder_layer_train* layer, const float* attn, const float* source,
                                       const float* g_out, const float* m1, const float* z1, const float* m2, float* g_attn,
                                       float* g_source, int32_t n_tokens, void* workspace, void* stream);
/* q = x_q Wq^T, k = x_kv Wk^T, v = x_kv Wv^T (mnerf_qkv_projection without the batch swap: the caller passes the key / value
 * source it used):  g_xq = g_q Wq and g_xkv = g_k Wk + g_v Wv are OVERWRITTEN ([n_tokens,128], two different buffers),
 * gw_* [128,128] += g_*^T x_* (NULL: skipped). */
int mnerf_qkv_backward(const float* w_q, const float* w_k, const float* w_v, const float* x_q, const float* x_kv,
                       const float* g_q, const float* g_k, const float* g_v, float* g_xq, float* g_xkv, float* gw_q,
                       float* gw_k, float* gw_v, int32_t n_tokens, void* stream);

/* Test hook — the strided GEMM under the backward entry points above:  C[I,J] (mode 0: =, 1: +=, 2: atomic +=)
 * sum_k A(i,k) B(k,j) (+ bias[j]),  A(i,k) = a[i sa_i + k sa_k],  B(k,j) = b[k sb_k + j sb_j],  C row stride sc_i.
 * math 1 = split-bf16 on the 16-bit matrix instruction (six term products per product, fp32-grade; what the library uses
 * for products with I, J >= 128 unless MNERF_GEMM_MATH=f32), 0 = the exact-f32 matrix instruction. */
int mnerf_debug_gemm(const float* a, int64_t sa_i, int64_t sa_k, const float* b, int64_t sb_k, int64_t sb_j, float* c,
                     int64_t sc_i, const float* bias, int32_t I, int32_t J, int32_t K, int32_t mode, int32_t math,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MNERF_H_ */
