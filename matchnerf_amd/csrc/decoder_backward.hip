// K3+K4 backward — gradients of the conditional MLP + ray transformer.
//
// Replaces the autograd pass of /root/reference/coach.py:215-243 through CondNeRF.forward
// (/root/reference/models/rfdecoder/cond_nerf.py:52-100) and MultiHeadAttention.forward (ray_transformer.py:29-79):
// given d(rgb_s [N,3]) and d(sigma [N]) of the N = rays x S samples of a chunk (from mnerf_composite_backward) it returns
// d(cond rows) (input of mnerf_cost_volume_backward) and ACCUMULATES the gradients of the decoder's 32 parameter tensors.
//
// Training chunks are small (rand_rays_train rays: 65 536 samples) and their activations fit HBM thousands of times over, so this
// is not one fused tile kernel like the forward: the forward is re-evaluated layer by layer from the saved conditioning rows
// with every pre-activation kept in a workspace (2 604 floats = 10.4 KB per sample), then walked backwards.  All matrix products — Y = X W^T,
// dX = dY W, dW = dY^T X — run through ONE strided MFMA GEMM (gemm_f32.hpp: fp32-grade split-bf16 `gemm_b6_kernel` for products with I, J >= 128 by default,
// the exact-fp32 `gemm_f32_kernel` with MNERF_GEMM_MATH=f32 and for the small ones; v_mfma_f32_32x32x2_f32, 64x64 tiles staged in
// LDS, strided operands so that transposes / column slices / the row stride of the conditioning rows cost nothing, split-K with
// float atomics for the weight gradients whose reduction runs over all samples).  What is not a matrix product is a handful
// of elementwise kernels, and everything 16 wide — alpha activation, q|k|v, the S x S attention of a ray with its softmax,
// fc + residual + LayerNorm, the density head — is `ray_head_kernel`: one workgroup per ray, one lane per sample, forward
// re-evaluated in registers and differentiated in place (attention backward in the flash form: D_i = <dO_i, O_i>, P recomputed
// from the saved row maxima and sums); it emits the small [N,16] operands whose outer products the GEMM turns into the
// gradients of the 16 x 16 tensors.
#include "gemm_f32.hpp"

// positional encoding of the sample coordinates (cond_nerf.py:108-131): [x, sin, cos] -> enc[N, 64] (column 63 zero) and the
// view direction of the sample's ray in dirs_s[N, 4] (column 3 zero)
__global__ __launch_bounds__(256) void encode_kernel(int N, int S, int L, int legacy, const float* __restrict__ x,
                                                     const float* __restrict__ dirs, float* __restrict__ enc, float* __restrict__ dirs_s) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float p[3] = {x[n * 3 + 0], x[n * 3 + 1], x[n * 3 + 2]};
  float* e = enc + (size_t)n * 64;
  e[0] = p[0], e[1] = p[1], e[2] = p[2];
  for (int l = 0; l < L; ++l) {
    const float f = legacy ? (float)(1 << l) : (float)(1 << l) * 3.14159265358979323846f;
    for (int c = 0; c < 3; ++c) {
      float sn, cs;
      sincosf(p[c] * f, &sn, &cs);
      if (legacy) {
        e[3 + l * 3 + c] = sn;
        e[3 + 3 * L + l * 3 + c] = cs;
      } else {
        e[3 + c * 2 * L + l] = sn;
        e[3 + c * 2 * L + L + l] = cs;
      }
    }
  }
  for (int c = 3 + 6 * L; c < 64; ++c) e[c] = 0.0f;
  const int r = n / S;
  dirs_s[n * 4 + 0] = dirs[r * 3 + 0];
  dirs_s[n * 4 + 1] = dirs[r * 3 + 1];
  dirs_s[n * 4 + 2] = dirs[r * 3 + 2];
  dirs_s[n * 4 + 3] = 0.0f;
}

// h = relu(z * film)
__global__ __launch_bounds__(256) void film_relu_kernel(long long n, const float* __restrict__ z, const float* __restrict__ film,
                                                        float* __restrict__ h) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) h[i] = fmaxf(z[i] * film[i], 0.0f);
}
// g = dL/dh, h = relu(z film):  dz = g [h>0] film,  dfilm += g [h>0] z
__global__ __launch_bounds__(256) void film_relu_bwd_kernel(long long n, const float* __restrict__ g, const float* __restrict__ h,
                                                            const float* __restrict__ z, const float* __restrict__ film,
                                                            float* __restrict__ dz, float* __restrict__ dfilm, int first) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gm = h[i] > 0.0f ? g[i] : 0.0f;
    dz[i] = gm * film[i];
    dfilm[i] = (first ? 0.0f : dfilm[i]) + gm * z[i];
  }
}
// h = relu(z) in place -> h;  dz = g [h>0]
__global__ __launch_bounds__(256) void relu_kernel(long long n, const float* __restrict__ z, float* __restrict__ h) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) h[i] = fmaxf(z[i], 0.0f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(long long n, const float* __restrict__ g, const float* __restrict__ h,
                                                       float* __restrict__ dz) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dz[i] = h[i] > 0.0f ? g[i] : 0.0f;
}
// rgb = sigmoid(zr[n, 0..2]) (row stride 4);  dzr = g_rgb rgb (1 - rgb), column 3 zero
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(int N, const float* __restrict__ zr, const float* __restrict__ g_rgb,
                                                          float* __restrict__ dzr) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float s = 1.0f / (1.0f + expf(-zr[n * 4 + c]));
    dzr[n * 4 + c] = g_rgb[n * 3 + c] * s * (1.0f - s);
  }
  dzr[n * 4 + 3] = 0.0f;
}
static inline int ew_grid(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : b);
}

// ------------------------------------------------------------------ everything 16 wide, one workgroup per ray, one lane per sample
struct RayHeadArgs {
  int S, cond_stride, cond_dim, n_views, elu, maskfill;
  const float* za;     // [N,16] alpha_linear pre-activation
  const float* cond;   // conditioning rows (visibility masks = last n_views of the cond_dim columns)
  const float* table;  // [S,16] ray-transformer position table or NULL
  const float* g_sigma;
  const float *wq, *wk, *wv, *wfc, *ln_w, *ln_b, *wo0, *bo0, *wo2, *bo2;
  // outputs, [N,16] unless noted
  float *dza, *a0, *dq, *dk, *dv, *o_att, *du, *y, *dy, *dyx, *t, *dt_pre;
  float* ds_pre;  // [N]
};

__device__ __forceinline__ float act_fwd(float x, int elu) { return elu ? (x > 0.0f ? x : expf(x) - 1.0f) : fmaxf(x, 0.0f); }
__device__ __forceinline__ float act_grad(float x, int elu) { return x > 0.0f ? 1.0f : (elu ? expf(x) : 0.0f); }

__device__ __forceinline__ void matvec16(const float* __restrict__ w /*[16,16] LDS*/, const float (&x)[16], float (&y)[16]) {
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += w[o * 16 + c] * x[c];
    y[o] = s;
  }
}
__device__ __forceinline__ void matvec16_t(const float* __restrict__ w, const float (&d)[16], float (&x)[16], bool add) {
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    float s = add ? x[c] : 0.0f;
#pragma unroll
    for (int o = 0; o < 16; ++o) s += w[o * 16 + c] * d[o];
    x[c] = s;
  }
}

__global__ __launch_bounds__(256) void ray_head_kernel(RayHeadArgs A) {
  extern __shared__ __attribute__((aligned(16))) float rh_smem[];
  const int S = A.S, i = threadIdx.x, ray = blockIdx.x;
  const bool live = i < S;
  const size_t n = (size_t)ray * S + (live ? i : 0);
  float* w_lds = rh_smem;                 // wq | wk | wv | wfc | wo0 (5 x 256), ln_w, ln_b, bo0, wo2 (4 x 16), bo2
  float* qs = w_lds + 5 * 256 + 4 * 16 + 16;  // [S][16]  q / 2
  float* ks = qs + S * 16;
  float* vs = ks + S * 16;
  float* dos = vs + S * 16;               // dL/dO
  float* ml = dos + S * 16;               // [S][4][2] row max / row sum
  float* dd = ml + S * 8;                 // [S][4]   D_i = <dO_i, O_i>
  float* vq = dd + S * 4;                 // [S] 1 = the query attends (n_valid > 1)
  for (int k = i; k < 256; k += blockDim.x) {
    w_lds[k] = A.wq[k];
    w_lds[256 + k] = A.wk[k];
    w_lds[512 + k] = A.wv[k];
    w_lds[768 + k] = A.wfc[k];
    w_lds[1024 + k] = A.wo0[k];
  }
  if (i < 16) {
    w_lds[1280 + i] = A.ln_w[i];
    w_lds[1296 + i] = A.ln_b[i];
    w_lds[1312 + i] = A.bo0[i];
    w_lds[1328 + i] = A.wo2[i];
  }
  if (i == 0) w_lds[1344] = A.bo2[0];
  __syncthreads();
  const float *wq = w_lds, *wk = w_lds + 256, *wv = w_lds + 512, *wfc = w_lds + 768, *wo0 = w_lds + 1024;
  const float *ln_w = w_lds + 1280, *ln_b = w_lds + 1296, *bo0 = w_lds + 1312, *wo2 = w_lds + 1328;

  // ---- forward
  float za[16], a0[16], q[16], kv[16], vv[16];
  float n_valid = 0.0f;
  if (live) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      za[c] = A.za[n * 16 + c];
      a0[c] = act_fwd(za[c], A.elu) + (A.table ? A.table[i * 16 + c] : 0.0f);
    }
    for (int v = 0; v < A.n_views; ++v) n_valid += A.cond[n * A.cond_stride + A.cond_dim - A.n_views + v];
    matvec16(wq, a0, q);
    matvec16(wk, a0, kv);
    matvec16(wv, a0, vv);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      qs[i * 16 + c] = 0.5f * q[c];
      ks[i * 16 + c] = kv[c];
      vs[i * 16 + c] = vv[c];
    }
    vq[i] = n_valid > 1.0f ? 1.0f : 0.0f;
  }
  __syncthreads();
  const bool attends = n_valid > 1.0f;
  float o[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) o[c] = 0.0f;
  if (live) {
    for (int h = 0; h < 4; ++h) {
      const float q0 = 0.5f * q[4 * h], q1 = 0.5f * q[4 * h + 1], q2 = 0.5f * q[4 * h + 2], q3 = 0.5f * q[4 * h + 3];
      float m = -3.0e38f;
      for (int j = 0; j < S; ++j) {
        const float* kj = ks + j * 16 + 4 * h;
        const float s = attends ? q0 * kj[0] + q1 * kj[1] + q2 * kj[2] + q3 * kj[3] : -1e9f;
        m = fmaxf(m, s);
      }
      float lsum = 0.0f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
      for (int j = 0; j < S; ++j) {
        const float* kj = ks + j * 16 + 4 * h;
        const float* vj = vs + j * 16 + 4 * h;
        const float s = attends ? q0 * kj[0] + q1 * kj[1] + q2 * kj[2] + q3 * kj[3] : -1e9f;
        const float p = expf(s - m);
        lsum += p;
        o0 += p * vj[0], o1 += p * vj[1], o2 += p * vj[2], o3 += p * vj[3];
      }
      const float il = 1.0f / lsum;
      o[4 * h] = o0 * il, o[4 * h + 1] = o1 * il, o[4 * h + 2] = o2 * il, o[4 * h + 3] = o3 * il;
      ml[(i * 4 + h) * 2] = m;
      ml[(i * 4 + h) * 2 + 1] = lsum;
    }
  }
  float u[16], xhat[16], y[16], t_pre[16], tt[16];
  float rstd = 0.0f, s_pre = 0.0f;
  if (live) {
    matvec16(wfc, o, u);
    float mean = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      u[c] += a0[c];
      mean += u[c];
    }
    mean *= 1.0f / 16.0f;
    float var = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) var += (u[c] - mean) * (u[c] - mean);
    rstd = 1.0f / sqrtf(var * (1.0f / 16.0f) + 1e-6f);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      xhat[c] = (u[c] - mean) * rstd;
      y[c] = xhat[c] * ln_w[c] + ln_b[c];
    }
    matvec16(wo0, y, t_pre);
    s_pre = w_lds[1344];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      t_pre[c] += bo0[c];
      tt[c] = act_fwd(t_pre[c], A.elu);
      s_pre += wo2[c] * tt[c];
    }
  }

  // ---- backward: density head, LayerNorm, fc
  float du[16], dq[16], dk[16], dv[16];
  if (live) {
    float ds = s_pre > 0.0f ? A.g_sigma[n] : 0.0f;
    if (A.maskfill && n_valid < 1.0f) ds = 0.0f;
    A.ds_pre[n] = ds;
    float dt_pre[16], dy[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) dt_pre[c] = ds * wo2[c] * act_grad(t_pre[c], A.elu);
    matvec16_t(wo0, dt_pre, dy, false);
    float m1 = 0.0f, m2 = 0.0f;
    float dxh[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      dxh[c] = dy[c] * ln_w[c];
      m1 += dxh[c];
      m2 += dxh[c] * xhat[c];
    }
    m1 *= 1.0f / 16.0f;
    m2 *= 1.0f / 16.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) du[c] = rstd * (dxh[c] - m1 - xhat[c] * m2);
    float d_o[16];
    matvec16_t(wfc, du, d_o, false);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      dos[i * 16 + c] = d_o[c];
      A.t[n * 16 + c] = tt[c];
      A.dt_pre[n * 16 + c] = dt_pre[c];
      A.y[n * 16 + c] = y[c];
      A.dy[n * 16 + c] = dy[c];
      A.dyx[n * 16 + c] = dy[c] * xhat[c];
      A.du[n * 16 + c] = du[c];
      A.o_att[n * 16 + c] = o[c];
      A.a0[n * 16 + c] = a0[c];
    }
#pragma unroll
    for (int h = 0; h < 4; ++h)
      dd[i * 4 + h] = d_o[4 * h] * o[4 * h] + d_o[4 * h + 1] * o[4 * h + 1] + d_o[4 * h + 2] * o[4 * h + 2] + d_o[4 * h + 3] * o[4 * h + 3];
  }
  __syncthreads();
  // ---- attention backward.  As a query (row i): dq_i = 1/2 sum_j dS_ij k_j;  as a key (column i): dk_i = 1/2 sum_r dS_ri q_r,
  //      dv_i = sum_r P_ri dO_r, with dS = P (dP - D), dP_rj = <dO_r, v_j>; rows that do not attend have constant scores
  //      (no dS) but still average the values (P = 1/S).
  if (live) {
    for (int h = 0; h < 4; ++h) {
      float a0_ = 0.f, a1_ = 0.f, a2_ = 0.f, a3_ = 0.f;
      if (attends) {
        const float q0 = 0.5f * q[4 * h], q1 = 0.5f * q[4 * h + 1], q2 = 0.5f * q[4 * h + 2], q3 = 0.5f * q[4 * h + 3];
        const float m = ml[(i * 4 + h) * 2], il = 1.0f / ml[(i * 4 + h) * 2 + 1], di = dd[i * 4 + h];
        const float* dor = dos + i * 16 + 4 * h;
        for (int j = 0; j < S; ++j) {
          const float* kj = ks + j * 16 + 4 * h;
          const float* vj = vs + j * 16 + 4 * h;
          const float p = expf(q0 * kj[0] + q1 * kj[1] + q2 * kj[2] + q3 * kj[3] - m) * il;
          const float dsij = p * (dor[0] * vj[0] + dor[1] * vj[1] + dor[2] * vj[2] + dor[3] * vj[3] - di);
          a0_ += dsij * kj[0], a1_ += dsij * kj[1], a2_ += dsij * kj[2], a3_ += dsij * kj[3];
        }
      }
      dq[4 * h] = 0.5f * a0_, dq[4 * h + 1] = 0.5f * a1_, dq[4 * h + 2] = 0.5f * a2_, dq[4 * h + 3] = 0.5f * a3_;
      float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
      const float* ki = ks + i * 16 + 4 * h;
      const float* vi = vs + i * 16 + 4 * h;
      const float unif = 1.0f / (float)S;
      for (int r = 0; r < S; ++r) {
        const float* qr = qs + r * 16 + 4 * h;
        const float* dor = dos + r * 16 + 4 * h;
        float p, dsri = 0.0f;
        if (vq[r] > 0.0f) {
          p = expf(qr[0] * ki[0] + qr[1] * ki[1] + qr[2] * ki[2] + qr[3] * ki[3] - ml[(r * 4 + h) * 2]) / ml[(r * 4 + h) * 2 + 1];
          dsri = p * (dor[0] * vi[0] + dor[1] * vi[1] + dor[2] * vi[2] + dor[3] * vi[3] - dd[r * 4 + h]);
        } else {
          p = unif;
        }
        k0 += dsri * qr[0], k1 += dsri * qr[1], k2 += dsri * qr[2], k3 += dsri * qr[3];  // (qs already holds q / 2)
        v0 += p * dor[0], v1 += p * dor[1], v2 += p * dor[2], v3 += p * dor[3];
      }
      dk[4 * h] = k0, dk[4 * h + 1] = k1, dk[4 * h + 2] = k2, dk[4 * h + 3] = k3;
      dv[4 * h] = v0, dv[4 * h + 1] = v1, dv[4 * h + 2] = v2, dv[4 * h + 3] = v3;
    }
    float da0[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) da0[c] = du[c];  // the residual
    matvec16_t(wq, dq, da0, true);
    matvec16_t(wk, dk, da0, true);
    matvec16_t(wv, dv, da0, true);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      A.dza[n * 16 + c] = da0[c] * act_grad(za[c], A.elu);
      A.dq[n * 16 + c] = dq[c];
      A.dk[n * 16 + c] = dk[c];
      A.dv[n * 16 + c] = dv[c];
    }
  }
}

// ------------------------------------------------------------------ the pass
// workspace, floats per sample
enum {
  WS_ENC = 0,                  // 64
  WS_DIRS = WS_ENC + 64,       // 4
  WS_FILM = WS_DIRS + 4,       // 128
  WS_Z = WS_FILM + 128,        // 6 x 128 trunk pre-activations (before the FiLM multiplier)
  WS_H = WS_Z + 6 * 128,       // 6 x 128 trunk outputs
  WS_FEAT = WS_H + 6 * 128,    // 128
  WS_HV = WS_FEAT + 128,       // 64  relu(views)
  WS_ZR = WS_HV + 64,          // 4   rgb pre-sigmoid
  WS_ZA = WS_ZR + 4,           // 16  alpha pre-activation
  WS_G = WS_ZA + 16,           // 128 running gradient of the trunk activation
  WS_DZ = WS_G + 128,          // 128
  WS_DFILM = WS_DZ + 128,      // 128
  WS_DHV = WS_DFILM + 128,     // 64
  WS_DZR = WS_DHV + 64,        // 4
  WS_SMALL = WS_DZR + 4,       // 13 x 16: dza a0 dq dk dv o_att du y dy dyx t dt_pre | ds_pre (1, padded to 16)
  WS_FLOATS = WS_SMALL + 13 * 16
};

extern "C" int64_t mnerf_decoder_backward_workspace_bytes(int32_t n_rays, int32_t n_samples) {
  return (int64_t)n_rays * n_samples * WS_FLOATS * (int64_t)sizeof(float);
}

extern "C" int mnerf_decoder_backward(const mnerf_decoder_train* D, int32_t n_rays, int32_t n_samples, const float* x_ndc,
                                      const float* dirs, const float* cond, int32_t cond_stride, const float* g_rgb_s,
                                      const float* g_sigma, float* g_cond, void* workspace, void* stream) {
  MNERF_REQUIRE(D, MNERF_E_NULL, "mnerf_decoder_backward: NULL parameter struct");
  MNERF_REQUIRE(n_rays >= 0 && n_samples >= 1 && n_samples <= 256, MNERF_E_RANGE, "mnerf_decoder_backward: n_rays=%d S=%d (S <= 256)",
                n_rays, n_samples);
  if (n_rays == 0) return MNERF_OK;  // an empty chunk contributes nothing (its buffers may be NULL)
  MNERF_REQUIRE(x_ndc && dirs && cond && g_rgb_s && g_sigma && workspace, MNERF_E_NULL, "mnerf_decoder_backward: NULL argument");
  MNERF_REQUIRE(D->n_trunk == 6 && D->net_width == 128, MNERF_E_UNSUPPORTED, "mnerf_decoder_backward: %d trunk layers of width %d (6 x 128)",
                D->n_trunk, D->net_width);
  MNERF_REQUIRE(D->L_3D >= 0 && 3 + 6 * D->L_3D <= 63, MNERF_E_RANGE, "mnerf_decoder_backward: L_3D=%d", D->L_3D);
  MNERF_REQUIRE(D->skip_layer >= -1 && D->skip_layer <= 4, MNERF_E_RANGE, "mnerf_decoder_backward: skip_layer=%d", D->skip_layer);
  MNERF_REQUIRE(D->cond_dim >= D->n_views && cond_stride >= D->cond_dim, MNERF_E_RANGE, "mnerf_decoder_backward: cond_dim=%d stride=%d",
                D->cond_dim, cond_stride);
  for (int k = 0; k < MNERF_DEC_TENSORS; ++k) MNERF_REQUIRE(D->w[k], MNERF_E_NULL, "mnerf_decoder_backward: parameter tensor %d is NULL", k);
  MNERF_REQUIRE(!D->raytrans_posenc || D->raytrans_table, MNERF_E_NULL, "mnerf_decoder_backward: raytrans_posenc without its table");
  hipStream_t st = (hipStream_t)stream;
  const int N = n_rays * n_samples, S = n_samples, E = 3 + 6 * D->L_3D, W = 128, Dc = D->cond_dim;
  float* ws = (float*)workspace;
  auto at = [&](int off) { return ws + (size_t)off * N; };  // planes: one [N, width] array per entry
  float *enc = at(WS_ENC), *dirs_s = at(WS_DIRS), *film = at(WS_FILM), *feat = at(WS_FEAT), *hv = at(WS_HV), *zr = at(WS_ZR),
        *za = at(WS_ZA), *g = at(WS_G), *dz = at(WS_DZ), *dfilm = at(WS_DFILM), *dhv = at(WS_DHV), *dzr = at(WS_DZR);
  auto zl = [&](int i) { return at(WS_Z) + (size_t)i * W * N; };
  auto hl = [&](int i) { return at(WS_H) + (size_t)i * W * N; };  // output of trunk layer i
  float* small = at(WS_SMALL);
  auto sm = [&](int k) { return small + (size_t)k * 16 * N; };
  const float* const* w = D->w;
  float* const* gw = D->g;
  const long long nW = (long long)N * W;

  // ================= forward, everything kept
  hipLaunchKernelGGL(encode_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, S, D->L_3D, D->legacy_coord, x_ndc, dirs, enc, dirs_s);
  linear_fwd(st, cond, cond_stride, w[MNERF_DT_BIAS_W], Dc, w[MNERF_DT_BIAS_B], film, W, N, W, Dc);
  for (int i = 0; i < 6; ++i) {
    const float* wi = w[MNERF_DT_PTS_W0 + 2 * i];
    const float* bi = w[MNERF_DT_PTS_W0 + 2 * i + 1];
    if (i == 0) {
      linear_fwd(st, enc, 64, wi, E, bi, zl(0), W, N, W, E);
    } else if (i - 1 == D->skip_layer) {  // input = [enc, h_{i-1}]
      linear_fwd(st, enc, 64, wi, E + W, bi, zl(i), W, N, W, E);
      linear_fwd(st, hl(i - 1), W, wi + E, E + W, nullptr, zl(i), W, N, W, W, true);
    } else {
      linear_fwd(st, hl(i - 1), W, wi, W, bi, zl(i), W, N, W, W);
    }
    hipLaunchKernelGGL(film_relu_kernel, dim3(ew_grid(nW)), dim3(256), 0, st, nW, zl(i), film, hl(i));
  }
  const float* hcur = hl(5);
  linear_fwd(st, hcur, W, w[MNERF_DT_ALPHA_W], W, w[MNERF_DT_ALPHA_B], za, 16, N, 16, W);
  linear_fwd(st, hcur, W, w[MNERF_DT_FEAT_W], W, w[MNERF_DT_FEAT_B], feat, W, N, W, W);
  linear_fwd(st, feat, W, w[MNERF_DT_VIEWS_W], W + 3, w[MNERF_DT_VIEWS_B], dhv, 64, N, 64, W);  // (dhv as scratch for the pre-activation)
  linear_fwd(st, dirs_s, 4, w[MNERF_DT_VIEWS_W] + W, W + 3, nullptr, dhv, 64, N, 64, 3, true);
  hipLaunchKernelGGL(relu_kernel, dim3(ew_grid((long long)N * 64)), dim3(256), 0, st, (long long)N * 64, dhv, hv);
  linear_fwd(st, hv, 64, w[MNERF_DT_RGB_W], 64, w[MNERF_DT_RGB_B], zr, 4, N, 3, 64);

  // ================= backward: colour branch
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, zr, g_rgb_s, dzr);
  linear_bwd_weight(st, dzr, 4, hv, 64, gw[MNERF_DT_RGB_W], 64, N, 3, 64);
  colsum(st, dzr, 4, N, 3, gw[MNERF_DT_RGB_B]);
  linear_bwd_data(st, dzr, 4, w[MNERF_DT_RGB_W], 64, dhv, 64, N, 3, 64, false);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid((long long)N * 64)), dim3(256), 0, st, (long long)N * 64, dhv, hv, dhv);
  linear_bwd_weight(st, dhv, 64, feat, W, gw[MNERF_DT_VIEWS_W], W + 3, N, 64, W);
  if (gw[MNERF_DT_VIEWS_W]) linear_bwd_weight(st, dhv, 64, dirs_s, 4, gw[MNERF_DT_VIEWS_W] + W, W + 3, N, 64, 3);
  colsum(st, dhv, 64, N, 64, gw[MNERF_DT_VIEWS_B]);
  linear_bwd_data(st, dhv, 64, w[MNERF_DT_VIEWS_W], W + 3, dz, W, N, 64, W, false);  // d(feature) in dz
  linear_bwd_weight(st, dz, W, hcur, W, gw[MNERF_DT_FEAT_W], W, N, W, W);
  colsum(st, dz, W, N, W, gw[MNERF_DT_FEAT_B]);
  linear_bwd_data(st, dz, W, w[MNERF_DT_FEAT_W], W, g, W, N, W, W, false);  // g = dL/dh5, colour part

  // ================= backward: density branch (16 wide, per ray)
  RayHeadArgs R{};
  R.S = S, R.cond_stride = cond_stride, R.cond_dim = Dc, R.n_views = D->n_views, R.elu = D->raytrans_elu, R.maskfill = D->density_maskfill;
  R.za = za, R.cond = cond, R.table = D->raytrans_posenc ? D->raytrans_table : nullptr, R.g_sigma = g_sigma;
  R.wq = w[MNERF_DT_WQ], R.wk = w[MNERF_DT_WK], R.wv = w[MNERF_DT_WV], R.wfc = w[MNERF_DT_FC];
  R.ln_w = w[MNERF_DT_LN_W], R.ln_b = w[MNERF_DT_LN_B];
  R.wo0 = w[MNERF_DT_OA0_W], R.bo0 = w[MNERF_DT_OA0_B], R.wo2 = w[MNERF_DT_OA2_W], R.bo2 = w[MNERF_DT_OA2_B];
  R.dza = sm(0), R.a0 = sm(1), R.dq = sm(2), R.dk = sm(3), R.dv = sm(4), R.o_att = sm(5), R.du = sm(6), R.y = sm(7), R.dy = sm(8),
  R.dyx = sm(9), R.t = sm(10), R.dt_pre = sm(11), R.ds_pre = sm(12);
  {
    const int threads = (S + 63) & ~63;
    const size_t lds = (size_t)(5 * 256 + 4 * 16 + 16 + S * (4 * 16 + 8 + 4 + 1)) * sizeof(float);
    // (S = 256: 84 KiB — more than the 64 KiB of earlier CDNA parts; this library is gfx950-only, 160 KiB per workgroup)
    MNERF_REQUIRE(lds <= 160 * 1024, MNERF_E_RANGE, "mnerf_decoder_backward: S=%d needs %zu bytes of LDS per ray (limit 160 KiB)", S, lds);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)ray_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ray_head_kernel, dim3(n_rays), dim3(threads), lds, st, R);
  }
  linear_bwd_weight(st, R.ds_pre, 1, R.t, 16, gw[MNERF_DT_OA2_W], 16, N, 1, 16);
  colsum(st, R.ds_pre, 1, N, 1, gw[MNERF_DT_OA2_B]);
  linear_bwd_weight(st, R.dt_pre, 16, R.y, 16, gw[MNERF_DT_OA0_W], 16, N, 16, 16);
  colsum(st, R.dt_pre, 16, N, 16, gw[MNERF_DT_OA0_B]);
  colsum(st, R.dyx, 16, N, 16, gw[MNERF_DT_LN_W]);
  colsum(st, R.dy, 16, N, 16, gw[MNERF_DT_LN_B]);
  linear_bwd_weight(st, R.du, 16, R.o_att, 16, gw[MNERF_DT_FC], 16, N, 16, 16);
  linear_bwd_weight(st, R.dq, 16, R.a0, 16, gw[MNERF_DT_WQ], 16, N, 16, 16);
  linear_bwd_weight(st, R.dk, 16, R.a0, 16, gw[MNERF_DT_WK], 16, N, 16, 16);
  linear_bwd_weight(st, R.dv, 16, R.a0, 16, gw[MNERF_DT_WV], 16, N, 16, 16);
  linear_bwd_weight(st, R.dza, 16, hcur, W, gw[MNERF_DT_ALPHA_W], W, N, 16, W);
  colsum(st, R.dza, 16, N, 16, gw[MNERF_DT_ALPHA_B]);
  linear_bwd_data(st, R.dza, 16, w[MNERF_DT_ALPHA_W], W, g, W, N, 16, W, true);  // g += density part

  // ================= backward: FiLM trunk
  for (int i = 5; i >= 0; --i) {
    const float* wi = w[MNERF_DT_PTS_W0 + 2 * i];
    float* gwi = gw[MNERF_DT_PTS_W0 + 2 * i];
    hipLaunchKernelGGL(film_relu_bwd_kernel, dim3(ew_grid(nW)), dim3(256), 0, st, nW, g, hl(i), zl(i), film, dz, dfilm, i == 5 ? 1 : 0);
    colsum(st, dz, W, N, W, gw[MNERF_DT_PTS_W0 + 2 * i + 1]);
    if (i == 0) {
      linear_bwd_weight(st, dz, W, enc, 64, gwi, E, N, W, E);
    } else if (i - 1 == D->skip_layer) {
      linear_bwd_weight(st, dz, W, enc, 64, gwi, E + W, N, W, E);
      if (gwi) linear_bwd_weight(st, dz, W, hl(i - 1), W, gwi + E, E + W, N, W, W);
      linear_bwd_data(st, dz, W, wi + E, E + W, g, W, N, W, W, false);
    } else {
      linear_bwd_weight(st, dz, W, hl(i - 1), W, gwi, W, N, W, W);
      linear_bwd_data(st, dz, W, wi, W, g, W, N, W, W, false);
    }
  }
  linear_bwd_weight(st, dfilm, W, cond, cond_stride, gw[MNERF_DT_BIAS_W], Dc, N, W, Dc);
  colsum(st, dfilm, W, N, W, gw[MNERF_DT_BIAS_B]);
  if (g_cond) linear_bwd_data(st, dfilm, W, w[MNERF_DT_BIAS_W], Dc, g_cond, cond_stride, N, W, Dc, false);
  return mnerf_check_launch("mnerf_decoder_backward");
}
