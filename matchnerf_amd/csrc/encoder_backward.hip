// Backward of a GMFlow transformer layer around the window attention, for gfx950 (round 4):
//   mnerf_qkv_backward            the three bias-free 128 -> 128 projections (transformer.py:147-151)
//   mnerf_encoder_layer_backward  everything after the attention (transformer.py:176-185; the forward is K7, encoder_block.hip):
//                                 message = norm1(merge(attn)); [ffn:] message = norm2(mlp.2(GELU(mlp.0(cat[source, message]))));
//                                 out = source + message
// i.e. what `loss.backward()` does to models/gmflow/transformer.py:108-185 in the reference's training loop
// (/root/reference/coach.py:215-243), with mnerf_window_attention_backward in between.
//
// Like the decoder's backward (decoder_backward.hip) this is deliberately NOT one fused tile kernel: the chain is re-evaluated
// from the layer's inputs with every pre-activation kept in a workspace (3 970 floats per token), then walked backwards.  Every
// matrix product — Y = X W^T, dX = dY W, dW = dY^T X — is the strided MFMA GEMM of gemm_f32.hpp (fp32-grade split-bf16 by default for I, J >= 128, exact fp32 with MNERF_GEMM_MATH=f32; strided operands: the
// column halves of mlp.0's [1024, 256] weight cost nothing; split-K with float atomics for the weight gradients).  What is not a
// matrix product: LayerNorm forward / backward (a half-wave per 128-feature row), exact-erf GELU and its derivative.
// Parameters travel in torch's own layouts (mnerf_encoder_layer_train); parameter gradients are ACCUMULATED.
#include "gemm_f32.hpp"

#define EB_C 128
#define EB_H 1024

// ---------------------------------------------------------------------------------------------- LayerNorm(128), eps 1e-5
// xhat = (x - mean) * rstd (kept for the backward), y = xhat * gamma + beta (y may be NULL); one half-wave per row
__global__ __launch_bounds__(256) void eb_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ xhat, float* __restrict__ y,
                                                        float* __restrict__ rstd_out, long long n_rows, float eps) {
  const int c4 = threadIdx.x & 31;
  const float4 g = reinterpret_cast<const float4*>(gamma)[c4], b = reinterpret_cast<const float4*>(beta)[c4];
  for (long long row = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); row < n_rows; row += (long long)gridDim.x * 8) {
    const float4 v = reinterpret_cast<const float4*>(x + row * EB_C)[c4];
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s * (1.0f / EB_C);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / EB_C) + eps);
    const float4 h = make_float4(dx * rstd, dy * rstd, dz * rstd, dw * rstd);
    reinterpret_cast<float4*>(xhat + row * EB_C)[c4] = h;
    if (y) reinterpret_cast<float4*>(y + row * EB_C)[c4] = make_float4(h.x * g.x + b.x, h.y * g.y + b.y, h.z * g.z + b.z, h.w * g.w + b.w);
    if (c4 == 0) rstd_out[row] = rstd;
  }
}
// dx = rstd (t - mean(t) - xhat mean(t xhat)), t = dy gamma;  dgamma += sum_rows dy xhat, dbeta += sum_rows dy
__global__ __launch_bounds__(256) void eb_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        long long n_rows) {
  __shared__ float red[2][8][EB_C];
  const int c4 = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long row = (long long)blockIdx.x * 8 + sub; row < n_rows; row += (long long)gridDim.x * 8) {
    const float4 d = reinterpret_cast<const float4*>(dy + row * EB_C)[c4];
    const float4 h = reinterpret_cast<const float4*>(xhat + row * EB_C)[c4];
    const float4 t = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
    float a = (t.x + t.y) + (t.z + t.w);
    float b = (t.x * h.x + t.y * h.y) + (t.z * h.z + t.w * h.w);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      b += __shfl_xor(b, off, 64);
    }
    a *= (1.0f / EB_C);
    b *= (1.0f / EB_C);
    const float r = rstd[row];
    reinterpret_cast<float4*>(dx + row * EB_C)[c4] =
        make_float4(r * (t.x - a - h.x * b), r * (t.y - a - h.y * b), r * (t.z - a - h.z * b), r * (t.w - a - h.w * b));
    ag.x += d.x * h.x, ag.y += d.y * h.y, ag.z += d.z * h.z, ag.w += d.w * h.w;
    ab.x += d.x, ab.y += d.y, ab.z += d.z, ab.w += d.w;
  }
  if (!dgamma && !dbeta) return;
  reinterpret_cast<float4*>(&red[0][sub][0])[c4] = ag;
  reinterpret_cast<float4*>(&red[1][sub][0])[c4] = ab;
  __syncthreads();
  if (threadIdx.x < EB_C) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      sg += red[0][s][threadIdx.x];
      sb += red[1][s][threadIdx.x];
    }
    if (dgamma) atomicAdd(dgamma + threadIdx.x, sg);
    if (dbeta) atomicAdd(dbeta + threadIdx.x, sb);
  }
}

// ---------------------------------------------------------------------------------------------- GELU (exact erf, torch's default)
__global__ __launch_bounds__(256) void eb_gelu_kernel(long long n, const float* __restrict__ z, float* __restrict__ g) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float x = z[i];
    g[i] = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
  }
}
// dz = dg * gelu'(z) in place (dg <- dz):  gelu'(z) = Phi(z) + z phi(z)
__global__ __launch_bounds__(256) void eb_gelu_bwd_kernel(long long n, const float* __restrict__ z, float* __restrict__ dg) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float x = z[i];
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    dg[i] *= cdf + x * pdf;
  }
}
__global__ __launch_bounds__(256) void eb_add_kernel(long long n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = a[i] + (b ? b[i] : 0.0f);
}
static int eb_grid(long long n) { return (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256); }
static int eb_row_grid(long long rows) { return (int)((rows + 7) / 8 > 512 ? 512 : (rows + 7) / 8); }  // (each block ends with 256 float atomics)

// workspace, floats per token
enum {
  EW_M1 = 0,                 // 128  merge(attn), pre-LN; later d(merge output)
  EW_XH1 = EW_M1 + EB_C,     // 128  normalised norm1 input
  EW_N1 = EW_XH1 + EB_C,     // 128  norm1 output (second half of mlp.0's input)
  EW_DN1 = EW_N1 + EB_C,     // 128  gradient of the norm1 output
  EW_M2 = EW_DN1 + EB_C,     // 128  mlp.2 output, pre-LN
  EW_XH2 = EW_M2 + EB_C,     // 128
  EW_DM2 = EW_XH2 + EB_C,    // 128
  EW_Z1 = EW_DM2 + EB_C,     // 1024 mlp.0 output
  EW_G1 = EW_Z1 + EB_H,      // 1024 GELU output
  EW_DG = EW_G1 + EB_H,      // 1024 gradient of the GELU output, then of its input
  EW_RSTD = EW_DG + EB_H,    // 2    reciprocal standard deviations of the two norms
  EW_FLOATS = EW_RSTD + 2
};

extern "C" int64_t mnerf_encoder_layer_backward_workspace_bytes(int32_t n_tokens) {
  return n_tokens < 0 ? -1 : (int64_t)n_tokens * EW_FLOATS * (int64_t)sizeof(float);
}

// m1_saved, z1_saved / m2_saved (all or none; the latter two for FFN layers): merge's output before norm1, mlp.0's output before the
// GELU and mlp.2's output before norm2 as the training forward left them (mnerf_encoder_block_save) - the GEMMs that would
// re-evaluate them (merge; 2 x [N,1024] K = 128, [N,128] K = 1024: 26 of an FFN layer's 76 GFLOP at the DTU shape) are skipped
static int encoder_layer_backward_impl(const char* who, const mnerf_encoder_layer_train* L, const float* attn, const float* source,
                                       const float* g_out, const float* m1_saved, const float* z1_saved, const float* m2_saved,
                                       float* g_attn, float* g_source, int32_t n_tokens, void* workspace, void* stream) {
  MNERF_REQUIRE(L && attn && source && g_out && g_attn && g_source, MNERF_E_NULL, "%s: NULL argument", who);
  MNERF_REQUIRE(n_tokens >= 0, MNERF_E_RANGE, "%s: n_tokens=%d", who, n_tokens);
  if (n_tokens == 0) return MNERF_OK;
  MNERF_REQUIRE(workspace, MNERF_E_NULL, "%s: NULL workspace", who);
  MNERF_REQUIRE(L->w_merge && L->ln1_w && L->ln1_b, MNERF_E_NULL, "%s: NULL parameter", who);
  MNERF_REQUIRE(!L->ffn || (L->w_mlp0 && L->w_mlp2 && L->ln2_w && L->ln2_b), MNERF_E_NULL, "%s: NULL FFN parameter", who);
  MNERF_REQUIRE(mnerf_aligned16(attn) && mnerf_aligned16(source) && mnerf_aligned16(g_out) && mnerf_aligned16(g_attn) &&
                    mnerf_aligned16(g_source) && mnerf_aligned16(workspace) && mnerf_aligned16(L->ln1_w) && mnerf_aligned16(L->ln1_b),
                MNERF_E_ALIGN, "%s: buffers must be 16-byte aligned", who);
  hipStream_t st = (hipStream_t)stream;
  const int N = n_tokens;
  const long long nC = (long long)N * EB_C, nH = (long long)N * EB_H;
  float* ws = (float*)workspace;
  auto at = [&](int off) { return ws + (size_t)off * N; };
  float *m1 = at(EW_M1), *xh1 = at(EW_XH1), *n1 = at(EW_N1), *dn1 = at(EW_DN1), *m2 = at(EW_M2), *xh2 = at(EW_XH2), *dm2 = at(EW_DM2),
        *z1 = at(EW_Z1), *g1 = at(EW_G1), *dg = at(EW_DG), *rstd1 = at(EW_RSTD), *rstd2 = at(EW_RSTD) + N;
  if (z1_saved) z1 = const_cast<float*>(z1_saved), m2 = const_cast<float*>(m2_saved);  // (read only from here on)
  const float eps = 1e-5f;  // nn.LayerNorm's default (transformer.py:136, 144)

  // ================= forward, everything kept
  if (!m1_saved) linear_fwd(st, attn, EB_C, L->w_merge, EB_C, nullptr, m1, EB_C, N, EB_C, EB_C);
  // (the workspace's m1 later takes the gradient of the merge output: the saved tensor is only read)
  hipLaunchKernelGGL(eb_ln_fwd_kernel, dim3(eb_row_grid(N)), dim3(256), 0, st, m1_saved ? m1_saved : m1, L->ln1_w, L->ln1_b, xh1, n1, rstd1,
                     (long long)N, eps);
  const float* d_n1 = g_out;  // gradient of the norm1 output; without the FFN it is the layer's output gradient
  if (L->ffn) {
    if (!z1_saved) {
      linear_fwd(st, source, EB_C, L->w_mlp0, 2 * EB_C, nullptr, z1, EB_H, N, EB_H, EB_C);                 // cat[source, message]:
      linear_fwd(st, n1, EB_C, L->w_mlp0 + EB_C, 2 * EB_C, nullptr, z1, EB_H, N, EB_H, EB_C, true);        // two column halves
    }
    hipLaunchKernelGGL(eb_gelu_kernel, dim3(eb_grid(nH)), dim3(256), 0, st, nH, z1, g1);
    if (!z1_saved) linear_fwd(st, g1, EB_H, L->w_mlp2, EB_H, nullptr, m2, EB_C, N, EB_C, EB_H);
    hipLaunchKernelGGL(eb_ln_fwd_kernel, dim3(eb_row_grid(N)), dim3(256), 0, st, m2, L->ln2_w, L->ln2_b, xh2, (float*)nullptr, rstd2,
                       (long long)N, eps);
    // ================= backward: norm2, mlp.2, GELU, mlp.0
    hipLaunchKernelGGL(eb_ln_bwd_kernel, dim3(eb_row_grid(N)), dim3(256), 0, st, g_out, xh2, rstd2, L->ln2_w, dm2, L->g_ln2_w, L->g_ln2_b,
                       (long long)N);
    linear_bwd_weight(st, dm2, EB_C, g1, EB_H, L->g_w_mlp2, EB_H, N, EB_C, EB_H);
    linear_bwd_data(st, dm2, EB_C, L->w_mlp2, EB_H, dg, EB_H, N, EB_C, EB_H, false);
    hipLaunchKernelGGL(eb_gelu_bwd_kernel, dim3(eb_grid(nH)), dim3(256), 0, st, nH, z1, dg);
    linear_bwd_weight(st, dg, EB_H, source, EB_C, L->g_w_mlp0, 2 * EB_C, N, EB_H, EB_C);
    if (L->g_w_mlp0) linear_bwd_weight(st, dg, EB_H, n1, EB_C, L->g_w_mlp0 + EB_C, 2 * EB_C, N, EB_H, EB_C);
    hipLaunchKernelGGL(eb_add_kernel, dim3(eb_grid(nC)), dim3(256), 0, st, nC, g_out, (const float*)nullptr, g_source);  // residual
    linear_bwd_data(st, dg, EB_H, L->w_mlp0, 2 * EB_C, g_source, EB_C, N, EB_H, EB_C, true);
    linear_bwd_data(st, dg, EB_H, L->w_mlp0 + EB_C, 2 * EB_C, dn1, EB_C, N, EB_H, EB_C, false);
    d_n1 = dn1;
  } else {
    hipLaunchKernelGGL(eb_add_kernel, dim3(eb_grid(nC)), dim3(256), 0, st, nC, g_out, (const float*)nullptr, g_source);
  }
  // ================= backward: norm1, merge
  hipLaunchKernelGGL(eb_ln_bwd_kernel, dim3(eb_row_grid(N)), dim3(256), 0, st, d_n1, xh1, rstd1, L->ln1_w, m1 /* d(merge output) */,
                     L->g_ln1_w, L->g_ln1_b, (long long)N);
  linear_bwd_weight(st, m1, EB_C, attn, EB_C, L->g_w_merge, EB_C, N, EB_C, EB_C);
  linear_bwd_data(st, m1, EB_C, L->w_merge, EB_C, g_attn, EB_C, N, EB_C, EB_C, false);
  return mnerf_check_launch(who);
}

extern "C" int mnerf_encoder_layer_backward(const mnerf_encoder_layer_train* L, const float* attn, const float* source,
                                            const float* g_out, float* g_attn, float* g_source, int32_t n_tokens,
                                            void* workspace, void* stream) {
  return encoder_layer_backward_impl("mnerf_encoder_layer_backward", L, attn, source, g_out, nullptr, nullptr, nullptr, g_attn,
                                     g_source, n_tokens, workspace, stream);
}

extern "C" int mnerf_encoder_layer_backward_saved(const mnerf_encoder_layer_train* L, const float* attn, const float* source,
                                                  const float* g_out, const float* m1, const float* z1, const float* m2,
                                                  float* g_attn, float* g_source, int32_t n_tokens, void* workspace, void* stream) {
  const char* who = "mnerf_encoder_layer_backward_saved";
  MNERF_REQUIRE(L, MNERF_E_NULL, "%s: NULL argument", who);
  MNERF_REQUIRE(n_tokens == 0 || (m1 && mnerf_aligned16(m1)), MNERF_E_NULL, "%s: m1 NULL or not 16-byte aligned", who);
  MNERF_REQUIRE(n_tokens == 0 || !L->ffn || (z1 && m2 && mnerf_aligned16(z1) && mnerf_aligned16(m2)), MNERF_E_NULL,
                "%s: z1 / m2 NULL or not 16-byte aligned", who);
  return encoder_layer_backward_impl(who, L, attn, source, g_out, m1, L->ffn ? z1 : nullptr, L->ffn ? m2 : nullptr, g_attn, g_source,
                                     n_tokens, workspace, stream);
}

// q = x_q Wq^T, k = x_kv Wk^T, v = x_kv Wv^T  ->  g_xq = g_q Wq,  g_xkv = g_k Wk + g_v Wv,  dW* += g*^T x*
extern "C" int mnerf_qkv_backward(const float* w_q, const float* w_k, const float* w_v, const float* x_q, const float* x_kv,
                                  const float* g_q, const float* g_k, const float* g_v, float* g_xq, float* g_xkv, float* gw_q,
                                  float* gw_k, float* gw_v, int32_t n_tokens, void* stream) {
  const char* who = "mnerf_qkv_backward";
  MNERF_REQUIRE(w_q && w_k && w_v && x_q && x_kv && g_q && g_k && g_v && g_xq && g_xkv, MNERF_E_NULL, "%s: NULL argument", who);
  MNERF_REQUIRE(g_xq != g_xkv, MNERF_E_RANGE, "%s: g_xq and g_xkv must be different buffers (both are overwritten)", who);
  MNERF_REQUIRE(n_tokens >= 0, MNERF_E_RANGE, "%s: n_tokens=%d", who, n_tokens);
  if (n_tokens == 0) return MNERF_OK;
  hipStream_t st = (hipStream_t)stream;
  const int N = n_tokens;
  linear_bwd_data(st, g_q, EB_C, w_q, EB_C, g_xq, EB_C, N, EB_C, EB_C, false);
  linear_bwd_data(st, g_k, EB_C, w_k, EB_C, g_xkv, EB_C, N, EB_C, EB_C, false);
  linear_bwd_data(st, g_v, EB_C, w_v, EB_C, g_xkv, EB_C, N, EB_C, EB_C, true);
  linear_bwd_weight(st, g_q, EB_C, x_q, EB_C, gw_q, EB_C, N, EB_C, EB_C);
  linear_bwd_weight(st, g_k, EB_C, x_kv, EB_C, gw_k, EB_C, N, EB_C, EB_C);
  linear_bwd_weight(st, g_v, EB_C, x_kv, EB_C, gw_v, EB_C, N, EB_C, EB_C);
  return mnerf_check_launch(who);
}

// Test hook: the strided GEMM the backward kernels are built on (gemm_f32.hpp), so that its arithmetic can be pinned on its own.
extern "C" int mnerf_debug_gemm(const float* a, int64_t sa_i, int64_t sa_k, const float* b, int64_t sb_k, int64_t sb_j, float* c,
                                int64_t sc_i, const float* bias, int32_t I, int32_t J, int32_t K, int32_t mode, int32_t math,
                                void* stream) {
  const char* who = "mnerf_debug_gemm";
  MNERF_REQUIRE(a && b && c, MNERF_E_NULL, "%s: NULL operand", who);
  MNERF_REQUIRE(I >= 1 && J >= 1 && K >= 1, MNERF_E_RANGE, "%s: I=%d J=%d K=%d", who, I, J, K);
  MNERF_REQUIRE(mode >= 0 && mode <= 2 && math >= 0 && math <= 2, MNERF_E_RANGE, "%s: mode=%d math=%d", who, mode, math);
  gemm_with((hipStream_t)stream, a, sa_i, sa_k, b, sb_k, sb_j, c, sc_i, bias, I, J, K, mode, math);
  return mnerf_check_launch(who);
}
