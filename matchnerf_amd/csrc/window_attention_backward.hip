// K6 backward — gradients of the GMFlow (shifted-)window attention for gfx950, flash style: no [windows, L_w, L_w] tensor.
//
// Replaces what `loss.backward()` does to models/gmflow/transformer.py:46-105 (single_head_split_window_attention: roll,
// window split, scores = q k^T / sqrt(C) (+ the -100 wrap-region mask), softmax, @ v, window merge, roll back) in the
// reference's training loop (/root/reference/coach.py:215-243).  Rounds 1-3 re-evaluated that op chain in torch under autograd:
// a [24, 1280, 1280] score tensor (157 MB) per attention call, written and read half a dozen times.
//
// For one window, with P = softmax(S), S = Q K^T * scale + mask and dO the gradient of O = P V:
//     D_i  = <dO_i, O_i>                        (wa_bwd_rowdot_kernel)
//     dV_j = sum_i P_ij dO_i                    dP_ij = <dO_i, V_j>
//     dS_ij = P_ij (dP_ij - D_i)
//     dQ_i = scale * sum_j dS_ij K_j            dK_j = scale * sum_i dS_ij Q_i
// P is recomputed tile by tile from the row statistics (max, sum of exponentials) of a first pass, never stored.
//   wa_bwd_dq_kernel   one workgroup per (window, 64-query tile): pass 1 over all key tiles -> row statistics (also written
//                      out for the second kernel), pass 2 -> dQ of its 64 queries.  Scores are held TRANSPOSED (S^T = K Q^T:
//                      lane = query, registers = keys), so the row reductions are in-lane + one cross-half shuffle, and dS^T in
//                      its accumulator registers IS the B operand of dQ^T = K^T dS^T (the decoder's accumulator -> operand
//                      chain: a K-step may pair any two keys as long as both operands use the same pair).
//   wa_bwd_dkv_kernel  one workgroup per (window, 64-key tile): loops over the query tiles with S = Q K^T (lane = key,
//                      registers = queries), so that P and dS in their accumulators are the B operands of
//                      dV^T = dO^T P and dK^T = Q^T dS.
// Three forms of the two kernels (MNERF_WA_BWD_MATH): the split 16-bit ones further down — "f16x3" (two range-managed fp16 terms
// per operand, three term products) and "bf16x6" (three bf16 terms, six term products), both fp32-grade on the 16-bit matrix
// instructions — and this one ("f32"), where every product is exact fp32 on v_mfma_f32_32x32x2_f32; its tiles are staged in LDS
// channel-major ([channel][row], row stride 65 floats: conflict-free for both operand roles).
// Roll, window split / merge and the wrap-region mask are index arithmetic (wa_common.hpp: win_token), as in the forward.
// No atomics: every output row is written by exactly one workgroup; results are deterministic.
#include <stdlib.h>
#include <string.h>

#include "wa_common.hpp"

#define WB_T 64             // rows (queries or keys) per tile
#define WB_LD 65            // LDS row stride of a [channel][row] tile
#define WB_TILE_FLOATS (WA_C * WB_LD)

// accumulator register r of lane (n, half) of a 32x32 block holds row f(r, half) of column n
__device__ __forceinline__ int wb_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 wb_mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// D[token] = <dO[token], O[token]>;  n2[token] = ||dO[token]||^2 (if n2: the f16x3 form's per-query gain)
__global__ __launch_bounds__(256) void wa_bwd_rowdot_kernel(const float* __restrict__ g_out, const float* __restrict__ out,
                                                            float* __restrict__ d, float* __restrict__ n2, long long n_tokens) {
  const long long tok = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);  // 32 lanes per token, 4 channels per lane
  if (tok >= n_tokens) return;
  const int c4 = threadIdx.x & 31;
  const float4 a = reinterpret_cast<const float4*>(g_out + tok * WA_C)[c4];
  const float4 b = reinterpret_cast<const float4*>(out + tok * WA_C)[c4];
  float s = (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
  float t = (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64), t += __shfl_xor(t, off, 64);
  if (c4 == 0) {
    d[tok] = s;
    if (n2) n2[tok] = t;
  }
}

// A tile = 64 rows (window-local indices i0 .. i0+63 of window (wy, wx)) x 128 channels, staged in LDS as [channel][row]; rows
// beyond the window are zero.  Loading is split in two so that the global loads of the NEXT tile are in flight while the
// current one is multiplied (one wave per SIMD, 512 registers: the 8 float4 per tile are free):
//   wb_fetch   global -> registers: a half-wave (32 lanes x 16 B) reads one row, 512 contiguous bytes
//   wb_store   registers -> LDS (transposing), after the barrier that retires the previous tile
//   wb_tokens  token id (-1 beyond the window) and wrap region of the tile's rows
struct WbTileRegs {
  float4 v[WB_T / 8];
};
__device__ __forceinline__ void wb_fetch(WbTileRegs& r, const float* __restrict__ src_seq, const WinGeom& G, int wy, int wx, int i0,
                                         int tid) {
  const int c4 = tid & 31;
#pragma unroll
  for (int k = 0; k < WB_T / 8; ++k) {
    const int li = i0 + (tid >> 5) + 8 * k;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (li < G.Lw) {
      int region;
      const int tok = win_token(G, wy, wx, li, region);
      v = reinterpret_cast<const float4*>(src_seq + (size_t)tok * WA_C)[c4];
    }
    r.v[k] = v;
  }
}
__device__ __forceinline__ void wb_store(float* __restrict__ dst, const WbTileRegs& r, int tid) {
  const int c4 = tid & 31;
#pragma unroll
  for (int k = 0; k < WB_T / 8; ++k) {
    const int row = (tid >> 5) + 8 * k;
    dst[(4 * c4 + 0) * WB_LD + row] = r.v[k].x;
    dst[(4 * c4 + 1) * WB_LD + row] = r.v[k].y;
    dst[(4 * c4 + 2) * WB_LD + row] = r.v[k].z;
    dst[(4 * c4 + 3) * WB_LD + row] = r.v[k].w;
  }
}
__device__ __forceinline__ void wb_tokens(const WinGeom& G, int wy, int wx, int i0, int tid, int* tok_lds, int* reg_lds) {
  if (tid < WB_T) {
    int region = 0;
    const int li = i0 + tid;
    const int tok = li < G.Lw ? win_token(G, wy, wx, li, region) : -1;
    tok_lds[tid] = tok;
    reg_lds[tid] = region;
  }
}
__device__ __forceinline__ void wb_load_tile(float* __restrict__ dst, const float* __restrict__ src_seq, const WinGeom& G, int wy,
                                             int wx, int i0, int tid, int* tok_lds, int* reg_lds) {
  WbTileRegs r;
  wb_fetch(r, src_seq, G, wy, wx, i0, tid);
  wb_store(dst, r, tid);
  if (tok_lds) wb_tokens(G, wy, wx, i0, tid, tok_lds, reg_lds);
}

// acc[rows of A-block ra] [cols of B-block cb] = sum_ch A[ch][ra*32 + m] * B[ch][cb*32 + n]   (both tiles [channel][row])
__device__ __forceinline__ f32x16 wb_tile_product(const float* __restrict__ a_t, int ra, const float* __restrict__ b_t, int cb, int lane) {
  const int m = lane & 31, kk = lane >> 5;
  const float* ap = a_t + kk * WB_LD + ra * 32 + m;
  const float* bp = b_t + kk * WB_LD + cb * 32 + m;
  f32x16 acc = (f32x16)(0.0f);
#pragma unroll 8
  for (int t = 0; t < WA_C / 2; ++t) acc = wb_mfma(ap[2 * t * WB_LD], bp[2 * t * WB_LD], acc);
  return acc;
}

// out^T[channel][col] += sum over the 32 rows held in `s`'s registers:  X[channel][row] * s[row][col]
// (x_t: [channel][row] tile, rows of block rb; s: accumulator block whose registers are those rows)
__device__ __forceinline__ void wb_chain_product(f32x16 (&out)[4], const float* __restrict__ x_t, int rb, const f32x16& s, int lane) {
  const int m = lane & 31, kk = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* xp = x_t + m * WB_LD + rb * 32 + wb_row(r, kk);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) out[mb] = wb_mfma(xp[mb * 32 * WB_LD], s[r], out[mb]);
  }
}

struct WaBwdArgs {
  const float *q, *k, *v, *g_out;
  float *g_q, *g_k, *g_v;
  float *row_m, *row_l;   // [batch * h * w] row statistics (log2 domain maximum, sum of exponentials)
  const float* row_d;     // [batch * h * w] <dO, O>
  const float* row_n2;    // [batch * h * w] ||dO||^2                                   (f16x3 form)
  const unsigned* absmax; // bit patterns of max|q|, max|k|, max|v|, max|dO|            (f16x3 form)
  WinGeom G;
  int do_shift;
  float scale;            // 1 / sqrt(C)
};

#define WB_LOG2E 1.4426950408889634f
// score of (query, key) in the log2 domain: (q.k * scale + mask) * log2(e); keys beyond the window: -inf
__device__ __forceinline__ float wb_score(float dot, float scale, bool masked, bool valid) {
  const float s = (dot * scale + (masked ? -100.0f : 0.0f)) * WB_LOG2E;
  return valid ? s : -INFINITY;
}

// ---------------------------------------------------------------------------------------------------------------- dQ
__global__ __launch_bounds__(256, 1) void wa_bwd_dq_kernel(WaBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float wb_smem[];
  float* q_t = wb_smem;                       // [C][65] this workgroup's query tile
  float* do_t = q_t + WB_TILE_FLOATS;         // dO of the same rows
  float* k_t = do_t + WB_TILE_FLOATS;         // current key tile
  float* v_t = k_t + WB_TILE_FLOATS;
  int* q_tok = reinterpret_cast<int*>(v_t + WB_TILE_FLOATS);
  int* q_reg = q_tok + WB_T;
  int* k_tok = q_reg + WB_T;
  int* k_reg = k_tok + WB_T;
  float* st_m = reinterpret_cast<float*>(k_reg + WB_T);  // [2 key halves][64 queries] partial statistics, then the final ones in [0]
  float* st_l = st_m + 2 * WB_T;
  float* st_d = st_l + 2 * WB_T;

  const WinGeom& G = A.G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = wave >> 1, wk = wave & 1;     // this wave: queries [32 wq, +32) x keys [32 wk, +32) of a 64 x 64 tile
  const int n = lane & 31, half = lane >> 5;
  const int n_tiles = (G.Lw + WB_T - 1) / WB_T;
  const int qt = blockIdx.x, win = blockIdx.y, seq = blockIdx.z;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const size_t seq_off = (size_t)seq * G.h * G.w;
  const float* qs = A.q + seq_off * WA_C;
  const float* ks = A.k + seq_off * WA_C;
  const float* vs = A.v + seq_off * WA_C;
  const float* gos = A.g_out + seq_off * WA_C;

  wb_load_tile(q_t, qs, G, wy, wx, qt * WB_T, tid, q_tok, q_reg);
  wb_load_tile(do_t, gos, G, wy, wx, qt * WB_T, tid, nullptr, nullptr);
  __syncthreads();
  const int my_q = wq * 32 + n;                // this lane's query (tile-local)
  const int my_qreg = q_reg[my_q];
  const bool q_ok = q_tok[my_q] >= 0;
  if (tid < WB_T) st_d[tid] = q_tok[tid] >= 0 ? A.row_d[seq_off + q_tok[tid]] : 0.0f;

  // ---- pass 1: row statistics over this wave's share of the keys (online softmax), merged across the waves afterwards
  float run_m = -INFINITY, run_l = 0.0f;
  WbTileRegs nk, nv;  // the next key / value tile, in flight
  wb_fetch(nk, ks, G, wy, wx, 0, tid);
  for (int kt = 0; kt < n_tiles; ++kt) {
    __syncthreads();  // the previous tile has been consumed
    wb_store(k_t, nk, tid);
    wb_tokens(G, wy, wx, kt * WB_T, tid, k_tok, k_reg);
    __syncthreads();
    if (kt + 1 < n_tiles) wb_fetch(nk, ks, G, wy, wx, (kt + 1) * WB_T, tid);
    const f32x16 st = wb_tile_product(k_t, wk, q_t, wq, lane);  // S^T block: rows = keys, columns = queries
    float sc[16], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = wk * 32 + wb_row(r, half);
      sc[r] = wb_score(st[r], A.scale, A.do_shift && k_reg[key] != my_qreg, k_tok[key] >= 0);
      mx = fmaxf(mx, sc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(run_m, mx);
    float sum = 0.0f;
    if (m_new > -INFINITY) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(sc[r] - m_new);
      sum += __shfl_xor(sum, 32, 64);
      run_l = run_l * (run_m > -INFINITY ? __builtin_amdgcn_exp2f(run_m - m_new) : 0.0f) + sum;
      run_m = m_new;
    }
  }
  if (half == 0) {
    st_m[wk * WB_T + my_q] = run_m;
    st_l[wk * WB_T + my_q] = run_l;
  }
  __syncthreads();
  if (tid < WB_T) {  // merge the two key halves; publish for the dK / dV kernel
    const float m0 = st_m[tid], m1 = st_m[WB_T + tid], l0 = st_l[tid], l1 = st_l[WB_T + tid];
    const float m = fmaxf(m0, m1);
    const float l = (m0 > -INFINITY ? l0 * __builtin_amdgcn_exp2f(m0 - m) : 0.0f) + (m1 > -INFINITY ? l1 * __builtin_amdgcn_exp2f(m1 - m) : 0.0f);
    st_m[tid] = m;
    st_l[tid] = l;
    if (q_tok[tid] >= 0) {
      A.row_m[seq_off + q_tok[tid]] = m;
      A.row_l[seq_off + q_tok[tid]] = l;
    }
  }
  __syncthreads();
  const float row_m = st_m[my_q], inv_l = 1.0f / st_l[my_q], row_d = st_d[my_q];

  // ---- pass 2: dQ^T[channel][query] = scale * sum_keys K^T[channel][key] dS^T[key][query]
  f32x16 dq[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) dq[mb] = (f32x16)(0.0f);
  wb_fetch(nk, ks, G, wy, wx, 0, tid);
  wb_fetch(nv, vs, G, wy, wx, 0, tid);
  for (int kt = 0; kt < n_tiles; ++kt) {
    __syncthreads();
    wb_store(k_t, nk, tid);
    wb_store(v_t, nv, tid);
    wb_tokens(G, wy, wx, kt * WB_T, tid, k_tok, k_reg);
    __syncthreads();
    if (kt + 1 < n_tiles) {
      wb_fetch(nk, ks, G, wy, wx, (kt + 1) * WB_T, tid);
      wb_fetch(nv, vs, G, wy, wx, (kt + 1) * WB_T, tid);
    }
    const f32x16 st = wb_tile_product(k_t, wk, q_t, wq, lane);
    const f32x16 dpt = wb_tile_product(v_t, wk, do_t, wq, lane);  // dP^T block
    f32x16 ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = wk * 32 + wb_row(r, half);
      const float s = wb_score(st[r], A.scale, A.do_shift && k_reg[key] != my_qreg, k_tok[key] >= 0);
      const float p = q_ok ? __builtin_amdgcn_exp2f(s - row_m) * inv_l : 0.0f;
      ds[r] = p * (dpt[r] - row_d);
    }
    wb_chain_product(dq, k_t, wk, ds, lane);
  }
  // ---- the two key halves of a query block are summed through LDS (the key / value tiles are dead), then stored
  __syncthreads();
  float* red = k_t + wq * (WA_C * 33);  // [channel][33] floats per query block (k_t and v_t are dead: 2 x 16.5 KiB of their 65 KiB)
  if (wk == 1) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(mb * 32 + wb_row(r, half)) * 33 + n] = dq[mb][r];
  }
  __syncthreads();
  if (wk == 0 && q_ok) {
    float* dst = A.g_q + (seq_off + q_tok[my_q]) * WA_C;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int ch = mb * 32 + 8 * q4 + 4 * half;  // registers 4 q4 .. 4 q4 + 3: four consecutive channels
        float4 o;
        o.x = (dq[mb][4 * q4 + 0] + red[(ch + 0) * 33 + n]) * A.scale;
        o.y = (dq[mb][4 * q4 + 1] + red[(ch + 1) * 33 + n]) * A.scale;
        o.z = (dq[mb][4 * q4 + 2] + red[(ch + 2) * 33 + n]) * A.scale;
        o.w = (dq[mb][4 * q4 + 3] + red[(ch + 3) * 33 + n]) * A.scale;
        *reinterpret_cast<float4*>(dst + ch) = o;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256, 1) void wa_bwd_dkv_kernel(WaBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float wb_smem[];
  float* k_t = wb_smem;                       // this workgroup's key tile
  float* v_t = k_t + WB_TILE_FLOATS;
  float* q_t = v_t + WB_TILE_FLOATS;          // current query tile
  float* do_t = q_t + WB_TILE_FLOATS;
  int* k_tok = reinterpret_cast<int*>(do_t + WB_TILE_FLOATS);
  int* k_reg = k_tok + WB_T;
  int* q_tok = k_reg + WB_T;
  int* q_reg = q_tok + WB_T;
  float* st_m = reinterpret_cast<float*>(q_reg + WB_T);  // [64] statistics of the current query tile
  float* st_il = st_m + WB_T;
  float* st_d = st_il + WB_T;

  const WinGeom& G = A.G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = wave >> 1, wk = wave & 1;
  const int n = lane & 31, half = lane >> 5;
  const int n_tiles = (G.Lw + WB_T - 1) / WB_T;
  const int ktile = blockIdx.x, win = blockIdx.y, seq = blockIdx.z;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const size_t seq_off = (size_t)seq * G.h * G.w;
  const float* qs = A.q + seq_off * WA_C;
  const float* ks = A.k + seq_off * WA_C;
  const float* vs = A.v + seq_off * WA_C;
  const float* gos = A.g_out + seq_off * WA_C;

  wb_load_tile(k_t, ks, G, wy, wx, ktile * WB_T, tid, k_tok, k_reg);
  wb_load_tile(v_t, vs, G, wy, wx, ktile * WB_T, tid, nullptr, nullptr);
  __syncthreads();
  const int my_k = wk * 32 + n;                // this lane's key (tile-local)
  const int my_kreg = k_reg[my_k];
  const bool k_ok = k_tok[my_k] >= 0;

  f32x16 dk[4], dv[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) dk[mb] = dv[mb] = (f32x16)(0.0f);
  WbTileRegs nq, ndo;  // the next query / dO tile, in flight
  wb_fetch(nq, qs, G, wy, wx, 0, tid);
  wb_fetch(ndo, gos, G, wy, wx, 0, tid);
  for (int qt = 0; qt < n_tiles; ++qt) {
    __syncthreads();
    wb_store(q_t, nq, tid);
    wb_store(do_t, ndo, tid);
    wb_tokens(G, wy, wx, qt * WB_T, tid, q_tok, q_reg);
    __syncthreads();
    if (qt + 1 < n_tiles) {
      wb_fetch(nq, qs, G, wy, wx, (qt + 1) * WB_T, tid);
      wb_fetch(ndo, gos, G, wy, wx, (qt + 1) * WB_T, tid);
    }
    if (tid < WB_T) {
      const int tok = q_tok[tid];
      st_m[tid] = tok >= 0 ? A.row_m[seq_off + tok] : 0.0f;
      st_il[tid] = tok >= 0 ? 1.0f / A.row_l[seq_off + tok] : 0.0f;  // 0: rows beyond the window contribute nothing
      st_d[tid] = tok >= 0 ? A.row_d[seq_off + tok] : 0.0f;
    }
    __syncthreads();
    const f32x16 s_ = wb_tile_product(q_t, wq, k_t, wk, lane);    // S block: rows = queries, columns = keys
    const f32x16 dp = wb_tile_product(do_t, wq, v_t, wk, lane);   // dP block
    f32x16 p, ds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = wq * 32 + wb_row(r, half);
      const float s = wb_score(s_[r], A.scale, A.do_shift && q_reg[qr] != my_kreg, k_ok);
      const float pr = k_ok ? __builtin_amdgcn_exp2f(s - st_m[qr]) * st_il[qr] : 0.0f;
      p[r] = pr;
      ds[r] = pr * (dp[r] - st_d[qr]);
    }
    wb_chain_product(dv, do_t, wq, p, lane);   // dV^T[channel][key] += dO^T[channel][query] P[query][key]
    wb_chain_product(dk, q_t, wq, ds, lane);   // dK^T[channel][key] += Q^T[channel][query] dS[query][key]
  }
  // ---- sum the two query halves through LDS (the query / dO tiles are dead: one [channel][33] buffer per key block, used
  // for dK and then for dV), then store
  float* red = q_t + wk * (WA_C * 33);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    f32x16 (&acc)[4] = pass == 0 ? dk : dv;
    __syncthreads();
    if (wq == 1) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(mb * 32 + wb_row(r, half)) * 33 + n] = acc[mb][r];
    }
    __syncthreads();
    if (wq == 0) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] += red[(mb * 32 + wb_row(r, half)) * 33 + n];
    }
  }
  if (wq == 0 && k_ok) {
    float* dst_k = A.g_k + (seq_off + k_tok[my_k]) * WA_C;
    float* dst_v = A.g_v + (seq_off + k_tok[my_k]) * WA_C;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int ch = mb * 32 + 8 * q4 + 4 * half;  // registers 4 q4 .. 4 q4 + 3: four consecutive channels
        *reinterpret_cast<float4*>(dst_k + ch) = make_float4(dk[mb][4 * q4 + 0] * A.scale, dk[mb][4 * q4 + 1] * A.scale,
                                                             dk[mb][4 * q4 + 2] * A.scale, dk[mb][4 * q4 + 3] * A.scale);
        *reinterpret_cast<float4*>(dst_v + ch) = make_float4(dv[mb][4 * q4 + 0], dv[mb][4 * q4 + 1], dv[mb][4 * q4 + 2],
                                                             dv[mb][4 * q4 + 3]);
      }
  }
}


// ========================================================================================================= split 16-bit forms
// The same two kernels with every product on the 16-bit matrix instructions, in two flavours of one template (NT = terms per
// operand element):
//   NT = 3  "bf16x6"  three bf16 terms per fp32 element (exact: 3 x 8 significand bits), a product from the six term products that
//           are not below 2^-24 of it, fp32 accumulation — the scheme of gemm_b6_kernel (gemm_f32.hpp): fp32's exponent range, no
//           gains to manage, 2.7 x the matrix rate of v_mfma_f32_32x32x2_f32.
//   NT = 2  "f16x3"   two fp16 terms per element (22 significand bits), three term products hi.lo + lo.hi + hi.hi (the dropped
//           lo.lo is below 2^-22 of the product) — the forward's and the decoder's scheme (split_f16.hpp): HALF the matrix
//           instructions and about half the split arithmetic of bf16x6.  fp16 has a 5-bit exponent, so every operand carries an
//           exact power-of-two gain:
//             q, k, v, dO   one gain per TENSOR from its absolute maximum (wa_bwd_absmax4_kernel -> 4 words in the workspace, read
//                           by every workgroup: max * gain in [2^14, 2^15)); hi + lo then represent an element to
//                           max(2^-22 relative, 2^-39 of the tensor's maximum);
//             P             2^15 (P <= 1);
//             dS (dK pass)  |dS_ij| <= P_ij (|<dO_i, V_j>| + |<dO_i, O_i>|) <= 2 * 128 * max|dO| * max|V|: in gained units below 2^38,
//                           so 2^-23 puts it below 2^15 whatever the data;
//             dQ pass       the query is the LANE (a column of every B operand it supplies), so gains per query are free: its dO
//                           row takes 2^(15 - e_i) with ||dO_i|| < 2^e_i (wa_bwd_rowdot_kernel publishes the norms) — a row 2^-24
//                           below the tensor's maximum keeps its 22 bits — and dS^T a constant 2^-20:
//                           |dS_ij| <= 2 ||dO_i|| sqrt(128) max|V| < 2^34.5 in gained units; folded into 1 / l_i.
//           The accumulators hold gained results; every gain is a power of two and is multiplied back in exactly.
//
// Geometry.  A workgroup owns 128 STATIONARY rows (dK/dV kernel: keys, dQ kernel: queries), one 32-row block per wave, and loops
// over 32-row STREAMING tiles of the other side.  Lane (n, half) of a 16-bit matrix instruction supplies 8 consecutive k of row /
// column n, so an operand must be stored k-contiguous, and the backward needs the streaming tiles in BOTH orientations:
//   role 1  [row][channel]  k = channels:  S = Q K^T, dP = dO V^T           -> R1: [term][32 rows][128 ch] 16-bit, row stride 272 B
//   role 2  [channel][row]  k = rows:      dV^T = dO^T P, dK^T = Q^T dS, dQ^T = K^T dS^T
//                                                                           -> R2: [term][128 ch][32 rows] 16-bit, row stride 80 B
// (strides chosen so that the 16 lanes of a ds_read_b128 group fall into distinct banks).  Role 2 stores the rows PERMUTED: the B
// operand of those products is P / dS straight out of its accumulator registers — register r of lane (n, half) is row
// f(r, half) = (r & 3) + 8 (r >> 2) + 4 half — so K16-step s of lane half h pairs slot j with row 16 s + 8 (j >> 2) + 4 h + (j & 3),
// and position 16 s + 8 h + j of an R2 line holds exactly that row: one ds_read_b128 per fragment.
// The stationary rows never touch LDS: each wave keeps the 8 K16-steps x NT terms of its own 32 rows (x 2 tensors) in registers
// (one wave per SIMD, 512 registers), read from global memory once.  A streaming tile is fetched into registers one iteration ahead
// (thread (row, 8-channel chunk): role 1's own shape), split and stored as R1, staged as fp32 [row][132] for the transposition, and
// re-read by thread (channel, 16 rows) in permuted order to be split and stored as R2.
// No cross-wave reduction, no atomics: a wave owns its 32 rows of the result; results are deterministic.
#define WB6_R 128                // stationary rows per workgroup
#define WB6_T 32                 // rows of a streaming tile
#define WB6_R1_ROW 272           // bytes of an R1 row (128 x 16 bit + 16 pad)
#define WB6_R1_TERM (WB6_T * WB6_R1_ROW)
#define WB6_R2_ROW 80            // bytes of an R2 line (32 x 16 bit + 16 pad)
#define WB6_R2_TERM (WA_C * WB6_R2_ROW)
#define WB6_ST_LD 132            // floats per row of the fp32 staging tile
#define WB6_ST_BYTES (WB6_T * WB6_ST_LD * 4)

typedef __bf16 wb6_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wb6_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wb6_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned wb6_pk(float a, float b) {  // v_cvt_pk_bf16_f32, round to nearest even
  const wb6_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wb6_bf16x2));
}
template <int NT>
struct WbsFrag {
  u32x4 t[NT];  // NT = 3: hi | mid | lo, 8 bf16 each;  NT = 2: hi | lo, 8 fp16 each
};
// 8 values -> their terms.  mult: the operand's power-of-two gain (NT = 2 only; bf16 has fp32's exponent range)
template <int NT>
__device__ __forceinline__ WbsFrag<NT> wbs_split8(const float (&v)[8], float mult) {
  WbsFrag<NT> f;
  if constexpr (NT == 3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = v[2 * i], b = v[2 * i + 1];
      const unsigned h = wb6_pk(a, b);
      const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
      const unsigned m = wb6_pk(ra, rb);
      const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
      f.t[0][i] = h;
      f.t[1][i] = m;
      f.t[2][i] = wb6_pk(sa, sb);
    }
  } else {
    const PartsH p = split8h(v, mult);
    f.t[0] = __builtin_bit_cast(u32x4, p.hi);
    f.t[1] = __builtin_bit_cast(u32x4, p.lo);
  }
  return f;
}
// acc += A . B from the term products that matter, smallest first
#ifndef WBS_EXP
#define WBS_EXP 0  // removal experiments (wrong results, meaningful times): tools/exp/wa_bwd_removal.sh
#endif
template <int NT>
__device__ __forceinline__ f32x16 wbs_mfma(const WbsFrag<NT>& a, const WbsFrag<NT>& b, f32x16 acc) {
  if (WBS_EXP == 1) return acc;
  if constexpr (NT == 3) {
#define WB6_P(TA_, TB_) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wb6_bf16x8, a.t[TA_]), __builtin_bit_cast(wb6_bf16x8, b.t[TB_]), acc, 0, 0, 0)
    WB6_P(2, 0);
    WB6_P(0, 2);
    WB6_P(1, 1);
    WB6_P(1, 0);
    WB6_P(0, 1);
    WB6_P(0, 0);
#undef WB6_P
  } else {
#define WB6_P(TA_, TB_) acc = mfma16h(__builtin_bit_cast(f16x8, a.t[TA_]), __builtin_bit_cast(f16x8, b.t[TB_]), acc)
    WB6_P(1, 0);
    WB6_P(0, 1);
    WB6_P(0, 0);
#undef WB6_P
  }
  return acc;
}
template <int NT>
__device__ __forceinline__ WbsFrag<NT> wbs_lds_frag(const unsigned char* base, int term_stride) {
  WbsFrag<NT> f;
#pragma unroll
  for (int t = 0; t < NT; ++t) f.t[t] = *reinterpret_cast<const u32x4*>(base + t * term_stride);
  return f;
}
template <int NT>
__device__ __forceinline__ void wbs_pin(WbsFrag<NT>& f) {
  if constexpr (NT == 3) asm volatile("" : "+v"(f.t[0]), "+v"(f.t[1]), "+v"(f.t[2]));
  else asm volatile("" : "+v"(f.t[0]), "+v"(f.t[1]));
}

// ---- the gains of the f16x3 flavour
// gain of a tensor from the bit pattern of its absolute maximum: max * gain in [2^14, 2^15) (2^15 for an all-zero tensor)
__device__ __forceinline__ float wbs_tensor_gain(unsigned absmax_bits) { return pow2i(gain_exp(__uint_as_float(absmax_bits))); }
#define WBS_P_GAIN 32768.0f               // P <= 1 (+ a rounding) -> below 2^15 + 1
#define WBS_DS_GAIN 1.1920928955078125e-7f  // 2^-23: gained dS of the dK pass, below 2^38, -> below 2^15
// absolute maxima of the four operand tensors as bit patterns (non-negative floats order like unsigned integers), out[4] zeroed
// by the caller.  A maximum is the same in any order: deterministic with atomics.
__global__ __launch_bounds__(256) void wa_bwd_absmax4_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, const float* __restrict__ g_out,
                                                             long long n_float4, unsigned* __restrict__ out) {
  const float* src = blockIdx.y == 0 ? q : blockIdx.y == 1 ? k : blockIdx.y == 2 ? v : g_out;
  unsigned m = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_float4; i += (long long)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4*>(src)[i];
    m = max(max(m, __float_as_uint(x.x) & 0x7fffffffu), max(__float_as_uint(x.y) & 0x7fffffffu, __float_as_uint(x.z) & 0x7fffffffu));
    m = max(m, __float_as_uint(x.w) & 0x7fffffffu);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  // ONE atomic per workgroup (per wave they were 7 680 read-modify-writes on four addresses: 96 us per call in the first version)
  __shared__ unsigned red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(red[0], red[1]), max(red[2], red[3]));
    if (m) atomicMax(out + blockIdx.y, m);
  }
}

// the 8 K16-steps of one stationary row (window-local index li; zero beyond the window): lane (n, half) holds channels 16 u + 8 half ..
template <int NT>
__device__ __forceinline__ void wbs_stationary(WbsFrag<NT> (&f)[8], const float* __restrict__ src_seq, int tok, int half, float mult) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (tok >= 0) {
      const float4 a = reinterpret_cast<const float4*>(src_seq + (size_t)tok * WA_C + 16 * u + 8 * half)[0];
      const float4 b = reinterpret_cast<const float4*>(src_seq + (size_t)tok * WA_C + 16 * u + 8 * half)[1];
      v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    }
    f[u] = wbs_split8<NT>(v, mult);
  }
}

// streaming tile, fetch role: thread t -> row t >> 3, chunks (t & 7) and (t & 7) + 8 of 8 channels
struct Wb6TileRegs {
  float4 v[4];
};
__device__ __forceinline__ void wb6_fetch(Wb6TileRegs& r, const float* __restrict__ src_seq, const WinGeom& G, int wy, int wx, int i0,
                                          int tid) {
  const int li = i0 + (tid >> 3), c = tid & 7;
  r.v[0] = r.v[1] = r.v[2] = r.v[3] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (WBS_EXP == 5) return;
  if (li < G.Lw) {
    int region;
    const int tok = win_token(G, wy, wx, li, region);
    const float4* p = reinterpret_cast<const float4*>(src_seq + (size_t)tok * WA_C);
    r.v[0] = p[2 * c], r.v[1] = p[2 * c + 1], r.v[2] = p[2 * (c + 8)], r.v[3] = p[2 * (c + 8) + 1];
  }
}
// registers -> R1 (split) and, if st, the fp32 staging tile (raw values: the gain is applied where they are split)
template <int NT>
__device__ __forceinline__ void wbs_store_r1(unsigned char* r1, float* st, const Wb6TileRegs& r, int tid, float mult) {
  if (WBS_EXP == 3) return;
  const int row = tid >> 3, c = tid & 7;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float4 a = r.v[2 * q], b = r.v[2 * q + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const WbsFrag<NT> f = wbs_split8<NT>(v, mult);
    const int ch0 = 8 * (c + 8 * q);
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<u32x4*>(r1 + t * WB6_R1_TERM + row * WB6_R1_ROW + ch0 * 2) = f.t[t];
    if (st) {
      *reinterpret_cast<float4*>(st + row * WB6_ST_LD + ch0) = a;
      *reinterpret_cast<float4*>(st + row * WB6_ST_LD + ch0 + 4) = b;
    }
  }
}
// registers -> the fp32 staging tile only (a tensor that is needed in role 2 alone)
__device__ __forceinline__ void wb6_store_staging(float* st, const Wb6TileRegs& r, int tid) {
  const int row = tid >> 3, c = tid & 7;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int ch0 = 8 * (c + 8 * q);
    *reinterpret_cast<float4*>(st + row * WB6_ST_LD + ch0) = r.v[2 * q];
    *reinterpret_cast<float4*>(st + row * WB6_ST_LD + ch0 + 4) = r.v[2 * q + 1];
  }
}
// staging tile -> R2: thread t -> channel t & 127, K16-step t >> 7; both lane halves' fragments of that step
template <int NT>
__device__ __forceinline__ void wbs_build_r2(unsigned char* r2, const float* st, int tid, float mult) {
  if (WBS_EXP == 2) return;
  const int ch = tid & 127, s = tid >> 7;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = st[(16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) * WB6_ST_LD + ch];
    const WbsFrag<NT> f = wbs_split8<NT>(v, mult);
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<u32x4*>(r2 + t * WB6_R2_TERM + ch * WB6_R2_ROW + (16 * s + 8 * h) * 2) = f.t[t];
  }
}
// S-type product: acc[rows = the tile's 32 rows][cols = this wave's stationary rows] over the 128 channels.  The fragments of
// step u + 1 are requested before the matrix instructions of step u (one wave per SIMD: nothing else hides an LDS round trip).
template <int NT>
__device__ __forceinline__ f32x16 wbs_tile_product(const unsigned char* r1, const WbsFrag<NT> (&stat)[8], int n, int half) {
  f32x16 acc = (f32x16)(0.0f);
  const unsigned char* base = r1 + n * WB6_R1_ROW + 16 * half;
  WbsFrag<NT> cur = wbs_lds_frag<NT>(base, WB6_R1_TERM);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    WbsFrag<NT> nxt = cur;
    if (u + 1 < 8) nxt = wbs_lds_frag<NT>(base + 32 * (u + 1), WB6_R1_TERM);
    __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise sinks the reads to their use: read, wait, multiply, read, wait ...)
    wbs_pin<NT>(cur);
    acc = wbs_mfma<NT>(cur, stat[u], acc);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
  return acc;
}
// chain product: out^T[channel][col] += sum over the tile's 32 rows X^T[channel][row] s[row][col], s = accumulator registers
// (mult: the gain of s, NT = 2)
template <int NT>
__device__ __forceinline__ void wbs_chain_product(f32x16 (&out)[4], const unsigned char* r2, const f32x16& s, int n, int half, float mult) {
  const unsigned char* base = r2 + n * WB6_R2_ROW + 16 * half;
  WbsFrag<NT> cur = wbs_lds_frag<NT>(base, WB6_R2_TERM);  // unit (st, mb) = (0, 0); in flight while s is split
  WbsFrag<NT> b[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = s[8 * st + j];
    b[st] = wbs_split8<NT>(v, mult);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int st = i >> 2, mb = i & 3;
    WbsFrag<NT> nxt = cur;
    if (i + 1 < 8) nxt = wbs_lds_frag<NT>(base + 32 * ((i + 1) >> 2) + ((i + 1) & 3) * 32 * WB6_R2_ROW, WB6_R2_TERM);
    __builtin_amdgcn_sched_barrier(0);
    wbs_pin<NT>(cur);
    out[mb] = wbs_mfma<NT>(cur, b[st], out[mb]);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
}
__device__ __forceinline__ void wb6_store_rows(float* __restrict__ dst_row, const f32x16 (&acc)[4], int half, float scale) {
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int ch = mb * 32 + 8 * q4 + 4 * half;  // registers 4 q4 .. 4 q4 + 3: four consecutive channels
      *reinterpret_cast<float4*>(dst_row + ch) = make_float4(acc[mb][4 * q4 + 0] * scale, acc[mb][4 * q4 + 1] * scale,
                                                             acc[mb][4 * q4 + 2] * scale, acc[mb][4 * q4 + 3] * scale);
    }
}

// LDS of both kernels: R1 x N_R1 | R2 x 2 | staging x 2 | token ids, regions, statistics of the streaming tile.
// bf16x6: 2 R1 regions (147 KiB in all); f16x3: its regions are 2/3 the size, so 4 R1 regions fit (142 KiB) and EVERY pass
// takes two 32-row sub-tiles per set of barriers (one wave per SIMD: each barrier phase is an exposed LDS round trip).
template <int NT>
struct WbsLds {
  static constexpr int N_R1 = NT == 2 ? 4 : 2;
  static constexpr int R1_BYTES = NT * WB6_R1_TERM;
  static constexpr int R2_BYTES = NT * WB6_R2_TERM;
  static constexpr int off_r1(int i) { return i * R1_BYTES; }
  static constexpr int off_r2(int i) { return N_R1 * R1_BYTES + i * R2_BYTES; }
  static constexpr int off_st(int i) { return N_R1 * R1_BYTES + 2 * R2_BYTES + i * WB6_ST_BYTES; }
  static constexpr int OFF_MISC = N_R1 * R1_BYTES + 2 * R2_BYTES + 2 * WB6_ST_BYTES;
  static constexpr size_t BYTES = (size_t)OFF_MISC + 2 * WB6_T * sizeof(float4);  // (statistics of two sub-tiles)
};

// ---------------------------------------------------------------------------------------------------------------- dK, dV
// Two launches, WHICH = 0: dV (S -> P -> dV^T = dO^T P), 1: dK (S, dP -> dS -> dK^T = Q^T dS).  One kernel for both needs the
// stationary K and V fragments (192 registers at NT = 3) next to two 64-register results and the tile in flight: 512 registers and
// 71 spilled ones, slower than the exact-f32 kernel.  Apart, the dV pass keeps only K (96 + 64) and the dK pass K and V (192 + 64);
// the price is S twice (240 instead of 192 matrix instructions per 32 x 32 block pair at NT = 3).
template <int WHICH, int NT>
__global__ __launch_bounds__(256, 1) void wa_bwd_dkv_split_kernel(WaBwdArgs A) {
  using L = WbsLds<NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char wb6_smem[];
  // NSUB 32-row sub-tiles per iteration (one set of barriers for all of them): 2 in the dV pass, whose LDS need per sub-tile is
  // one R1, one R2 and one staging tile (the regions of the tensor it does not store are free), and in every f16x3 pass; 1 in
  // the bf16x6 dK pass
  constexpr int NSUB = (WHICH == 0 || NT == 2) ? 2 : 1;
  unsigned char* q_r1 = wb6_smem + L::off_r1(0);
  unsigned char* do_r1 = wb6_smem + L::off_r1(NSUB);
  unsigned char* x_r2 = wb6_smem + L::off_r2(0);   // role 2 of the tensor the chain product reads: dO (dV pass) or Q (dK pass)
  float* x_st = reinterpret_cast<float*>(wb6_smem + L::off_st(0));
  float4* q_info = reinterpret_cast<float4*>(wb6_smem + L::OFF_MISC);  // per query of the tile: maximum, 1 / sum, <dO, O>, wrap region

  const WinGeom& G = A.G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, half = lane >> 5;
  const int n_tiles = (G.Lw + WB6_T - 1) / WB6_T;
  const int ktile = blockIdx.x, win = blockIdx.y, seq = blockIdx.z;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const size_t seq_off = (size_t)seq * G.h * G.w;
  const float* qs = A.q + seq_off * WA_C;
  const float* ks = A.k + seq_off * WA_C;
  const float* vs = A.v + seq_off * WA_C;
  const float* gos = A.g_out + seq_off * WA_C;

  // gains (NT = 2; all 1 otherwise): operands, the score's scale, <dO, O> in the units of the dP accumulator, the result
  float g_q = 1.0f, g_k = 1.0f, g_v = 1.0f, g_do = 1.0f, x_gain = 1.0f;
  if constexpr (NT == 2) {
    g_q = wbs_tensor_gain(A.absmax[0]), g_k = wbs_tensor_gain(A.absmax[1]), g_v = wbs_tensor_gain(A.absmax[2]);
    g_do = wbs_tensor_gain(A.absmax[3]);
    x_gain = WHICH == 0 ? WBS_P_GAIN : WBS_DS_GAIN;
  }
  const float s_scale = A.scale / (g_q * g_k);
  const float d_gain = g_do * g_v;
  // dV^T holds gain(dO) * 2^15 * dV;  dK^T holds gain(Q) * gain(dO) gain(V) 2^-23 * dK / scale
  const float out_scale = WHICH == 0 ? 1.0f / (g_do * x_gain) : A.scale / (g_q * (NT == 2 ? d_gain * x_gain : 1.0f));

  // this lane's key: row n of the wave's block
  const int my_li = ktile * WB6_R + wave * 32 + n;
  int my_kreg = 0;
  const int my_ktok = my_li < G.Lw ? win_token(G, wy, wx, my_li, my_kreg) : -1;
  const bool k_ok = my_ktok >= 0;
  WbsFrag<NT> kf[8], vf[WHICH ? 8 : 1];
  wbs_stationary<NT>(kf, ks, my_ktok, half, g_k);
  if constexpr (WHICH == 1) wbs_stationary<NT>(vf, vs, my_ktok, half, g_v);

  f32x16 res[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) res[mb] = (f32x16)(0.0f);
  const int n_iter = (n_tiles + NSUB - 1) / NSUB;
  Wb6TileRegs nq[NSUB], ndo[NSUB];  // the next query / dO sub-tiles, in flight
#pragma unroll
  for (int sb = 0; sb < NSUB; ++sb) {
    wb6_fetch(nq[sb], qs, G, wy, wx, sb * WB6_T, tid);
    wb6_fetch(ndo[sb], gos, G, wy, wx, sb * WB6_T, tid);
  }
  for (int qt = 0; qt < n_iter; ++qt) {
    __syncthreads();  // the previous tile has been consumed
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb) {
      if constexpr (WHICH == 0) {
        wbs_store_r1<NT>(q_r1 + sb * L::R1_BYTES, nullptr, nq[sb], tid, g_q);
        wb6_store_staging(x_st + sb * (WB6_ST_BYTES / 4), ndo[sb], tid);
      } else {
        wbs_store_r1<NT>(q_r1 + sb * L::R1_BYTES, x_st + sb * (WB6_ST_BYTES / 4), nq[sb], tid, g_q);
        wbs_store_r1<NT>(do_r1 + sb * L::R1_BYTES, nullptr, ndo[sb], tid, g_do);
      }
    }
    if (tid < NSUB * WB6_T) {
      int region = 0;
      const int li = qt * NSUB * WB6_T + tid;
      const int tok = li < G.Lw ? win_token(G, wy, wx, li, region) : -1;
      // (1 / sum = 0 for rows beyond the window: they contribute nothing)
      q_info[tid] = make_float4(tok >= 0 ? A.row_m[seq_off + tok] : 0.0f, tok >= 0 ? 1.0f / A.row_l[seq_off + tok] : 0.0f,
                                tok >= 0 ? A.row_d[seq_off + tok] * d_gain : 0.0f, __int_as_float(region));
    }
    __syncthreads();
    if (qt + 1 < n_iter) {
#pragma unroll
      for (int sb = 0; sb < NSUB; ++sb) {
        wb6_fetch(nq[sb], qs, G, wy, wx, ((qt + 1) * NSUB + sb) * WB6_T, tid);
        wb6_fetch(ndo[sb], gos, G, wy, wx, ((qt + 1) * NSUB + sb) * WB6_T, tid);
      }
    }
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb)
      wbs_build_r2<NT>(x_r2 + sb * L::R2_BYTES, x_st + sb * (WB6_ST_BYTES / 4), tid, WHICH == 0 ? g_do : g_q);
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb) {
      const f32x16 s_ = wbs_tile_product<NT>(q_r1 + sb * L::R1_BYTES, kf, n, half);    // S block: rows = queries, columns = keys
      f32x16 dp = (f32x16)(0.0f);
      if constexpr (WHICH == 1) dp = wbs_tile_product<NT>(do_r1 + sb * L::R1_BYTES, vf, n, half);   // dP block
      float4 info[16];  // one batch of 16 reads, one wait (read where they are used they are 64 reads with a wait each)
#pragma unroll
      for (int r = 0; r < 16; ++r) info[r] = q_info[sb * WB6_T + wb_row(r, half)];
      __builtin_amdgcn_sched_barrier(0);
      f32x16 x;  // P (dV pass) or dS (dK pass)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = wb_score(s_[r], s_scale, A.do_shift && __float_as_int(info[r].w) != my_kreg, k_ok);
        const float pr = k_ok ? __builtin_amdgcn_exp2f(s - info[r].x) * info[r].y : 0.0f;
        x[r] = WHICH == 0 ? pr : pr * (dp[r] - info[r].z);
        if (WBS_EXP == 4) x[r] = s_[r] + dp[r];
      }
      // dV^T[channel][key] += dO^T[channel][query] P[query][key]   |   dK^T[channel][key] += Q^T[channel][query] dS[query][key]
      wbs_chain_product<NT>(res, x_r2 + sb * L::R2_BYTES, x, n, half, x_gain);
    }
  }
  if (k_ok) wb6_store_rows((WHICH == 0 ? A.g_v : A.g_k) + (seq_off + my_ktok) * WA_C, res, half, out_scale);
}

// ---------------------------------------------------------------------------------------------------------------- dQ
// HAVE_STATS: the forward published the row statistics (mnerf_window_attention_presplit_stats): no first pass
template <bool HAVE_STATS, int NT>
__global__ __launch_bounds__(256, 1) void wa_bwd_dq_split_kernel(WaBwdArgs A) {
  using L = WbsLds<NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char wb6_smem[];
  constexpr int NSUB2 = NT == 2 ? 2 : 1;  // 32-key sub-tiles per iteration of pass 2
  unsigned char* k_r1 = wb6_smem + L::off_r1(0);
  unsigned char* v_r1 = wb6_smem + L::off_r1(NSUB2);
  unsigned char* k_r2 = wb6_smem + L::off_r2(0);
  float* k_st = reinterpret_cast<float*>(wb6_smem + L::off_st(0));
  int* k_info = reinterpret_cast<int*>(wb6_smem + L::OFF_MISC);  // per key of the tile: wrap region, -1 beyond the window

  const WinGeom& G = A.G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, half = lane >> 5;
  const int n_tiles = (G.Lw + WB6_T - 1) / WB6_T;
  const int qtile = blockIdx.x, win = blockIdx.y, seq = blockIdx.z;
  const int wy = win / G.splits, wx = win - wy * G.splits;
  const size_t seq_off = (size_t)seq * G.h * G.w;
  const float* qs = A.q + seq_off * WA_C;
  const float* ks = A.k + seq_off * WA_C;
  const float* vs = A.v + seq_off * WA_C;
  const float* gos = A.g_out + seq_off * WA_C;

  // this lane's query: row n of the wave's block
  const int my_li = qtile * WB6_R + wave * 32 + n;
  int my_qreg = 0;
  const int my_qtok = my_li < G.Lw ? win_token(G, wy, wx, my_li, my_qreg) : -1;
  const bool q_ok = my_qtok >= 0;

  // gains (NT = 2; all 1 otherwise).  The query is the lane, i.e. a column of every B operand it supplies, so its dO row takes a
  // gain of its OWN (g_do, from ||dO_i||: the row's elements land below 2^15 however far the row is below the tensor's
  // maximum) and dS^T a constant one: |dS_ij| <= 2 ||dO_i|| sqrt(128) max|V| < 2^(15 + 4.5 + 15) in gained units, times 2^-20.
  float g_q = 1.0f, g_k = 1.0f, g_v = 1.0f, g_do = 1.0f, lane_gain = 1.0f;
  if constexpr (NT == 2) {
    g_q = wbs_tensor_gain(A.absmax[0]), g_k = wbs_tensor_gain(A.absmax[1]), g_v = wbs_tensor_gain(A.absmax[2]);
    g_do = pow2i(gain_exp(q_ok ? sqrtf(A.row_n2[seq_off + my_qtok]) : 0.0f));
    lane_gain = 9.5367431640625e-7f;  // 2^-20
  }
  const float s_scale = A.scale / (g_q * g_k);
  const float d_gain = g_do * g_v;

  WbsFrag<NT> qf[8], dof[8];
  wbs_stationary<NT>(qf, qs, my_qtok, half, g_q);
  wbs_stationary<NT>(dof, gos, my_qtok, half, g_do);
  const float row_d = (q_ok ? A.row_d[seq_off + my_qtok] : 0.0f) * d_gain;

  // ---- pass 1: row statistics (online softmax over the key tiles; a wave sees every key of its queries: no merge across waves)
  // (four 32-key sub-tiles per iteration: this pass only needs K in role 1, the LDS of the other regions is free, and one set of
  // barriers then serves 128 keys)
  constexpr int NSUB1 = 4;
  const int n_iter1 = (n_tiles + NSUB1 - 1) / NSUB1;
  float run_m = -INFINITY, run_l = 0.0f;
  if constexpr (HAVE_STATS) {
    run_m = q_ok ? A.row_m[seq_off + my_qtok] : 0.0f;
    run_l = q_ok ? A.row_l[seq_off + my_qtok] : 1.0f;
  } else {
    Wb6TileRegs nk1[NSUB1];
#pragma unroll
    for (int sb = 0; sb < NSUB1; ++sb) wb6_fetch(nk1[sb], ks, G, wy, wx, sb * WB6_T, tid);
    for (int kt = 0; kt < n_iter1; ++kt) {
      __syncthreads();
#pragma unroll
      for (int sb = 0; sb < NSUB1; ++sb) wbs_store_r1<NT>(wb6_smem + sb * L::R1_BYTES, nullptr, nk1[sb], tid, g_k);
      if (tid < NSUB1 * WB6_T) {
        int region = 0;
        const int li = kt * NSUB1 * WB6_T + tid;
        const int tok = li < G.Lw ? win_token(G, wy, wx, li, region) : -1;
        k_info[tid] = tok >= 0 ? region : -1;
      }
      __syncthreads();
      if (kt + 1 < n_iter1) {
#pragma unroll
        for (int sb = 0; sb < NSUB1; ++sb) wb6_fetch(nk1[sb], ks, G, wy, wx, ((kt + 1) * NSUB1 + sb) * WB6_T, tid);
      }
#pragma unroll
      for (int sb = 0; sb < NSUB1; ++sb) {
        const f32x16 st = wbs_tile_product<NT>(wb6_smem + sb * L::R1_BYTES, qf, n, half);  // S^T block: rows = keys, columns = queries
        int4 ki[4];  // keys 4 half + 8 g .. + 3 are registers 4 g .. 4 g + 3
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) ki[gq] = reinterpret_cast<const int4*>(k_info)[sb * 8 + half + 2 * gq];
        __builtin_amdgcn_sched_barrier(0);
        float sc[16], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kinfo = (r & 3) == 0 ? ki[r >> 2].x : (r & 3) == 1 ? ki[r >> 2].y : (r & 3) == 2 ? ki[r >> 2].z : ki[r >> 2].w;
          sc[r] = wb_score(st[r], s_scale, A.do_shift && kinfo != my_qreg, kinfo >= 0);
          mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(run_m, mx);
        if (m_new > -INFINITY) {
          float sum = 0.0f;
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(sc[r] - m_new);
          sum += __shfl_xor(sum, 32, 64);
          run_l = run_l * (run_m > -INFINITY ? __builtin_amdgcn_exp2f(run_m - m_new) : 0.0f) + sum;
          run_m = m_new;
        }
      }
    }
  }
  __syncthreads();  // pass 2 stores into the same LDS
  if constexpr (!HAVE_STATS) {
    if (half == 0 && q_ok) {  // publish for the dK / dV kernel
      A.row_m[seq_off + my_qtok] = run_m;
      A.row_l[seq_off + my_qtok] = run_l;
    }
  }
  const float inv_l = lane_gain / run_l;  // (the query's dS gain rides on 1 / l)

  // ---- pass 2: dQ^T[channel][query] = scale * sum_keys K^T[channel][key] dS^T[key][query]
  f32x16 dq[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) dq[mb] = (f32x16)(0.0f);
  const int n_iter2 = (n_tiles + NSUB2 - 1) / NSUB2;
  Wb6TileRegs nk[NSUB2], nv[NSUB2];
#pragma unroll
  for (int sb = 0; sb < NSUB2; ++sb) {
    wb6_fetch(nk[sb], ks, G, wy, wx, sb * WB6_T, tid);
    wb6_fetch(nv[sb], vs, G, wy, wx, sb * WB6_T, tid);
  }
  for (int kt = 0; kt < n_iter2; ++kt) {
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < NSUB2; ++sb) {
      wbs_store_r1<NT>(k_r1 + sb * L::R1_BYTES, k_st + sb * (WB6_ST_BYTES / 4), nk[sb], tid, g_k);
      wbs_store_r1<NT>(v_r1 + sb * L::R1_BYTES, nullptr, nv[sb], tid, g_v);
    }
    if (tid < NSUB2 * WB6_T) {
      int region = 0;
      const int li = kt * NSUB2 * WB6_T + tid;
      const int tok = li < G.Lw ? win_token(G, wy, wx, li, region) : -1;
      k_info[tid] = tok >= 0 ? region : -1;
    }
    __syncthreads();
    if (kt + 1 < n_iter2) {
#pragma unroll
      for (int sb = 0; sb < NSUB2; ++sb) {
        wb6_fetch(nk[sb], ks, G, wy, wx, ((kt + 1) * NSUB2 + sb) * WB6_T, tid);
        wb6_fetch(nv[sb], vs, G, wy, wx, ((kt + 1) * NSUB2 + sb) * WB6_T, tid);
      }
    }
#pragma unroll
    for (int sb = 0; sb < NSUB2; ++sb) wbs_build_r2<NT>(k_r2 + sb * L::R2_BYTES, k_st + sb * (WB6_ST_BYTES / 4), tid, g_k);
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < NSUB2; ++sb) {
      const f32x16 st = wbs_tile_product<NT>(k_r1 + sb * L::R1_BYTES, qf, n, half);
      const f32x16 dpt = wbs_tile_product<NT>(v_r1 + sb * L::R1_BYTES, dof, n, half);  // dP^T block
      int4 ki[4];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) ki[gq] = reinterpret_cast<const int4*>(k_info)[sb * 8 + half + 2 * gq];
      __builtin_amdgcn_sched_barrier(0);
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kinfo = (r & 3) == 0 ? ki[r >> 2].x : (r & 3) == 1 ? ki[r >> 2].y : (r & 3) == 2 ? ki[r >> 2].z : ki[r >> 2].w;
        const float s = wb_score(st[r], s_scale, A.do_shift && kinfo != my_qreg, kinfo >= 0);
        const float p = q_ok ? __builtin_amdgcn_exp2f(s - run_m) * inv_l : 0.0f;
        ds[r] = p * (dpt[r] - row_d);
        if (WBS_EXP == 4) ds[r] = st[r] + dpt[r];
      }
      wbs_chain_product<NT>(dq, k_r2 + sb * L::R2_BYTES, ds, n, half, 1.0f);  // (ds already carries its gain)
    }
  }
  // dQ^T holds gain(K) * gain(dO) gain(V) * lane_gain * dQ / scale
  if (q_ok) wb6_store_rows(A.g_q + (seq_off + my_qtok) * WA_C, dq, half, A.scale / (g_k * (NT == 2 ? d_gain * lane_gain : 1.0f)));
}

static size_t wb_lds_bytes() { return (size_t)4 * WB_TILE_FLOATS * sizeof(float) + 4 * WB_T * sizeof(int) + 6 * WB_T * sizeof(float); }

extern "C" int64_t mnerf_window_attention_backward_workspace_bytes(int32_t batch, int32_t h, int32_t w) {
  if (batch < 0 || h < 1 || w < 1) return -1;
  // row maximum | row sum | <dO, O> | ||dO||^2 | the four operand maxima (f16x3 form)
  return ((int64_t)4 * batch * h * w + 4) * (int64_t)sizeof(float);
}

static int wa_backward_impl(const char* who, const float* q, const float* k, const float* v, const float* out, const float* g_out,
                            const float* row_stats, float* g_q, float* g_k, float* g_v, int32_t batch, int32_t h, int32_t w,
                            int32_t num_splits, int32_t shifted, void* workspace, size_t workspace_bytes, void* stream) {
  MNERF_REQUIRE(q && k && v && out && g_out && g_q && g_k && g_v, MNERF_E_NULL, "%s: NULL buffer", who);
  MNERF_REQUIRE(mnerf_aligned16(q) && mnerf_aligned16(k) && mnerf_aligned16(v) && mnerf_aligned16(out) && mnerf_aligned16(g_out) &&
                    mnerf_aligned16(g_q) && mnerf_aligned16(g_k) && mnerf_aligned16(g_v),
                MNERF_E_ALIGN, "%s: buffers must be 16-byte aligned", who);
  WinGeom G;
  int do_shift;
  if (const int rc = wa_geometry(who, batch, h, w, num_splits, shifted, G, do_shift)) return rc;
  if (batch == 0) return MNERF_OK;
  const int64_t need = mnerf_window_attention_backward_workspace_bytes(batch, h, w);
  MNERF_REQUIRE(workspace && (int64_t)workspace_bytes >= need, MNERF_E_RANGE, "%s: workspace of %zu bytes, %lld needed", who,
                workspace_bytes, (long long)need);
  const long long n_tok = (long long)batch * h * w;
  WaBwdArgs A{};
  A.q = q, A.k = k, A.v = v, A.g_out = g_out, A.g_q = g_q, A.g_k = g_k, A.g_v = g_v;
  A.row_m = reinterpret_cast<float*>(workspace);
  A.row_l = A.row_m + n_tok;
  float* row_d = A.row_l + n_tok;
  A.row_d = row_d;
  A.G = G, A.do_shift = do_shift, A.scale = 1.0f / sqrtf((float)WA_C);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = wb_lds_bytes();
  static std::atomic<unsigned long long> attr_set{0};
  if (mnerf_once_per_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)wa_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)wa_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  // MNERF_WA_BWD_MATH (environment): "f16x3" (default) / "bf16x6" = the split 16-bit kernels, "f32" = the exact-f32 ones
  // (read at every call: a dozen getenv per training iteration, and the tests switch forms inside one process)
  const char* math_env = getenv("MNERF_WA_BWD_MATH");
  const int math = (math_env && !strcmp(math_env, "f32")) ? 0 : (math_env && !strcmp(math_env, "bf16x6")) ? 3 : 2;
  float* row_n2 = row_d + n_tok;
  unsigned* absmax = reinterpret_cast<unsigned*>(row_n2 + n_tok);
  A.row_n2 = row_n2, A.absmax = absmax;
  hipLaunchKernelGGL(wa_bwd_rowdot_kernel, dim3((unsigned)((n_tok + 7) / 8)), dim3(256), 0, st, g_out, out, row_d,
                     math == 2 ? row_n2 : nullptr, n_tok);
  if (math) {
    static std::atomic<unsigned long long> attr6_set{0};
    if (mnerf_once_per_device(attr6_set)) {
#define WBS_ATTR(K_, NT_) (void)hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WbsLds<NT_>::BYTES)
      WBS_ATTR((wa_bwd_dq_split_kernel<false, 3>), 3);
      WBS_ATTR((wa_bwd_dq_split_kernel<true, 3>), 3);
      WBS_ATTR((wa_bwd_dkv_split_kernel<0, 3>), 3);
      WBS_ATTR((wa_bwd_dkv_split_kernel<1, 3>), 3);
      WBS_ATTR((wa_bwd_dq_split_kernel<false, 2>), 2);
      WBS_ATTR((wa_bwd_dq_split_kernel<true, 2>), 2);
      WBS_ATTR((wa_bwd_dkv_split_kernel<0, 2>), 2);
      WBS_ATTR((wa_bwd_dkv_split_kernel<1, 2>), 2);
#undef WBS_ATTR
    }
    const dim3 grid6((G.Lw + WB6_R - 1) / WB6_R, num_splits * num_splits, batch);
    if (row_stats) {  // the forward's statistics: read in place by all three kernels
      A.row_m = const_cast<float*>(row_stats);
      A.row_l = A.row_m + n_tok;
    }
    if (math == 2) {
      if (hipMemsetAsync(absmax, 0, 4 * sizeof(unsigned), st) != hipSuccess) return mnerf_check_launch(who);
      const long long n4 = n_tok * (WA_C / 4);
      const unsigned nblk = (unsigned)((n4 + 256 * 8 - 1) / (256 * 8) < 256 ? (n4 + 256 * 8 - 1) / (256 * 8) : 256);
      hipLaunchKernelGGL(wa_bwd_absmax4_kernel, dim3(nblk, 4), dim3(256), 0, st, q, k, v, g_out, n4, absmax);
      const size_t lds2 = WbsLds<2>::BYTES;
      if (row_stats) hipLaunchKernelGGL((wa_bwd_dq_split_kernel<true, 2>), grid6, dim3(256), lds2, st, A);
      else hipLaunchKernelGGL((wa_bwd_dq_split_kernel<false, 2>), grid6, dim3(256), lds2, st, A);
      hipLaunchKernelGGL((wa_bwd_dkv_split_kernel<0, 2>), grid6, dim3(256), lds2, st, A);
      hipLaunchKernelGGL((wa_bwd_dkv_split_kernel<1, 2>), grid6, dim3(256), lds2, st, A);
    } else {
      const size_t lds3 = WbsLds<3>::BYTES;
      if (row_stats) hipLaunchKernelGGL((wa_bwd_dq_split_kernel<true, 3>), grid6, dim3(256), lds3, st, A);
      else hipLaunchKernelGGL((wa_bwd_dq_split_kernel<false, 3>), grid6, dim3(256), lds3, st, A);
      hipLaunchKernelGGL((wa_bwd_dkv_split_kernel<0, 3>), grid6, dim3(256), lds3, st, A);
      hipLaunchKernelGGL((wa_bwd_dkv_split_kernel<1, 3>), grid6, dim3(256), lds3, st, A);
    }
    return mnerf_check_launch(who);
  }
  const int n_tiles = (G.Lw + WB_T - 1) / WB_T;
  const dim3 grid(n_tiles, num_splits * num_splits, batch);
  hipLaunchKernelGGL(wa_bwd_dq_kernel, grid, dim3(256), lds, st, A);
  hipLaunchKernelGGL(wa_bwd_dkv_kernel, grid, dim3(256), lds, st, A);
  return mnerf_check_launch(who);
}

extern "C" int mnerf_window_attention_backward(const float* q, const float* k, const float* v, const float* out, const float* g_out,
                                               float* g_q, float* g_k, float* g_v, int32_t batch, int32_t h, int32_t w,
                                               int32_t num_splits, int32_t shifted, void* workspace, size_t workspace_bytes,
                                               void* stream) {
  return wa_backward_impl("mnerf_window_attention_backward", q, k, v, out, g_out, nullptr, g_q, g_k, g_v, batch, h, w, num_splits,
                          shifted, workspace, workspace_bytes, stream);
}

// the same with the row statistics of the forward (mnerf_window_attention_presplit_stats): the dQ kernel's first pass is skipped
// (the exact-f32 form, MNERF_WA_BWD_MATH=f32, ignores them and recomputes)
extern "C" int mnerf_window_attention_backward_stats(const float* q, const float* k, const float* v, const float* out,
                                                     const float* g_out, const float* row_stats, float* g_q, float* g_k, float* g_v,
                                                     int32_t batch, int32_t h, int32_t w, int32_t num_splits, int32_t shifted,
                                                     void* workspace, size_t workspace_bytes, void* stream) {
  const char* who = "mnerf_window_attention_backward_stats";
  MNERF_REQUIRE(row_stats, MNERF_E_NULL, "%s: row_stats is NULL", who);
  return wa_backward_impl(who, q, k, v, out, g_out, row_stats, g_q, g_k, g_v, batch, h, w, num_splits, shifted, workspace,
                          workspace_bytes, stream);
}
