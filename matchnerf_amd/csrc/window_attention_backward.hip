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
// Every product is exact fp32 on v_mfma_f32_32x32x2_f32 (the gradients feed an optimizer: no reduced-precision shortcut);
// tiles are staged in LDS channel-major ([channel][row], row stride 65 floats: conflict-free for both operand roles).
// Roll, window split / merge and the wrap-region mask are index arithmetic (wa_common.hpp: win_token), as in the forward.
// No atomics: every output row is written by exactly one workgroup; results are deterministic.
#include "wa_common.hpp"

#define WB_T 64             // rows (queries or keys) per tile
#define WB_LD 65            // LDS row stride of a [channel][row] tile
#define WB_TILE_FLOATS (WA_C * WB_LD)

// accumulator register r of lane (n, half) of a 32x32 block holds row f(r, half) of column n
__device__ __forceinline__ int wb_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 wb_mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// D[token] = <dO[token], O[token]>
__global__ __launch_bounds__(256) void wa_bwd_rowdot_kernel(const float* __restrict__ g_out, const float* __restrict__ out,
                                                            float* __restrict__ d, long long n_tokens) {
  const long long tok = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);  // 32 lanes per token, 4 channels per lane
  if (tok >= n_tokens) return;
  const int c4 = threadIdx.x & 31;
  const float4 a = reinterpret_cast<const float4*>(g_out + tok * WA_C)[c4];
  const float4 b = reinterpret_cast<const float4*>(out + tok * WA_C)[c4];
  float s = (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (c4 == 0) d[tok] = s;
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

static size_t wb_lds_bytes() { return (size_t)4 * WB_TILE_FLOATS * sizeof(float) + 4 * WB_T * sizeof(int) + 6 * WB_T * sizeof(float); }

extern "C" int64_t mnerf_window_attention_backward_workspace_bytes(int32_t batch, int32_t h, int32_t w) {
  if (batch < 0 || h < 1 || w < 1) return -1;
  return (int64_t)3 * batch * h * w * (int64_t)sizeof(float);  // row maximum | row sum | <dO, O>
}

extern "C" int mnerf_window_attention_backward(const float* q, const float* k, const float* v, const float* out, const float* g_out,
                                               float* g_q, float* g_k, float* g_v, int32_t batch, int32_t h, int32_t w,
                                               int32_t num_splits, int32_t shifted, void* workspace, size_t workspace_bytes,
                                               void* stream) {
  const char* who = "mnerf_window_attention_backward";
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
  hipLaunchKernelGGL(wa_bwd_rowdot_kernel, dim3((unsigned)((n_tok + 7) / 8)), dim3(256), 0, st, g_out, out, row_d, n_tok);
  const int n_tiles = (G.Lw + WB_T - 1) / WB_T;
  const dim3 grid(n_tiles, num_splits * num_splits, batch);
  hipLaunchKernelGGL(wa_bwd_dq_kernel, grid, dim3(256), lds, st, A);
  hipLaunchKernelGGL(wa_bwd_dkv_kernel, grid, dim3(256), lds, st, A);
  return mnerf_check_launch(who);
}
