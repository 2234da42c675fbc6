// Shared by window_attention.hip and qkv.hip: window geometry of the GMFlow swin attention and the layout of the
// prepared K / V operand images (one 32-key tile of one window = 32 KiB + a 32-byte record).
#pragma once
#include "split_f16.hpp"

#define WA_C 128
#define WA_KT 32            // keys per tile
#define WA_IMG_BYTES 32768  // K part [t = 0..7][hi|lo][lane] x 16 B, then V part [t = 0..1][m = 0..3][hi|lo][lane] x 16 B
#define WA_IMG_VOFF 1024    // u32x4 index of the V part inside an image
#define WA_REC_INTS 8       // side record of a tile: ek, ev, 4 words of key wrap regions (one nibble per key), 2 pad

struct WinGeom {
  int h, w, wh, ww, sh, sw, splits, Lw;
};

// window-local index -> token id in the un-rolled [h*w] sequence, and the wrap region of its
// rolled position (0..8), cf. transformer.py:24-36
__device__ __forceinline__ int win_token(const WinGeom& G, int wy, int wx, int li, int& region) {
  const int ly = li / G.ww, lx = li - ly * G.ww;
  const int ry = wy * G.wh + ly, rx = wx * G.ww + lx;  // position after roll by (-sh,-sw)
  int oy = ry + G.sh, ox = rx + G.sw;                  // original position
  if (oy >= G.h) oy -= G.h;
  if (ox >= G.w) ox -= G.w;
  const int regy = (ry >= G.h - G.wh) + (ry >= G.h - G.sh);
  const int regx = (rx >= G.w - G.ww) + (rx >= G.w - G.sw);
  region = regy * 3 + regx;
  return oy * G.w + ox;
}

// wrap region (0..8) of every key of a tile, one nibble per key: lane n < 32 owns key n; rec[2..5] <- the four words
__device__ __forceinline__ void wa_store_regions(const WinGeom& G, int wy, int wx, int kt, int lane, int* rec) {
  const int n = lane & 31;
  int li = kt * WA_KT + n, region;
  if (li >= G.Lw) li = G.Lw - 1;
  (void)win_token(G, wy, wx, li, region);
  unsigned nib = (unsigned)region << (4 * (n & 7));
  nib |= __shfl_xor(nib, 1, 64);
  nib |= __shfl_xor(nib, 2, 64);
  nib |= __shfl_xor(nib, 4, 64);  // lanes 8g .. 8g+7 now hold word g
  if (lane < 32 && (lane & 7) == 0) rec[2 + (lane >> 3)] = (int)nib;
}

static inline int wa_geometry(const char* who, int32_t batch, int32_t h, int32_t w, int32_t num_splits, int32_t shifted,
                       WinGeom& G, int& do_shift) {
  MNERF_REQUIRE(batch >= 0 && h >= 1 && w >= 1 && num_splits >= 1, MNERF_E_RANGE, "%s: batch=%d h=%d w=%d splits=%d",
                who, batch, h, w, num_splits);
  MNERF_REQUIRE(h % num_splits == 0 && w % num_splits == 0, MNERF_E_RANGE, "%s: %dx%d not divisible into %d splits",
                who, h, w, num_splits);
  MNERF_REQUIRE(batch <= 65535 && num_splits * num_splits <= 65535, MNERF_E_RANGE, "%s: grid too large", who);
  G.h = h;
  G.w = w;
  G.splits = num_splits;
  G.wh = h / num_splits;
  G.ww = w / num_splits;
  do_shift = (shifted && num_splits > 1) ? 1 : 0;
  G.sh = do_shift ? G.wh / 2 : 0;
  G.sw = do_shift ? G.ww / 2 : 0;
  G.Lw = G.wh * G.ww;
  return MNERF_OK;
}


static inline size_t wa_workspace_bytes(int32_t batch, int32_t h, int32_t w, int32_t num_splits) {
  if (batch <= 0 || h < 1 || w < 1 || num_splits < 1 || h % num_splits || w % num_splits) return 0;
  const size_t lw = (size_t)(h / num_splits) * (w / num_splits);
  const size_t tiles = (size_t)batch * num_splits * num_splits * ((lw + WA_KT - 1) / WA_KT);
  return tiles * (WA_IMG_BYTES + WA_REC_INTS * sizeof(int32_t));
}
