"""How many 16-texel chunks does a ray tile's bilinear footprint touch per (depth index, view, scale)?  CPU study behind the tile
shape of cost_volume_mm_kernel: the bench scene's geometry (config[1]) in float64, a subset of tiles.
usage: cvmm_chunks.py [n_views]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from matchnerf_amd import synthetic as syn  # noqa: E402

H, W, S = 512, 640, 64
V = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = syn.make_scene(H, W, V, seed=0 if V == 3 else 32)
E, K, nf = sc["extrinsics"][0].astype(np.float64), sc["intrinsics"][0].astype(np.float64), sc["near_fars"][0].astype(np.float64)
Et = np.vstack([E[-1][:3], [0, 0, 0, 1]])
c2w = np.linalg.inv(Et)
Kinv = np.linalg.inv(K[-1])
ys, xs = np.meshgrid(np.arange(0, H), np.arange(0, W), indexing="ij")
cam = np.stack([xs, ys, np.ones_like(xs)], -1) @ Kinv.T
ray = cam @ c2w[:3, :3].T
cen = c2w[:3, 3]
depth = nf[-1, 0] + np.arange(S) / (S - 1) * (nf[-1, 1] - nf[-1, 0])


def texel(v, fh, fw, d):
    p = cen + ray * d
    q = (p @ E[v][:3, :3].T + E[v][:3, 3]) @ K[v].T
    u, w_ = q[..., 0] / q[..., 2] / (W - 1), q[..., 1] / q[..., 2] / (H - 1)
    x = np.clip(u * (fw - 1), 0, fw - 1)
    y = np.clip(w_ * (fh - 1), 0, fh - 1)
    return np.floor(x).astype(int), np.floor(y).astype(int)


def count(th, tw, mode, fh, fw):
    """mean chunks per (tile, depth index, view): mode 'rows4' = chunk 4 rows x 4 columns, rows anchored at the tile's first row
    pair, x blocks aligned (the kernel); 'abs4' = rows aligned to 4 as well; 'pair8' = chunk 2 rows x 8 columns (row pairs aligned,
    columns anchored at the tile's first x block)"""
    tot = n = 0
    for v in range(V):
        for d in depth[::7]:
            x0, y0 = texel(v, fh, fw, d)
            x1, y1 = np.minimum(x0 + 1, fw - 1), np.minimum(y0 + 1, fh - 1)
            for ty in range(0, H - th + 1, th * 5):
                for tx in range(0, W - tw + 1, tw * 3):
                    sl = (slice(ty, ty + th), slice(tx, tx + tw))
                    a, b, c, e = x0[sl].ravel(), y0[sl].ravel(), x1[sl].ravel(), y1[sl].ravel()
                    if mode == "rows4":
                        p0 = b.min() >> 1
                        ra, rb, ca, cb = ((b >> 1) - p0) >> 1, ((e >> 1) - p0) >> 1, a >> 2, c >> 2
                    elif mode == "abs4":
                        ra, rb, ca, cb = b >> 2, e >> 2, a >> 2, c >> 2
                    elif mode == "free4":  # chunk 4 rows x 4 columns, rows AND columns anchored at the tile's first row pair / texel column
                        p0, xa = b.min() >> 1, a.min()
                        ra, rb, ca, cb = ((b >> 1) - p0) >> 1, ((e >> 1) - p0) >> 1, (a - xa) >> 2, (c - xa) >> 2
                    else:
                        c0 = a.min() >> 2
                        ra, rb, ca, cb = b >> 1, e >> 1, ((a >> 2) - c0) >> 1, ((c >> 2) - c0) >> 1
                    s = set(zip(ra, ca)) | set(zip(ra, cb)) | set(zip(rb, ca)) | set(zip(rb, cb))
                    tot += len(s)
                    n += 1
    return tot / n


for fh, fw, name in ((H // 8, W // 8, "1/8"), (H // 4, W // 4, "1/4")):
    for th, tw in ((4, 8), (8, 4), (8, 8)):
        print(f"scale {name} tile {tw}x{th} px ({th * tw} rays): chunks per (depth index, view) "
              + "  ".join(f"{m} {count(th, tw, m, fh, fw):.2f}" for m in ("rows4", "free4", "abs4", "pair8")), flush=True)
