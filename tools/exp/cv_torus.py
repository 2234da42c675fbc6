"""Design input for cost_volume_tile_kernel: how many tap references of a 16-ray x 8-sample tile lose their LDS place for a
given torus shape (first claim wins, as the kernel's compare-and-swap does).  CPU only (numpy).  usage: cv_torus.py [views]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from matchnerf_amd import synthetic as syn

V = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W, S = 512, 640, 64
sc = syn.make_scene(H, W, V, seed=0 if V == 3 else 32)
ex, it, nf = sc["extrinsics"][0], sc["intrinsics"][0], sc["near_fars"][0]
tgt_e, tgt_k = ex[-1], it[-1]
c2w = np.linalg.inv(tgt_e)
near, far = float(nf[-1, 0]), float(nf[-1, 1])
depths = near + (far - near) * np.arange(S) / (S - 1)


def project(px, py):
    cam = np.linalg.inv(tgt_k) @ np.stack([px, py, np.ones_like(px)], 0)
    pts_c = cam[:, :, None] * depths[None, None, :]
    pts_w = (c2w[:3, :3] @ pts_c.reshape(3, -1) + c2w[:3, 3:4]).reshape(3, -1)
    out = []
    for v in range(V):
        p = ex[v][:3, :3] @ pts_w + ex[v][:3, 3:4]
        uv = it[v] @ p
        out.append((uv[:2] / np.maximum(uv[2:], 1e-6)).T.reshape(len(px), S, 2))
    return np.stack(out, 0)


SHAPES = {8: [(8, 4), (16, 2), (16, 3), (12, 3), (16, 4)], 4: [(8, 6), (16, 3), (12, 4), (16, 4), (16, 6), (32, 2), (24, 2), (32,3)]}
SEG = int(os.environ.get("SEG", 8))
rng = np.random.default_rng(0)
lost = {(s, sh): 0 for s in SHAPES for sh in SHAPES[s]}
refs = {8: 0, 4: 0}
distinct = {8: [], 4: []}
for _ in range(300):
    y0 = rng.integers(0, H)
    x0 = rng.integers(0, W - 16)
    xx = np.arange(x0, x0 + 16).astype(np.float64)
    uv = project(xx, np.full_like(xx, y0))                     # [V,16,S,2] full-res pixel coords
    for j0 in range(0, S, SEG):
        blk = uv[:, :, j0:j0 + SEG]
        for scale in (8, 4):
            fw, fh = W // scale, H // scale
            # the kernel's coordinate handling: u = x/(W-1) normalised, texel = u*(fw-1), clamped
            tx = np.clip(blk[..., 0] / (W - 1) * (fw - 1), 0, fw - 1)
            ty = np.clip(blk[..., 1] / (H - 1) * (fh - 1), 0, fh - 1)
            ix0, iy0 = np.floor(tx).astype(int), np.floor(ty).astype(int)
            for v in range(V):
                xs = np.stack([ix0[v], np.minimum(ix0[v] + 1, fw - 1)] * 2, -1).reshape(-1)
                ys = np.stack([iy0[v], iy0[v], np.minimum(iy0[v] + 1, fh - 1), np.minimum(iy0[v] + 1, fh - 1)], -1).reshape(-1)
                key = ys * fw + xs
                refs[scale] += len(key)
                distinct[scale].append(len(np.unique(key)))
                for (tw, th) in SHAPES[scale]:
                    place = (ys % th) * tw + (xs % tw)
                    owner = {}
                    n_lost = 0
                    for k, pl in zip(key.tolist(), place.tolist()):
                        o = owner.setdefault(pl, k)
                        n_lost += o != k
                    lost[(scale, (tw, th))] += n_lost
for scale in (8, 4):
    d = np.array(distinct[scale])
    print(f"V={V} seg {SEG} scale 1/{scale}: distinct texels per (tile, view): median {np.median(d):.0f} p95 {np.percentile(d, 95):.0f} max {d.max()}")
    for sh in SHAPES[scale]:
        print(f"    torus {sh[0]:2d} x {sh[1]:2d} ({sh[0] * sh[1]:3d} places, {sh[0] * sh[1] / 2:4.0f} KiB/map): {100.0 * lost[(scale, sh)] / refs[scale]:6.2f} % of references lose their place")
