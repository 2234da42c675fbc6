"""Design input for an LDS-tiled cost-volume kernel: texel bounding boxes of a block of rays x a segment of samples
in each source view at both feature scales, for the bench scene (BASELINE config[1], synthetic cameras).
CPU only (numpy)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from matchnerf_amd import synthetic as syn

H, W, V, S = 512, 640, 3, 64
sc = syn.make_scene(H, W, V, seed=0)
ex = sc["extrinsics"][0]          # [V+1, 4, 4] world->cam, last = target
it = sc["intrinsics"][0]          # [V+1, 3, 3]
nf = sc["near_fars"][0]
tgt_e, tgt_k = ex[-1], it[-1]
c2w = np.linalg.inv(tgt_e)
near, far = float(nf[-1, 0]), float(nf[-1, 1])
depths = near + (far - near) * np.arange(S) / (S - 1)


def project(px, py):
    """pixels [N] -> per source view texel coords at full res [V, N, S, 2]"""
    cam = np.linalg.inv(tgt_k) @ np.stack([px, py, np.ones_like(px)], 0)          # [3,N]
    pts_c = cam[:, :, None] * depths[None, None, :]                                  # [3,N,S]
    pts_w = (c2w[:3, :3] @ pts_c.reshape(3, -1) + c2w[:3, 3:4]).reshape(3, -1)
    out = []
    for v in range(V):
        p = ex[v][:3, :3] @ pts_w + ex[v][:3, 3:4]
        uv = it[v] @ p
        out.append((uv[:2] / np.maximum(uv[2:], 1e-6)).T.reshape(len(px), S, 2))
    return np.stack(out, 0)


rng = np.random.default_rng(0)
for shape_name, (bh, bw) in (("16x1 row", (1, 16)), ("4x4 tile", (4, 4)), ("8x8 tile", (8, 8))):
    for seg in (8, 16):
        areas = {8: [], 4: []}
        for _ in range(400):
            y0 = rng.integers(0, H - bh)
            x0 = rng.integers(0, W - bw)
            yy, xx = np.meshgrid(np.arange(y0, y0 + bh), np.arange(x0, x0 + bw), indexing="ij")
            uv = project(xx.reshape(-1).astype(np.float64), yy.reshape(-1).astype(np.float64))   # [V,N,S,2]
            for j0 in range(0, S, seg):
                blk = uv[:, :, j0:j0 + seg]                                                     # [V,N,seg,2]
                for scale in (8, 4):
                    t = blk / scale
                    t[..., 0] = np.clip(t[..., 0], 0, W // scale - 1)
                    t[..., 1] = np.clip(t[..., 1], 0, H // scale - 1)
                    lo = np.floor(t.min(axis=(1, 2)))
                    hi = np.floor(t.max(axis=(1, 2))) + 1
                    ext = np.minimum(hi, [W // scale - 1, H // scale - 1]) - lo + 1              # [V,2]
                    areas[scale].extend((ext[:, 0] * ext[:, 1]).tolist())
        for scale in (8, 4):
            a = np.array(areas[scale])
            print(f"{shape_name:9s} seg {seg:2d} scale 1/{scale}: texels per (view) box  median {np.median(a):5.0f}  p95 {np.percentile(a, 95):5.0f}"
                  f"  max {a.max():5.0f}   => KiB per map (x512 B): median {np.median(a) / 2:5.1f}  p95 {np.percentile(a, 95) / 2:5.1f}")
