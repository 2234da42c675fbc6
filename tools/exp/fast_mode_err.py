"""Errors of the one-product fp16 fast mode (MNERF_DECODER_MATH=f16) against the reference goldens, per case:
per-sample colours / densities and rendered RGB / opacity.  usage: fast_mode_err.py [case ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch

from gpu_helpers import cond_with_stride, make_decoder_struct, make_rays_struct
from helpers import golden_case, linf
from matchnerf_amd import hip

for name in sys.argv[1:] or ["c1_default", "rect_wide", "nonlegacy", "v4", "inverse_depth", "demo_own_small", "demo_own"]:
    g, cfg, sd, batch = golden_case(name)
    idx = torch.from_numpy(g["stage_rays"]).int().cuda()
    dc = g["cond"].shape[-1]
    cond = cond_with_stride(torch.from_numpy(g["cond"]).reshape(-1, dc), ((dc + 1 + 7) // 8) * 8).cuda()
    line = f"{name}:"
    for math in ("f16x3", "f16"):
        dec, keep = make_decoder_struct(cfg, sd, setbg_opaque=g["meta"]["setbg_opaque"], math=math)
        rays = make_rays_struct(cfg, batch, idx.numel(), ray_idx_gpu=idx)
        view0 = hip.make_view(batch["extrinsics"][0, 0, :3].numpy(), batch["intrinsics"][0, 0].numpy(),
                              float(batch["near_fars"][0, 0, 0]), float(batch["near_fars"][0, 0, 1]))
        rgb, depth, opacity, rgb_s, sigma = hip.decoder_chunk(dec, view0, rays, cond, want_samples=True)
        sel = g["stage_rays"]
        line += (f"  [{math}] rgb_s {linf(rgb_s.reshape(g['rgb_samples'].shape), g['rgb_samples']):.2e} sigma "
                 f"{linf(sigma.reshape(g['sigma'].shape), g['sigma']):.2e} rgb {linf(rgb, g['rgb'][0, sel]):.2e} opacity "
                 f"{linf(opacity, g['opacity'][0, sel, 0]):.2e}")
    print(line, flush=True)
