"""Helpers for the -m gpu parity tests: build C-ABI argument structs from golden cases."""
import numpy as np
import torch

from matchnerf_amd import camera, cond_nerf as CN, hip
from oracle import matchnerf_oracle as O


def ref_layout_to_pair_major(feat_ref, n_views):
    """[V,(V-1)*128,h,w] (reference per-view chunks) -> [P,2,h,w,128] (include/mnerf.h)."""
    out = []
    for a, b in camera.pair_list(n_views):
        f0 = feat_ref[a, (b - 1) * 128:b * 128]      # view a, chunk of pair (a,b)
        f1 = feat_ref[b, a * 128:(a + 1) * 128]      # view b, chunk of pair (a,b)
        out.append(torch.stack([f0, f1], 0))
    return torch.stack(out, 0).permute(0, 1, 3, 4, 2).contiguous()


def pair_feats_to_pair_major(pair_feats):
    """oracle encode_pairs output [(f0 [P,C,h,w], f1)] per scale -> list of [P,2,h,w,128]."""
    return [torch.stack([f0, f1], 1).permute(0, 1, 3, 4, 2).contiguous() for f0, f1 in pair_feats]


def images_rgba(images):
    """[V,3,H,W] -> [V,H,W,4] channel-last, zero pad."""
    v, _, h, w = images.shape
    out = torch.zeros(v, h, w, 4)
    out[..., :3] = images.permute(0, 2, 3, 1)
    return out.contiguous()


def make_scene_struct(cfg, batch, feats_pm_gpu, images_gpu, b=0):
    v = cfg.n_src_views
    sc = hip.Scene()
    sc.n_views, sc.n_scales = v, len(feats_pm_gpu)
    for s, f in enumerate(feats_pm_gpu):
        sc.fh[s], sc.fw[s] = f.shape[2], f.shape[3]
        sc.n_group[s] = cfg.cos_n_group[s]
        sc.feat[s] = f.data_ptr()
    sc.images = images_gpu.data_ptr()
    for i in range(v):
        sc.views[i] = hip.make_view(batch["extrinsics"][b, i, :3].numpy(), batch["intrinsics"][b, i].numpy(),
                                    float(batch["near_fars"][b, i, 0]), float(batch["near_fars"][b, i, 1]))
    return sc


def make_rays_struct(cfg, batch, n_rays, ray_begin=0, ray_idx_gpu=None, b=0):
    h, w = batch["images"].shape[-2:]
    kinv, c2w = camera.target_ray_consts(batch["extrinsics"][b, -1, :3], batch["intrinsics"][b, -1], cfg.legacy_coord)
    return hip.make_rays(n_rays, cfg.sample_intvs, h, w, kinv, c2w, float(batch["near_fars"][b, -1, 0]),
                         float(batch["near_fars"][b, -1, 1]), ray_begin=ray_begin, legacy=cfg.legacy_coord,
                         depth_inverse=(cfg.depth_param == "inverse"),
                         ray_idx_ptr=ray_idx_gpu.data_ptr() if ray_idx_gpu is not None else None)


def make_decoder_struct(cfg, sd, setbg_opaque=False, device="cuda", math=None):
    """math: 'f16x3' / 'bf16x6' / 'f32' (default: MNERF_DECODER_MATH, i.e. what the product uses)."""
    math = math or CN.decoder_math()
    ws, cond_dim, cs = CN.pack_for_math(math)(sd, cfg.n_src_views, cfg.cos_n_group, cfg.L_3D, cfg.legacy_coord)
    small = CN.pack_small(sd, cfg.sample_intvs, cfg.raytrans_posenc)
    ws_t, small_t = torch.from_numpy(ws).to(device), torch.from_numpy(small).to(device)
    d = hip.Decoder()
    d.wstream, d.wstream_floats, d.small_ = ws_t.data_ptr(), ws_t.numel(), small_t.data_ptr()
    d.n_views, d.cond_dim, d.cond_stride, d.L_3D = cfg.n_src_views, cond_dim, cs, cfg.L_3D
    d.raytrans_posenc, d.raytrans_elu = int(cfg.raytrans_posenc), int(cfg.raytrans_act == "ELU")
    d.density_maskfill, d.wo_render_interval = int(cfg.density_maskfill), int(cfg.wo_render_interval)
    d.setbg_opaque = int(setbg_opaque)
    d.wstream_format = CN.WSTREAM_FORMATS[math]
    return d, (ws_t, small_t)  # keep the tensors alive


def cond_with_stride(cond, stride):
    """[N, Dc] -> [N, stride] with the constant-1 column at Dc (layout of mnerf_cost_volume)."""
    n, dc = cond.shape
    out = torch.zeros(n, stride)
    out[:, :dc] = cond
    out[:, dc] = 1.0
    return out
