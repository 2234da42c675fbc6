"""Helpers for the -m gpu parity tests: build C-ABI argument structs from golden cases."""
import math

import numpy as np
import torch
import torch.nn.functional as F

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


def decoder_torch(opt, dec, x, dirs, cond, n_views):
    """Plain torch reference of CondNeRF.forward (cond_nerf.py:52-100, ray_transformer.py), differentiable: what
    mnerf_decoder_backward is compared with.
    x [R,S,3] coordinates w.r.t. source view 0, dirs [R,3] unit directions in that view's frame,
    cond [R,S,Dc] = cat(feat_info, color_info, mask_info) -> rgb_s [R,S,3], sigma [R,S]."""
    dev = x.device
    n_r, s_n, _ = x.shape
    legacy = bool(opt.nerf.legacy_coord)
    mask = cond[..., -n_views:]
    L = dec.L_3D
    freq = 2.0 ** torch.arange(L, device=dev, dtype=x.dtype)
    if legacy:
        spec = (x[..., None, :] * freq[:, None]).reshape(n_r, s_n, -1)
        enc = torch.cat([x, spec.sin(), spec.cos()], -1)
    else:
        spec = x[..., None] * (freq * math.pi)
        enc = torch.cat([x, torch.stack([spec.sin(), spec.cos()], -2).reshape(n_r, s_n, -1)], -1)
    film = dec.pts_bias(cond)
    hcur = enc
    for i, lin in enumerate(dec.pts_linears):
        hcur = F.relu(lin(hcur) * film)
        if i in list(opt.decoder.skip):
            hcur = torch.cat([enc, hcur], -1)
    act = F.elu if opt.decoder.raytrans_act == "ELU" else F.relu
    a = act(dec.alpha_linear[0](hcur))
    if opt.decoder.raytrans_posenc:
        from matchnerf_amd.cond_nerf import raytrans_table
        a = a + torch.from_numpy(raytrans_table(s_n)).to(dev, x.dtype)[None]
    ra = dec.ray_attention
    q = ra.w_qs(a).reshape(n_r, s_n, 4, 4).permute(0, 2, 1, 3)
    k = ra.w_ks(a).reshape(n_r, s_n, 4, 4).permute(0, 2, 1, 3)
    v_ = ra.w_vs(a).reshape(n_r, s_n, 4, 4).permute(0, 2, 1, 3)
    n_valid = mask.sum(-1)
    scores = (q / 2.0) @ k.transpose(-1, -2)
    scores = torch.where((n_valid > 1)[:, None, :, None], scores, torch.full_like(scores, -1e9))
    o = (torch.softmax(scores, -1) @ v_).permute(0, 2, 1, 3).reshape(n_r, s_n, 16)
    o = ra.layer_norm(ra.fc(o) + a)
    sigma = F.relu(dec.out_alpha_linear[2](act(dec.out_alpha_linear[0](o))))[..., 0]
    if opt.decoder.density_maskfill:
        sigma = torch.where(n_valid < 1, torch.zeros_like(sigma), sigma)
    hv = F.relu(dec.views_linears[0](torch.cat([dec.feature_linear(hcur), dirs[:, None].expand(-1, s_n, -1)], -1)))
    return torch.sigmoid(dec.rgb_linear(hv)), sigma
