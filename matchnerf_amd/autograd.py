"""Autograd bridges for training through the HIP path (first step).

The reference trains by back-propagating through its eager op chain
(/root/reference/coach.py:215-243).  Here the FORWARD values of ``mode='train'`` come from
the same HIP kernels as inference (K1-K6); the BACKWARD uses hand-written HIP
kernels for compositing and the cost volume and, where a backward kernel does not exist yet (window attention,
the conditional MLP + ray transformer), re-evaluates that op with differentiable PyTorch-ROCm ops on the GPU
(activation-checkpoint style) and back-propagates through the re-evaluation:

* ``window_attention``  — K6 forward, torch roll/split/softmax re-evaluation for grad(q,k,v)
* ``render_ray_chunk``  — K1..K5 forward in HIP; backward = K5 backward kernel (mnerf_composite_backward) ->
  torch re-evaluation of the conditional MLP + ray transformer (K3+K4) from the saved conditioning rows, with
  sample coordinates from mnerf_ray_samples (the forward's bits) -> K1+K2 backward kernel
  (mnerf_cost_volume_backward, atomic scatter-add into the feature-map gradients)

This module is only entered when gradients are required; inference never touches it, and it
is not a fallback: the forward pass still fails loudly without ``libmnerf_hip.so``.
Training-time ray counts are tiny (``rand_rays_train`` = 1024), so the re-evaluation cost is
irrelevant next to the encoder.
"""
import math

import torch
import torch.nn.functional as F

from . import hip
from .camera import pair_list

# ----------------------------------------------------------------------------- K6


def _window_attention_torch(q, k, v, h, w, splits, shifted):
    """Differentiable swin window attention (transformer.py:46-105) on [B, h*w, C] tokens."""
    b, _, c = q.shape
    if splits <= 1:
        return torch.softmax((q @ k.transpose(1, 2)) / math.sqrt(c), -1) @ v
    wh, ww = h // splits, w // splits

    def to_windows(t):
        t = t.reshape(b, h, w, c)
        if shifted:
            t = torch.roll(t, shifts=(-(wh // 2), -(ww // 2)), dims=(1, 2))
        return t.reshape(b, splits, wh, splits, ww, c).permute(0, 1, 3, 2, 4, 5).reshape(b * splits * splits, wh * ww, c)

    qw, kw, vw = to_windows(q), to_windows(k), to_windows(v)
    scores = (qw @ kw.transpose(1, 2)) / math.sqrt(c)
    if shifted:
        ys = torch.arange(h, device=q.device)
        xs = torch.arange(w, device=q.device)
        reg = ((ys >= h - wh).long() + (ys >= h - wh // 2).long())[:, None] * 3 + \
              ((xs >= w - ww).long() + (xs >= w - ww // 2).long())[None, :]
        reg = reg.reshape(splits, wh, splits, ww).permute(0, 2, 1, 3).reshape(splits * splits, wh * ww)
        mask = torch.where(reg[:, :, None] != reg[:, None, :], -100.0, 0.0)
        scores = scores + mask.repeat(b, 1, 1)
    out = torch.softmax(scores, -1) @ vw
    out = out.reshape(b, splits, splits, wh, ww, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, c)
    if shifted:
        out = torch.roll(out, shifts=(wh // 2, ww // 2), dims=(1, 2))
    return out.reshape(b, h * w, c)


class _WindowAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, h, w, splits, shifted):
        ctx.save_for_backward(q, k, v)
        ctx.geom = (h, w, splits, shifted)
        return hip.window_attention(q.contiguous(), k.contiguous(), v.contiguous(), h, w, splits, shifted)

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v = (t.detach().requires_grad_(True) for t in ctx.saved_tensors)
        with torch.enable_grad():
            out = _window_attention_torch(q, k, v, *ctx.geom)
        gq, gk, gv = torch.autograd.grad(out, (q, k, v), grad_out)
        return gq, gk, gv, None, None, None, None


def window_attention(q, k, v, h, w, splits, shifted):
    """HIP forward; differentiable when any input requires grad."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _WindowAttentionFn.apply(q, k, v, h, w, splits, shifted)
    return hip.window_attention(q, k, v, h, w, splits, shifted)


# ----------------------------------------------------------------------------- K1..K5


def ray_directions_torch(kinv, c2w, ray_idx, width, legacy):
    """target-ray directions [R,3] (un-normalised, camera.py:255-278) for pixel indices ``ray_idx`` on the GPU"""
    dev = ray_idx.device
    kinv, c2w = torch.from_numpy(kinv).to(dev), torch.from_numpy(c2w).to(dev)
    off = 0.0 if legacy else 0.5
    py = torch.div(ray_idx, width, rounding_mode="floor")
    px = ray_idx - py * width
    pix = torch.stack([px.float() + off, py.float() + off, torch.ones_like(px, dtype=torch.float32)], -1)
    cam = pix @ kinv.t()
    center = c2w[:, 3][None].expand_as(cam)
    return (torch.cat([cam, torch.ones_like(cam[:, :1])], -1) @ c2w.t()) - center


def decoder_torch(opt, dec, x, dirs, cond, n_views):
    """Differentiable CondNeRF.forward (cond_nerf.py:52-100, ray_transformer.py) with torch ops.
    x [R,S,3] coordinates w.r.t. source view 0, dirs [R,3] unit directions in that view's frame,
    cond [R,S,Dc] = cat(feat_info, color_info, mask_info) -> rgb_s [R,S,3], sigma [R,S]."""
    dev = x.device
    n_r, s_n, _ = x.shape
    legacy = bool(opt.nerf.legacy_coord)
    mask = cond[..., -n_views:]
    L = dec.L_3D
    freq = 2.0 ** torch.arange(L, device=dev, dtype=torch.float32)
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
        from .cond_nerf import raytrans_table
        a = a + torch.from_numpy(raytrans_table(s_n)).to(dev)[None]
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


class RayChunkLaunch:
    """Everything one differentiable ray chunk of one batch element needs to (re)build its C-ABI argument structs:
    the forward launches and the backward kernels must see the same scene, rays and decoder."""

    def __init__(self, opt, dec_module, make_scene, make_rays, make_decoder, view0_extr, kinv, c2w, ray_idx, width,
                 n_views, setbg_opaque):
        self.opt, self.dec_module = opt, dec_module
        self.make_scene, self.make_rays, self.make_decoder = make_scene, make_rays, make_decoder
        self.view0_extr, self.kinv, self.c2w = view0_extr, kinv, c2w
        self.ray_idx, self.width, self.n_views, self.setbg_opaque = ray_idx, width, n_views, setbg_opaque


class _RayChunkFn(torch.autograd.Function):
    """forward : mnerf_cost_volume -> mnerf_decoder_chunk (HIP; per-sample colours / densities kept for the backward)
    backward: mnerf_composite_backward (HIP) -> torch re-evaluation of the conditional MLP + ray transformer from the
              saved conditioning rows (gradients of the decoder parameters and of the rows) ->
              mnerf_cost_volume_backward (HIP) scatters the rows' gradient into the feature maps."""

    @staticmethod
    def forward(ctx, launch, n_feat, *tensors):
        feats = [f.contiguous() for f in tensors[:n_feat]]
        with torch.no_grad():
            sc = launch.make_scene(feats)
            rays, keep = launch.make_rays()
            dec = launch.make_decoder()
            cond = hip.cost_volume(sc, rays, dec.cond_stride, device=feats[0].device)
            rgb, depth, opacity, rgb_s, sigma = hip.decoder_chunk(dec, sc.views[0], rays, cond, want_samples=True)
        ctx.launch, ctx.n_feat = launch, n_feat
        ctx.save_for_backward(cond, rgb_s, sigma, *feats)
        return rgb, depth[:, None], opacity[:, None]

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_op):
        launch, n_feat = ctx.launch, ctx.n_feat
        cond, rgb_s, sigma, *feats = ctx.saved_tensors
        opt, dec_m = launch.opt, launch.dec_module
        r, s = sigma.shape
        sc = launch.make_scene(feats)
        rays, keep = launch.make_rays()
        dec = launch.make_decoder()
        legacy = bool(opt.nerf.legacy_coord)
        _, x_ndc, depth_s = hip.ray_samples(rays, sc.views[0], device=sigma.device)   # same bits as the forward's geometry
        ray = ray_directions_torch(launch.kinv, launch.c2w, launch.ray_idx, launch.width, legacy)
        wo = bool(opt.nerf.wo_render_interval)
        g_rgb_s, g_sigma = hip.composite_backward(
            rgb_s, sigma, depth_s, g_rgb.contiguous().float(), g_depth.reshape(r).contiguous().float(),
            g_op.reshape(r).contiguous().float(), None if wo else ray.norm(dim=-1).contiguous(),
            wo_render_interval=wo, setbg_opaque=launch.setbg_opaque)
        dc = dec.cond_dim
        cond_t = cond.reshape(r, s, dec.cond_stride)[..., :dc].detach().clone().requires_grad_(True)
        dirs = F.normalize(ray, dim=-1) @ torch.as_tensor(launch.view0_extr, device=ray.device)[:, :3].t()
        params = [p for p in dec_m.parameters() if p.requires_grad]
        with torch.enable_grad():
            rgb_s2, sigma2 = decoder_torch(opt, dec_m, x_ndc.detach(), dirs.detach(), cond_t, launch.n_views)
        grads = torch.autograd.grad([rgb_s2, sigma2], [cond_t] + params, [g_rgb_s, g_sigma], allow_unused=True)
        g_feats = [None] * n_feat
        if any(ctx.needs_input_grad[2:2 + n_feat]):
            g_cond = torch.zeros(r * s, dec.cond_stride, device=cond.device)
            g_cond[:, :dc] = grads[0].reshape(r * s, dc)
            g_feats = hip.cost_volume_backward(sc, rays, dec.cond_stride, g_cond, [torch.zeros_like(f) for f in feats])
        it = iter(grads[1:])
        g_params = [next(it) if p.requires_grad else None for p in dec_m.parameters()]
        return (None, None, *g_feats, *g_params)


def render_ray_chunk(launch, feats):
    """Differentiable render of one chunk of rays of one batch element -> (rgb [R,3], depth [R,1], opacity [R,1])."""
    params = list(launch.dec_module.parameters())
    return _RayChunkFn.apply(launch, len(feats), *feats, *params)
