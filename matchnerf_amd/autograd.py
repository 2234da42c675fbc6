"""Autograd bridges for training through the HIP path (first step).

The reference trains by back-propagating through its eager op chain
(/root/reference/coach.py:215-243).  Here the FORWARD values of ``mode='train'`` come from
the same HIP kernels as inference (K1-K6); until hand-written backward kernels exist, the
BACKWARD of each custom op re-evaluates that op with differentiable PyTorch-ROCm ops on the
GPU (activation-checkpoint style) and back-propagates through the re-evaluation:

* ``window_attention``  — K6 forward, torch roll/split/softmax re-evaluation for grad(q,k,v)
* ``render_rays``       — K1..K5 forward (mnerf_render_chunk), torch re-evaluation of the ray
  chunk for grad(feature maps, decoder parameters)

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


def _sample_cl(fmap_cl, grid):
    """bilinear / border / align_corners=True lookup of a channel-last map [h,w,C] at grid [N,2]
    -> [N,C] (matchnerf.py:245, gmflow/utils.py:133-134)."""
    out = F.grid_sample(fmap_cl.permute(2, 0, 1)[None], grid[None, :, None, :], mode="bilinear",
                        padding_mode="border", align_corners=True)
    return out[0, :, :, 0].t()


def render_rays_torch(opt, dec, feats_b, images_b, src_extr, src_intr, src_nf, tgt_extr, tgt_intr, tgt_nf,
                      ray_idx, strat_u, height, width, setbg_opaque):
    """Differentiable re-evaluation of one ray chunk of one batch element (matchnerf.py:88-143,
    cond_nerf.py:52-100, ray_transformer.py, nerf.py:101-124) with torch ops.
    feats_b: per scale [P,2,h,w,128]; images_b [V,3,H,W]; dec = CondNeRF parameter holder."""
    from . import camera
    dev = images_b.device
    legacy = bool(opt.nerf.legacy_coord)
    s_n = int(opt.nerf.sample_intvs)
    v_n = images_b.shape[0]
    kinv, c2w = camera.target_ray_consts(tgt_extr, tgt_intr, legacy)
    kinv, c2w = torch.from_numpy(kinv).to(dev), torch.from_numpy(c2w).to(dev)
    off = 0.0 if legacy else 0.5
    py = torch.div(ray_idx, width, rounding_mode="floor")
    px = ray_idx - py * width
    pix = torch.stack([px.float() + off, py.float() + off, torch.ones_like(px, dtype=torch.float32)], -1)
    cam = pix @ kinv.t()
    center = c2w[:, 3][None].expand_as(cam)
    ray = (torch.cat([cam, torch.ones_like(cam[:, :1])], -1) @ c2w.t()) - center
    t = torch.arange(s_n, device=dev, dtype=torch.float32)[None] + (strat_u if strat_u is not None else off)
    near, far = float(tgt_nf[0]), float(tgt_nf[1])
    depth = t / ((s_n - 1) if legacy else s_n) * (far - near) + near
    if opt.nerf.depth.param == "inverse":
        depth = 1 / (depth + 1e-8)
    pts = center[:, None] + ray[:, None] * depth[..., None]                      # [R,S,3]
    n_r = pts.shape[0]
    flat = pts.reshape(-1, 3)

    def project(v):
        e = torch.as_tensor(src_extr[v], device=dev)
        k = torch.as_tensor(src_intr[v], device=dev)
        q = (torch.cat([flat, torch.ones_like(flat[:, :1])], -1) @ e.t()) @ k.t()
        u = q[:, 0] / q[:, 2] / (width - 1)
        w_ = q[:, 1] / q[:, 2] / (height - 1)
        z = (q[:, 2] - float(src_nf[v][0])) / (float(src_nf[v][1]) - float(src_nf[v][0]))
        return u, w_, z

    uvz = [project(v) for v in range(v_n)]
    grids = [torch.stack([u * 2 - 1, w_ * 2 - 1], -1) for u, w_, _ in uvz]
    colors = [_sample_cl(images_b[v].permute(1, 2, 0), grids[v]) for v in range(v_n)]
    masks = [((g[:, 0] > -1) & (g[:, 0] < 1) & (g[:, 1] > -1) & (g[:, 1] < 1)).float() for g in grids]
    groups = list(opt.encoder.cos_n_group)
    pairs = pair_list(v_n)
    feat_cols = []
    for s, fm in enumerate(feats_b):
        acc = 0
        for p, (a, b) in enumerate(pairs):
            fa = _sample_cl(fm[p, 0], grids[a]).reshape(-1, groups[s], 128 // groups[s])
            fb = _sample_cl(fm[p, 1], grids[b]).reshape(-1, groups[s], 128 // groups[s])
            acc = acc + F.cosine_similarity(fa, fb, dim=2, eps=1e-8)
        feat_cols.append(acc / len(pairs))
    cond = torch.cat(feat_cols + colors + [torch.stack(masks, -1)], -1).reshape(n_r, s_n, -1)
    mask = torch.stack(masks, -1).reshape(n_r, s_n, v_n)

    x = torch.stack(uvz[0], -1).reshape(n_r, s_n, 3)
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
    e0 = torch.as_tensor(src_extr[0], device=dev)
    d_ref = F.normalize(ray, dim=-1) @ e0[:, :3].t()
    hv = F.relu(dec.views_linears[0](torch.cat([dec.feature_linear(hcur), d_ref[:, None].expand(-1, s_n, -1)], -1)))
    rgb_s = torch.sigmoid(dec.rgb_linear(hv))
    if opt.nerf.wo_render_interval:
        sd = sigma
    else:
        intv = torch.cat([depth[:, 1:] - depth[:, :-1], torch.full_like(depth[:, :1], 1e10)], 1)
        sd = sigma * intv * ray.norm(dim=-1, keepdim=True)
    alpha = 1 - torch.exp(-sd)
    excl = torch.cat([torch.zeros_like(sd[:, :1]), sd[:, :-1]], 1).cumsum(1)
    wgt = torch.exp(-excl) * alpha
    rgb = (rgb_s * wgt[..., None]).sum(1)
    dep = (depth * wgt).sum(1, keepdim=True)
    opa = wgt.sum(1, keepdim=True)
    if setbg_opaque:
        rgb = rgb + (1 - opa)
    return rgb, dep, opa


class _RenderRaysFn(torch.autograd.Function):
    """forward: mnerf_render_chunk (HIP).  backward: re-evaluate with torch ops and back-propagate
    into the feature maps and the decoder parameters."""

    @staticmethod
    def forward(ctx, dec, launch, n_feat, *tensors):
        feats = tensors[:n_feat]
        ctx.dec, ctx.launch, ctx.n_feat = dec, launch, n_feat
        ctx.save_for_backward(*tensors)
        with torch.no_grad():
            rgb, depth, opacity = launch["hip_render"](feats)
        return rgb, depth, opacity

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_op):
        dec, launch, n_feat = ctx.dec, ctx.launch, ctx.n_feat
        saved = ctx.saved_tensors
        feats = [t.detach().requires_grad_(True) for t in saved[:n_feat]]
        params = list(dec.parameters())
        with torch.enable_grad():
            outs = launch["torch_render"](feats)
        wanted = [t for t in feats if True] + [p for p in params if p.requires_grad]
        grads = torch.autograd.grad(outs, wanted, (g_rgb, g_depth, g_op), allow_unused=True)
        g_feats = list(grads[:n_feat])
        it = iter(grads[n_feat:])
        g_params = [next(it) if p.requires_grad else None for p in params]
        return (None, None, None, *g_feats, *g_params)


def render_rays(dec, launch, feats):
    """Differentiable render of one chunk: ``dec`` is the CondNeRF parameter holder, ``launch`` carries two
    closures over the same arguments, ``hip_render(feats)`` and ``torch_render(feats)``."""
    params = list(dec.parameters())
    return _RenderRaysFn.apply(dec, launch, len(feats), *feats, *params)
