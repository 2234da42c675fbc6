"""Autograd bridges for training through the HIP path.

The reference trains by back-propagating through its eager op chain
(/root/reference/coach.py:215-243).  Here the FORWARD values of ``mode='train'`` come from
the same HIP kernels as inference; the BACKWARD is hand-written HIP for the whole ray chunk and for the
GMFlow transformer:

* ``render_ray_chunk``  — K1..K5 forward in HIP; backward = K5 backward kernel (mnerf_composite_backward) ->
  K3+K4 backward (mnerf_decoder_backward: forward re-evaluated from the saved conditioning rows with sample coordinates
  from mnerf_ray_samples — the forward's bits; GEMMs fp32-grade in split-bf16 by default, exact fp32 with
  MNERF_GEMM_MATH=f32) -> K1+K2 backward kernel (mnerf_cost_volume_backward, walking, atomic adds into the map gradients)
* ``transformer_layer`` — a whole TransformerLayer (q|k|v projections, K6 window attention, K7 merge / LayerNorm / FFN /
  residual) as ONE autograd node: inference kernels forward (their weight streams are re-packed on the device after an
  optimizer step: packing.py), mnerf_encoder_layer_backward -> mnerf_window_attention_backward_stats -> mnerf_qkv_backward
* ``window_attention``  — K6 alone with its flash-style HIP backward (no score tensor)

What still back-propagates through PyTorch-ROCm library ops: the CNN backbone, the up-sampler and their InstanceNorms
(convolution dgrad / wgrad on MIOpen) — gmflow.py's ``forward`` of those modules under autograd.

This module is only entered when gradients are required; inference never touches it, and it
is not a fallback: the forward pass still fails loudly without ``libmnerf_hip.so``.
Training-time ray counts are small (``rand_rays_train`` = 1024).
"""
import math
import os

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


def _forward_stats():
    """the training forward publishes the softmax's row statistics for the backward (MNERF_WA_FWD_STATS=0: the backward recomputes
    them in a first pass, the round-4 form before the statistics pair existed)"""
    return hip.wa_math() == hip.WA_PRESPLIT_F16 and os.environ.get("MNERF_WA_FWD_STATS", "1") != "0"


class _WindowAttentionFn(torch.autograd.Function):
    """HIP forward (the inference kernel, publishing the softmax's row statistics) and HIP backward
    (mnerf_window_attention_backward_stats: flash style, the [windows, L_w, L_w] score tensor is never built; fp32-grade
    split-fp16 products (MNERF_WA_BWD_MATH: f16x3 default, bf16x6, or exact f32); deterministic).  MNERF_WA_BACKWARD=torch keeps the round 1-3
    form for comparison: re-evaluation of the op chain with torch ops under autograd (``_window_attention_torch``)."""

    @staticmethod
    def forward(ctx, q, k, v, h, w, splits, shifted):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        stats = torch.empty(2, q.shape[0] * q.shape[1], device=q.device) if _forward_stats() else None
        out = hip.window_attention(q, k, v, h, w, splits, shifted, row_stats=stats)
        ctx.save_for_backward(q, k, v, out)
        ctx.geom, ctx.stats = (h, w, splits, shifted), stats
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v, out = ctx.saved_tensors
        if os.environ.get("MNERF_WA_BACKWARD", "hip") == "torch":
            q, k, v = (t.detach().requires_grad_(True) for t in (q, k, v))
            with torch.enable_grad():
                re = _window_attention_torch(q, k, v, *ctx.geom)
            gq, gk, gv = torch.autograd.grad(re, (q, k, v), grad_out)
        else:
            gq, gk, gv = hip.window_attention_backward(q, k, v, out, grad_out.contiguous(), *ctx.geom, row_stats=ctx.stats)
        return gq, gk, gv, None, None, None, None


def window_attention(q, k, v, h, w, splits, shifted):
    """HIP forward; differentiable when any input requires grad."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _WindowAttentionFn.apply(q, k, v, h, w, splits, shifted)
    return hip.window_attention(q, k, v, h, w, splits, shifted)


# ----------------------------------------------------------------------------- a whole transformer layer


class _TransformerLayerFn(torch.autograd.Function):
    """TransformerLayer.forward (gmflow/transformer.py:147-185) as one autograd node.  Forward: the inference kernels
    (mnerf_qkv_projection, K6 with its row statistics, K7 = mnerf_encoder_block).  Backward, all HIP:
    mnerf_encoder_layer_backward_saved (the chain after the attention from the pre-norm activations the training forward kept:
    merge's output and, with an FFN, mlp.0's pre-GELU output and mlp.2's output; norms and GELU re-evaluated; MNERF_ENC_SAVE=0:
    everything re-evaluated from the attention output and the layer input; GEMMs split-bf16 or, MNERF_GEMM_MATH=f32, exact)
    -> mnerf_window_attention_backward_stats -> mnerf_qkv_backward.
    Saved: the layer's two inputs, q, k, v, the attention output (six [B, h*w, 128] tensors), two floats per token and the
    pre-norm activations ([B h w, 128] + with an FFN [B h w, 1024] + [B h w, 128])."""

    @staticmethod
    def forward(ctx, source, target, geom, layer, *params):
        h, w, splits, shifted = geom
        source, target = source.contiguous(), target.contiguous()
        ws, ews = layer._packed_qkv(source.device)
        q, k, v = hip.qkv_projection(ws, ews, source, target, False)
        b, n, c = source.shape
        # the softmax's row statistics travel to the backward (2 floats per token), which then has no statistics pass
        stats = torch.empty(2, b * n, device=source.device) if _forward_stats() else None
        attn = hip.window_attention(q, k, v, h, w, splits, shifted, row_stats=stats)
        bws, ln, bews = layer._packed_block(source.device)
        # the layer keeps its pre-norm activations - merge's output and, with an FFN, mlp.0's pre-GELU output and mlp.2's output
        # (MNERF_ENC_SAVE=0: the backward re-evaluates them with one / four GEMMs)
        ctx.saved_pre = None
        if os.environ.get("MNERF_ENC_SAVE", "1") != "0":
            out, m1, z1, m2 = hip.encoder_block(attn.reshape(b * n, c), source.reshape(b * n, c), bws, ln, not layer.no_ffn, bews, save=True)
            out = out.reshape(b, n, c)
            ctx.saved_pre = (m1, z1, m2)
        else:
            out = hip.encoder_block(attn.reshape(b * n, c), source.reshape(b * n, c), bws, ln, not layer.no_ffn, bews).reshape(b, n, c)
        ctx.save_for_backward(source, target, q, k, v, attn)
        ctx.stats = stats
        ctx.geom, ctx.layer, ctx.params = geom, layer, params
        return out

    @staticmethod
    def backward(ctx, g_out):
        source, target, q, k, v, attn = ctx.saved_tensors
        layer, params = ctx.layer, ctx.params
        h, w, splits, shifted = ctx.geom
        b, n, c = source.shape
        flat = lambda t: t.reshape(b * n, c)
        # (the kernels ACCUMULATE into the parameter gradients - split-K atomics -: one zero-filled buffer per layer, cut into views,
        # instead of up to ten fill launches)
        need = [p for p, nd in zip(params, ctx.needs_input_grad[4:]) if nd]
        flat_grads = torch.zeros(sum(p.numel() for p in need), device=source.device, dtype=source.dtype)
        grads, off = {}, 0
        for p in need:
            grads[p] = flat_grads[off:off + p.numel()].view_as(p)
            off += p.numel()
        g_attn, g_source = hip.encoder_layer_backward(layer, flat(attn), flat(source), flat(g_out.contiguous()), grads, saved=ctx.saved_pre)
        gq, gk, gv = hip.window_attention_backward(q, k, v, attn, g_attn.reshape(b, n, c), h, w, splits, shifted, row_stats=ctx.stats)
        wq, wk, wv = layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight
        g_xq, g_xkv = hip.qkv_backward(wq, wk, wv, flat(source), flat(target), flat(gq), flat(gk), flat(gv),
                                       grads.get(wq), grads.get(wk), grads.get(wv))
        g_src = (g_source + g_xq).reshape(b, n, c) if ctx.needs_input_grad[0] else None
        g_tgt = g_xkv.reshape(b, n, c) if ctx.needs_input_grad[1] else None
        return (g_src, g_tgt, None, None) + tuple(grads.get(p) for p in params)


def transformer_layer(layer, source, target, h, w, splits, shifted):
    """``layer``: a gmflow.TransformerLayer; source / target [B, h*w, 128] (target = the key / value source, already batch-swapped
    for cross attention).  Differentiable w.r.t. both inputs and every parameter of the layer."""
    params = [layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight, layer.merge.weight, layer.norm1.weight, layer.norm1.bias]
    if not layer.no_ffn:
        params += [layer.mlp[0].weight, layer.mlp[2].weight, layer.norm2.weight, layer.norm2.bias]
    return _TransformerLayerFn.apply(source, target, (h, w, splits, shifted), layer, *params)


# ----------------------------------------------------------------------------- K1..K5


class _ConvFn(torch.autograd.Function):
    """Conv2d(c_in, c_out, k, stride, padding=k//2) of the GMFlow backbone / up-sampler on the training path: forward, data gradient
    and weight gradient on the exact-f32 matrix instruction (csrc/conv_backward.hip) - what autograd ran on MIOpen until round 6
    (backbone.py:6-122, superres.py:5-38 under coach.py:215-243)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.stride, ctx.has_bias = int(stride), bias is not None
        return hip.conv2d_forward_f32(x, weight, bias, stride)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        c_out, c_in, k, _ = weight.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = hip.conv2d_backward_data(dy, weight, x.shape[2], x.shape[3], ctx.stride)
        if ctx.needs_input_grad[1]:
            dw = hip.conv_stem_backward_weight(x, dy) if (c_in, k) == (3, 7) else hip.conv2d_backward_weight(x, dy, k, ctx.stride)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None


class _ConvFn16(torch.autograd.Function):
    """The same node with the FORWARD - and, where the data gradient is itself a convolution the forward kernel builds (stride 1,
    c_in 64 / 96 / 128), the DATA GRADIENT - on the split-fp16 convolution of inference (csrc/conv.hip: three fp16 products per
    MAC at ~4 x the exact-f32 matrix rate), from weight streams packed on the device once per optimizer step
    (packing.ConvPacker).  Stride-2 data gradients and all weight gradients stay on conv_backward.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pack):
        x = x.contiguous()
        c_out, c_in, k, _ = weight.shape
        # the two absmax regions of this node (max |x|, max |dy|): from the owner's per-step block if it handed one out
        # (gmflow._train_packs: ONE zero fill per CNN forward instead of two per convolution)
        regs = pack[3] if len(pack) > 3 and pack[3] is not None else hip.absmax_regions(2, x.device)
        region = regs[0]
        hip.absmax(x, region)
        ws_f, ws_b, ew = pack[:3]
        if (c_in, k) == (3, 7):
            y = hip.conv_stem(x, ws_f, ew, region)
        else:
            y = hip.conv2d(x, ws_f, None if bias is None else bias.detach(), c_in, c_out, k, int(stride), ew, region)
        ctx.save_for_backward(x, weight, regs)
        ctx.stride, ctx.has_bias, ctx.ws_b, ctx.ew = int(stride), bias is not None, ws_b, ew
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, regs = ctx.saved_tensors
        x_region, dy_region = regs[0], regs[1]
        dy = dy.contiguous()
        c_out, c_in, k, _ = weight.shape
        dx = dw = db = None
        hip.absmax(dy, dy_region)  # max |dy|: the gain of the split-fp16 operands of both gradients
        if ctx.needs_input_grad[0]:
            if ctx.ws_b is not None:
                dx = hip.conv2d(dy, ctx.ws_b, None, c_out, c_in, k, 1, ctx.ew, dy_region)
            else:
                dx = hip.conv2d_backward_data(dy, weight, x.shape[2], x.shape[3], ctx.stride)
        if ctx.needs_input_grad[1]:
            if (c_in, k) == (3, 7):
                dw = hip.conv_stem_backward_weight(x, dy)
            elif os.environ.get("MNERF_TRAIN_WGRAD", "f16x3") == "f32":
                dw = hip.conv2d_backward_weight(x, dy, k, ctx.stride)
            else:
                dw = hip.conv2d_backward_weight(x, dy, k, ctx.stride, x_region, dy_region)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None, None


def conv2d(conv, x):
    """``conv`` (an nn.Conv2d of a shape conv_backward.hip builds) applied to x [N,C,H,W] as one autograd node of HIP kernels; with
    a ``_mnerf_train_pack`` on the module (gmflow: set per optimizer step) the split-fp16 kernels serve forward and data gradient"""
    pack = getattr(conv, "_mnerf_train_pack", None)
    if pack is not None:
        return _ConvFn16.apply(x, conv.weight, conv.bias, conv.stride[0], pack)
    return _ConvFn.apply(x, conv.weight, conv.bias, conv.stride[0])


def conv2d_supported(conv):
    k, s, ci, co = conv.kernel_size[0], conv.stride[0], conv.in_channels, conv.out_channels
    return (co % 32 == 0 and co <= 128 and conv.padding[0] == k // 2 and s in (1, 2) and conv.dilation[0] == 1 and conv.groups == 1
            and (((k in (1, 3)) and ci % 32 == 0 and ci <= 128) or (k, ci, co, s) == (7, 3, 64, 2)))


class _InstanceNormFn(torch.autograd.Function):
    """[relu](F.instance_norm(x)) (no affine, eps 1e-5): the fused forward kernel of inference, and a backward that re-derives the
    plane statistics from x (csrc/instance_norm.hip)."""

    @staticmethod
    def forward(ctx, x, relu):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.relu = bool(relu)
        return hip.instance_norm(x, relu_inner=relu)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return hip.instance_norm_backward(x, dy.contiguous(), ctx.relu), None


def instance_norm(x, relu=False):
    return _InstanceNormFn.apply(x, relu)


class _UpBilinearAddFn(torch.autograd.Function):
    """F.interpolate(right, scale_factor=2, mode="bilinear", align_corners=False) + left (superres.py:37) as one kernel each way"""

    @staticmethod
    def forward(ctx, right, left):
        return hip.upsample_bilinear2x(right.contiguous(), left.contiguous())

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return (hip.upsample_bilinear2x_backward(g) if ctx.needs_input_grad[0] else None), (g if ctx.needs_input_grad[1] else None)


def upsample_bilinear2x_add(right, left):
    return _UpBilinearAddFn.apply(right, left)


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
    backward: mnerf_composite_backward -> mnerf_decoder_backward (the conditional MLP + ray transformer re-evaluated from the
              saved conditioning rows and differentiated: gradients of the decoder parameters and of the rows) ->
              mnerf_cost_volume_backward (scatters the rows' gradient into the feature maps).  All HIP."""

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
        dirs = (F.normalize(ray, dim=-1) @ torch.as_tensor(launch.view0_extr, device=ray.device)[:, :3].t()).contiguous()
        named = dict(dec_m.named_parameters())
        table = None
        if opt.decoder.raytrans_posenc:
            from .cond_nerf import raytrans_table
            table = torch.from_numpy(raytrans_table(s))
        g_cond, g_named = hip.decoder_backward(
            opt, {k: named[k].detach() for k in hip.DEC_TRAIN_TENSORS}, launch.n_views, x_ndc.reshape(r * s, 3).contiguous(), dirs,
            cond, dec.cond_stride, g_rgb_s.contiguous(), g_sigma.contiguous(), want_g_cond=any(ctx.needs_input_grad[2:2 + n_feat]),
            raytrans_table=table, grads={k: None for k in hip.DEC_TRAIN_TENSORS if not named[k].requires_grad})
        g_feats = [None] * n_feat
        if g_cond is not None:
            g_feats = hip.cost_volume_backward(sc, rays, dec.cond_stride, g_cond, [torch.zeros_like(f) for f in feats])
        by_param = {id(named[k]): g_named.get(k) for k in hip.DEC_TRAIN_TENSORS}
        g_params = [by_param.get(id(p)) for p in dec_m.parameters()]
        return (None, None, *g_feats, *g_params)


def render_ray_chunk(launch, feats):
    """Differentiable render of one chunk of rays of one batch element -> (rgb [R,3], depth [R,1], opacity [R,1])."""
    params = list(launch.dec_module.parameters())
    return _RayChunkFn.apply(launch, len(feats), *feats, *params)
