"""Deterministic synthetic scenes and weights for parity tests, goldens and bench.py.

No dataset or checkpoint of the reference is available offline, so
every measurement and parity check runs on inputs generated here:

* ``make_scene``  – a posed multi-view batch with the batch contract of the reference
  datasets (``images``, ``extrinsics`` world->cam 4x4, ``intrinsics`` 3x3, ``near_fars``;
  target view LAST — /root/reference/models/matchnerf.py:75-86, datasets/dtu.py:186-209).
* ``state_dict_spec`` / ``seeded_state_dict`` – the 153-tensor parameter layout of
  ``MatchNeRF`` filled from a numpy PCG64 stream, with non-zero
  biases / LayerNorm affine so that every term of the path is exercised.

Everything is numpy-only so that the generator is identical in the build container
(golden generation through the imported reference) and on the GPU box.
"""
from collections import OrderedDict

import numpy as np

# ----------------------------------------------------------------------------- scenes


def _lowpass_noise(rng, c, h, w, passes=3):
    img = rng.random((c, h, w), dtype=np.float64)
    for _ in range(passes):  # separable [1 2 1]/4 blur, wrap padding (keeps U[0,1)-ish range)
        img = (np.roll(img, 1, 1) + 2 * img + np.roll(img, -1, 1)) / 4
        img = (np.roll(img, 1, 2) + 2 * img + np.roll(img, -1, 2)) / 4
    lo, hi = img.min(), img.max()
    return ((img - lo) / (hi - lo + 1e-12)).astype(np.float32)


def _look_at_w2c(center, target, up=(0.0, -1.0, 0.0)):
    """OpenCV-style camera (x right, y down, z forward); returns world->cam 4x4 (float64)."""
    center = np.asarray(center, np.float64)
    z = np.asarray(target, np.float64) - center
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, np.float64))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    rot = np.stack([x, y, z], 0)  # rows = camera axes in world coords
    w2c = np.eye(4)
    w2c[:3, :3] = rot
    w2c[:3, 3] = -rot @ center
    return w2c


def make_scene(height, width, n_src_views=3, seed=0, focal_scale=2.26, baseline=0.6,
               near_far=(2.125, 4.525), batch_size=1, wide=False):
    """Seeded synthetic batch (numpy arrays, float32) following SURVEY.md §8(d).

    Source cameras sit on a line/arc (V<=5) or a 2xN grid (V>5) ``baseline`` apart and
    converge at mid-depth; the target camera sits between the first two sources with a small
    vertical offset.  ``wide=True`` (Blender-like) widens the baseline so that many samples
    project outside the source frusta (exercises border clamp + visibility mask).
    """
    rng = np.random.default_rng(seed)
    v_all = n_src_views + 1
    near, far = near_far
    mid = 0.5 * (near + far)
    target_pt = np.array([0.0, 0.0, mid])
    b = baseline * (2.5 if wide else 1.0)
    if n_src_views <= 5:
        xs = (np.arange(n_src_views) - (n_src_views - 1) / 2.0) * b
        centres = [np.array([x, 0.05 * b * ((i % 2) * 2 - 1), 0.02 * i]) for i, x in enumerate(xs)]
    else:
        cols = (n_src_views + 1) // 2
        centres = []
        for i in range(n_src_views):
            r, c = divmod(i, cols)
            centres.append(np.array([(c - (cols - 1) / 2.0) * b, (r - 0.5) * b, 0.01 * i]))
    tgt_centre = 0.5 * (centres[0] + centres[1]) + np.array([0.07 * b, 0.11 * b, -0.03])
    centres = centres + [tgt_centre]

    images = np.stack([
        np.stack([_lowpass_noise(rng, 3, height, width) for _ in range(v_all)], 0)
        for _ in range(batch_size)], 0)
    extr = np.zeros((batch_size, v_all, 4, 4), np.float32)
    intr = np.zeros((batch_size, v_all, 3, 3), np.float32)
    nfs = np.zeros((batch_size, v_all, 2), np.float32)
    for bi in range(batch_size):
        for vi, c in enumerate(centres):
            jitter = rng.normal(0, 0.01, 3) if bi else 0.0
            extr[bi, vi] = _look_at_w2c(c + jitter, target_pt).astype(np.float32)
            f = focal_scale * width * (1.0 + 0.01 * vi)
            intr[bi, vi] = np.array([[f, 0, width / 2.0 + 0.3 * vi],
                                     [0, f * 1.003, height / 2.0 - 0.2 * vi],
                                     [0, 0, 1]], np.float32)
            nfs[bi, vi] = (near * (1 + 0.002 * vi), far * (1 - 0.001 * vi))
    return dict(images=images, extrinsics=extr, intrinsics=intr, near_fars=nfs)


# ----------------------------------------------------------------------------- weights


def state_dict_spec(n_src_views=3, cos_n_group=(2, 8), net_width=128, net_depth=6, skip=(4,),
                    L_3D=10, L_view=0, num_transformer_layers=6, upsample_factor=2,
                    feature_channels=128, ffn_dim_expansion=4):
    """name -> shape, in ``MatchNeRF.state_dict()`` order (SURVEY.md Appendix B;
    /root/reference/models/gmflow/backbone.py:39-99, transformer.py:108-277,
    superres.py:9-25, rfdecoder/cond_nerf.py:15-45)."""
    C = feature_channels
    spec = OrderedDict()
    bb = "feat_enc.backbone."
    spec[bb + "conv1.weight"] = (64, 3, 7, 7)

    def res_block(prefix, cin, cout, stride):
        spec[prefix + "conv1.weight"] = (cout, cin, 3, 3)
        spec[prefix + "conv2.weight"] = (cout, cout, 3, 3)
        if stride != 1 or cin != cout:
            spec[prefix + "downsample.0.weight"] = (cout, cin, 1, 1)
            spec[prefix + "downsample.0.bias"] = (cout,)

    res_block(bb + "layer1.0.", 64, 64, 1)
    res_block(bb + "layer1.1.", 64, 64, 1)
    res_block(bb + "layer2.0.", 64, 96, 2)
    res_block(bb + "layer2.1.", 96, 96, 1)
    res_block(bb + "layer3.0.", 96, 128, 2)
    res_block(bb + "layer3.1.", 128, 128, 1)
    spec[bb + "conv2.weight"] = (C, 128, 1, 1)
    spec[bb + "conv2.bias"] = (C,)

    tr = "feat_enc.transformer.layers."
    for i in range(num_transformer_layers):
        for blk, ffn in (("self_attn", False), ("cross_attn_ffn", True)):
            p = f"{tr}{i}.{blk}."
            for n in ("q_proj", "k_proj", "v_proj", "merge"):
                spec[p + n + ".weight"] = (C, C)
            spec[p + "norm1.weight"] = (C,)
            spec[p + "norm1.bias"] = (C,)
            if ffn:
                spec[p + "mlp.0.weight"] = (2 * C * ffn_dim_expansion, 2 * C)
                spec[p + "mlp.2.weight"] = (C, 2 * C * ffn_dim_expansion)
                spec[p + "norm2.weight"] = (C,)
                spec[p + "norm2.bias"] = (C,)

    n_blocks = int(np.log2(upsample_factor))
    fu = "feat_enc.featup_net."
    for i in range(n_blocks):
        spec[f"{fu}conv_ls.{i}.weight"] = (C, C, 3, 3)
        spec[f"{fu}conv_ls.{i}.bias"] = (C,)
    for i in range(n_blocks + 1):
        spec[f"{fu}conv_l2rs.{i}.weight"] = (C, C, 3, 3)
        spec[f"{fu}conv_l2rs.{i}.bias"] = (C,)

    W = net_width
    d3 = 3 + 6 * L_3D
    dv = 3 + 6 * L_view
    cond = sum(cos_n_group) + 4 * n_src_views
    nd = "nerf_dec."
    for i in range(net_depth):
        cin = d3 if i == 0 else (W + d3 if (i - 1) in skip else W)
        spec[f"{nd}pts_linears.{i}.weight"] = (W, cin)
        spec[f"{nd}pts_linears.{i}.bias"] = (W,)
    spec[nd + "pts_bias.weight"] = (W, cond)
    spec[nd + "pts_bias.bias"] = (W,)
    spec[nd + "views_linears.0.weight"] = (W // 2, W + dv)
    spec[nd + "views_linears.0.bias"] = (W // 2,)
    spec[nd + "alpha_linear.0.weight"] = (16, W)
    spec[nd + "alpha_linear.0.bias"] = (16,)
    for n in ("w_qs", "w_ks", "w_vs", "fc"):
        spec[f"{nd}ray_attention.{n}.weight"] = (16, 16)
    spec[nd + "ray_attention.layer_norm.weight"] = (16,)
    spec[nd + "ray_attention.layer_norm.bias"] = (16,)
    spec[nd + "out_alpha_linear.0.weight"] = (16, 16)
    spec[nd + "out_alpha_linear.0.bias"] = (16,)
    spec[nd + "out_alpha_linear.2.weight"] = (1, 16)
    spec[nd + "out_alpha_linear.2.bias"] = (1,)
    spec[nd + "feature_linear.weight"] = (W, W)
    spec[nd + "feature_linear.bias"] = (W,)
    spec[nd + "rgb_linear.weight"] = (3, W // 2)
    spec[nd + "rgb_linear.bias"] = (3,)
    return spec


def seeded_state_dict(spec, seed=1):
    """Fill ``spec`` deterministically (numpy arrays, float32).

    Scales follow the reference's own initialisers in spirit (kaiming for conv / decoder
    linears, xavier for the transformer: backbone.py:83-90, transformer.py:275-277,
    cond_nerf.py:102-106) but biases and norm affines are made non-trivial so that
    goldens pin every term.  The density head is biased positive so that opacity is not
    degenerate under random weights.
    """
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in spec.items():
        leaf = name.rsplit(".", 1)[-1]
        is_norm = ".norm1." in name or ".norm2." in name or "layer_norm" in name
        if is_norm and leaf == "weight":
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif is_norm and leaf == "bias":
            a = 0.05 * rng.standard_normal(shape)
        elif leaf == "bias":
            a = 0.05 * rng.standard_normal(shape)
            if name.endswith("pts_bias.bias"):
                a = a + 1.0  # FiLM multiplier centred near 1 keeps activations alive
            if name.endswith("out_alpha_linear.2.bias"):
                a = a + 0.26
        elif len(shape) == 4:  # conv: kaiming normal, fan_out
            fan_out = shape[0] * shape[2] * shape[3]
            a = rng.standard_normal(shape) * np.sqrt(2.0 / fan_out)
        elif "transformer" in name:  # xavier uniform
            bound = np.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-bound, bound, shape)
        elif name.endswith("pts_bias.weight"):
            a = rng.standard_normal(shape) * 0.5 / np.sqrt(shape[1])
        elif name.endswith("out_alpha_linear.2.weight"):
            a = rng.standard_normal(shape) * 0.15 * np.sqrt(2.0 / shape[1])
        else:  # decoder linears: kaiming normal, fan_in
            a = rng.standard_normal(shape) * np.sqrt(2.0 / shape[1])
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def to_torch(d, device="cpu"):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in d.items()}
