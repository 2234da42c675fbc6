"""CPU ORACLE for the MatchNeRF per-ray rendering hot path  —  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 torch ops on the CPU, the algorithm of the reference's
hot path.  It exists to *check* the HIP path; it is never the thing that is
shipped or measured as the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package ``matchnerf_amd``
must not import anything from ``oracle/`` and fails loudly when the HIP library is missing.

Pinning: every function below is compared against outputs of the *imported reference itself*
(run in the build container by ``tools/gen_golden.py``), committed as small fixtures under
``tests/golden/`` — see ``tests/test_oracle_golden.py``.  The reference ships no tests or golden
vectors of its own, and its published PSNR table needs datasets/checkpoints that
are not available offline, so for real-data PSNR parity is UNPINNED; for synthetic seeded inputs
it is pinned by those fixtures.

The code is written independently (explicit bilinear taps instead of ``grid_sample``, pair-major
cost volume instead of per-view channel chunks, index-based shifted windows instead of
roll/split/merge) so that agreement with the reference is evidence, not tautology.
Each function cites the reference lines it follows (paths relative to /root/reference).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# =============================================================================== options


class OracleConfig:
    """The option fields the hot path reads (configs/base.yaml:10-51, test.yaml:7-8)."""

    def __init__(self, n_src_views=3, attn_splits=2, cos_n_group=(2, 8), num_transformer_layers=6,
                 upsample_factor=2, wo_self_attn=False, net_width=128, net_depth=6, skip=(4,),
                 L_3D=10, L_view=0, raytrans_posenc=False, density_maskfill=False,
                 raytrans_act="ReLU", legacy_coord=True, wo_render_interval=True,
                 depth_param="metric", sample_intvs=64):
        self.n_src_views = n_src_views
        self.attn_splits = attn_splits
        self.cos_n_group = tuple(cos_n_group)
        self.num_transformer_layers = num_transformer_layers
        self.upsample_factor = upsample_factor
        self.wo_self_attn = wo_self_attn
        self.net_width = net_width
        self.net_depth = net_depth
        self.skip = tuple(skip)
        self.L_3D = L_3D
        self.L_view = L_view
        self.raytrans_posenc = raytrans_posenc
        self.density_maskfill = density_maskfill
        self.raytrans_act = raytrans_act
        self.legacy_coord = legacy_coord
        self.wo_render_interval = wo_render_interval
        self.depth_param = depth_param
        self.sample_intvs = sample_intvs

    @classmethod
    def from_opts(cls, opts):
        enc, dec, nerf = opts["encoder"], opts["decoder"], opts["nerf"]
        splits = enc["attn_splits_list"]
        return cls(n_src_views=opts["n_src_views"], attn_splits=splits[0] if isinstance(splits, (list, tuple)) else splits,
                   cos_n_group=enc["cos_n_group"], num_transformer_layers=enc["num_transformer_layers"],
                   upsample_factor=enc["upsample_factor"], wo_self_attn=enc["wo_self_attn"],
                   net_width=dec["net_width"], net_depth=dec["net_depth"], skip=dec["skip"],
                   L_3D=dec["posenc"]["L_3D"], L_view=dec["posenc"]["L_view"],
                   raytrans_posenc=dec["raytrans_posenc"], density_maskfill=dec["density_maskfill"],
                   raytrans_act=dec["raytrans_act"], legacy_coord=nerf["legacy_coord"],
                   wo_render_interval=nerf["wo_render_interval"], depth_param=nerf["depth"]["param"],
                   sample_intvs=nerf["sample_intvs"])


def pair_list(n_views):
    """Ordered view pairs (a<b), lexicographic (models/gmflow/gmflow.py:49, matchnerf.py:194)."""
    return [(a, b) for a in range(n_views - 1) for b in range(a + 1, n_views)]


# =============================================================================== camera (a8-a10)


def target_rays(height, width, extr_t, intr_t, legacy=True):
    """Ray origin and (un-normalised) direction of every target pixel, row-major.

    misc/camera.py:255-278 (get_center_and_ray), :221-222 (img2cam), :231-240
    (cam2world_legacy: 4x4 inverse taken in float64, cast to float32) and :225-228 / :36-42
    for the non-legacy pose inverse.  extr_t [3,4] world->cam, intr_t [3,3].
    Returns center [HW,3], ray [HW,3].
    """
    off = 0.0 if legacy else 0.5
    ys = torch.arange(height, dtype=torch.float32) + off
    xs = torch.arange(width, dtype=torch.float32) + off
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    pix = torch.stack([gx.reshape(-1), gy.reshape(-1), torch.ones(height * width)], -1)  # [HW,3]
    cam = pix @ intr_t.inverse().t()
    if legacy:
        sq = torch.eye(4)
        sq[:3] = extr_t
        c2w = sq.double().inverse()[:3].float()
    else:
        # Pose.invert: R_inv = R^T, t_inv = -R_inv @ t  (misc/camera.py:36-42)
        rot_inv = extr_t[:, :3].t()
        c2w = torch.cat([rot_inv, -(rot_inv @ extr_t[:, 3:])], 1)
    cam_h = torch.cat([cam, torch.ones_like(cam[:, :1])], -1)
    grid_world = cam_h @ c2w.t()
    zero_h = torch.cat([torch.zeros_like(cam), torch.ones_like(cam[:, :1])], -1)
    center = zero_h @ c2w.t()
    return center, grid_world - center


def depth_samples(cfg, near, far, n_rays, stratified_u=None):
    """models/matchnerf.py:163-181.  Legacy: d_i = near + i/(S-1)*(far-near); otherwise
    (i+0.5)/S.  ``stratified_u`` ([R,S] uniforms) replaces the 0 / 0.5 shift in train mode."""
    s = cfg.sample_intvs
    shift = 0.0 if cfg.legacy_coord else 0.5
    denom = (s - 1) if cfg.legacy_coord else s
    t = torch.arange(s, dtype=torch.float32)[None, :].expand(n_rays, s)
    t = t + (stratified_u if stratified_u is not None else shift)
    d = t / denom * (far - near) + near
    if cfg.depth_param == "inverse":
        d = 1 / (d + 1e-8)
    return d  # [R,S]


def project_to_view(pts, extr, intr, width, height, near, far):
    """misc/camera.py:351-379 (get_coord_ref_ndc) + :212-214 (world2cam).
    pts [...,3] -> (u, v, z) with u = q_x/q_z/(W-1), v = q_y/q_z/(H-1), z=(q_z-near)/(far-near)."""
    hom = torch.cat([pts, torch.ones_like(pts[..., :1])], -1)
    cam = hom @ extr.t()            # [...,3]   extr [3,4]
    q = cam @ intr.t()
    u = q[..., 0] / q[..., 2] / (width - 1)
    v = q[..., 1] / q[..., 2] / (height - 1)
    z = (q[..., 2] - near) / (far - near)
    return torch.stack([u, v, z], -1)


# =============================================================================== K1 + K2 (a11)


def bilinear_border(fmap, gx, gy):
    """Bilinear lookup with padding_mode='border', align_corners=True, written out as four
    explicit taps (what F.grid_sample does at matchnerf.py:245 and gmflow/utils.py:133-134).
    fmap [C,h,w]; gx, gy normalised to [-1,1] with arbitrary leading shape N... -> [C,N...]."""
    c, h, w = fmap.shape
    x = ((gx + 1) / 2) * (w - 1)
    y = ((gy + 1) / 2) * (h - 1)
    x = x.clamp(0, w - 1)
    y = y.clamp(0, h - 1)
    x0 = x.floor()
    y0 = y.floor()
    fx = x - x0
    fy = y - y0
    x0i = x0.long()
    y0i = y0.long()
    x1i = (x0i + 1).clamp(max=w - 1)  # weight is exactly 0 whenever the clamp acts
    y1i = (y0i + 1).clamp(max=h - 1)
    flat = fmap.reshape(c, h * w)

    def tap(yi, xi):
        return flat[:, (yi * w + xi).reshape(-1)].reshape(c, *gx.shape)

    return (tap(y0i, x0i) * ((1 - fx) * (1 - fy)) + tap(y0i, x1i) * (fx * (1 - fy)) +
            tap(y1i, x0i) * ((1 - fx) * fy) + tap(y1i, x1i) * (fx * fy))


def group_cosine(a, b, n_group, eps=1e-8):
    """torch.nn.CosineSimilarity(dim=channel-within-group) with each norm clamped at eps
    separately (installed torch 2.10 semantics).  a, b [C,N...] -> [G,N...]."""
    c = a.shape[0]
    ag = a.reshape(n_group, c // n_group, *a.shape[1:])
    bg = b.reshape(n_group, c // n_group, *b.shape[1:])
    na = ag.norm(dim=1).clamp_min(eps)
    nb = bg.norm(dim=1).clamp_min(eps)
    return ((ag / na[:, None]) * (bg / nb[:, None])).sum(1)


def cost_volume_cond(cfg, pts, src_extr, src_intr, src_nf, src_images, pair_feats, height, width):
    """The conditioning vector of every 3D sample (models/matchnerf.py:209-293).

    pts [R,S,3]; src_* per source view (extr [V,3,4], intr [V,3,3], nf [V,2], images [V,3,H,W]);
    ``pair_feats`` = list over scales of (f0 [P,C,h,w], f1 [P,C,h,w]) — the pair-specific
    GMFlow features (first/second member of pair (a,b)).  The reference stores them as
    per-view channel chunks (matchnerf.py:192-205) and pairs chunk j of view i with chunk i of
    view j+1 (matchnerf.py:260-268): that is exactly f0[p] sampled at view a versus f1[p]
    sampled at view b for p=(a,b), which is how it is written here.
    Returns cond [R,S, sum(G)+4V] = [cos@scale0, cos@scale1, rgb_v0.., mask_v0..]
    (cond_nerf.py:59) and the per-view masks [R,S,V]."""
    v_n = src_extr.shape[0]
    pairs = pair_list(v_n)
    grids, colors, masks = [], [], []
    for v in range(v_n):
        uvz = project_to_view(pts, src_extr[v], src_intr[v], width, height, src_nf[v, 0], src_nf[v, 1])
        g = uvz[..., :2] * 2.0 - 1.0
        grids.append(g)
        colors.append(bilinear_border(src_images[v], g[..., 0], g[..., 1]))          # [3,R,S]
        inside = (g[..., 0] > -1.0) & (g[..., 0] < 1.0) & (g[..., 1] > -1.0) & (g[..., 1] < 1.0)
        masks.append(inside.float())
    feats = []
    for scale, (f0, f1) in enumerate(pair_feats):
        acc = 0
        for p, (a, b) in enumerate(pairs):
            fa = bilinear_border(f0[p], grids[a][..., 0], grids[a][..., 1])
            fb = bilinear_border(f1[p], grids[b][..., 0], grids[b][..., 1])
            acc = acc + group_cosine(fa, fb, cfg.cos_n_group[scale])
        feats.append(acc / len(pairs))                                               # [G,R,S]
    feat = torch.cat(feats, 0).permute(1, 2, 0)
    color = torch.cat(colors, 0).permute(1, 2, 0)
    mask = torch.stack(masks, -1)
    return torch.cat([feat, color, mask], -1), mask


# =============================================================================== K3 + K4 (a12, a13)


POSENC_EXACT_ARGS = False  # see posenc_3d (non-legacy coordinates)


def posenc_3d(cfg, x, L):
    """Legacy: cond_nerf.py:108-116 (freq 2^l, no pi, layout [sin(l-major,c) | cos]);
    non-legacy: nerf.py:126-133 (freq 2^l*pi, layout [c][sin|cos][l])."""
    if L == 0:
        return x
    freq = 2.0 ** torch.arange(L, dtype=torch.float32)
    if cfg.legacy_coord:
        spec = (x[..., None, :] * freq[:, None]).reshape(*x.shape[:-1], -1)          # [.., L*3]
        enc = torch.cat([spec.sin(), spec.cos()], -1)
    else:
        # The ARGUMENT is formed in float32 whatever the dtype of x: that is what the reference does (nerf.py:128: a float32
        # frequency tensor 2^l * pi times float32 points; up to 512 pi ~ 1600 rad, so its rounding is 1e-4 rad), and it is the
        # quantity the HIP kernels reproduce bit for bit.  A float64 evaluation of the network (the judge of the fp32-grade
        # matrix paths and of the backward kernels) must see the SAME arguments: with a float64 product the encoding moves by
        # 1e-4, a ReLU flips here and there, and weight gradients of the first layers jump by 1e-2 — discontinuities of the
        # network, not errors of a kernel.  (float32 inputs: unchanged, the casts are no-ops.)
        # POSENC_EXACT_ARGS = True restores the plain evaluation in x's own dtype (the independent float64 reference of round 3);
        # tests/test_oracle_golden.py bounds the difference between the two forms, so the alignment is explicit and measured.
        if POSENC_EXACT_ARGS:
            spec = x[..., None] * (freq * math.pi).to(x.dtype)
        else:
            spec = (x[..., None].to(freq.dtype) * (freq * math.pi)).to(x.dtype)      # [.., 3, L]
        enc = torch.stack([spec.sin(), spec.cos()], -2).reshape(*x.shape[:-1], -1)
    return torch.cat([x, enc], -1)


def raytrans_table(n_samples, d_hid=16):
    """cond_nerf.py:118-127: pos/10000^(2*(j//2)/d), sin on even j, cos on odd j."""
    pos = np.arange(n_samples, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    ang = pos / np.power(10000, 2 * (j // 2) / d_hid)
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(ang).float()


def _act(name, x):
    return {"ReLU": F.relu, "ELU": F.elu}[name](x)


def ray_attention(sd, a, valid):
    """models/rfdecoder/ray_transformer.py:29-79 for one batch of rays.
    a [R,S,16]; valid [R,S] (1 where the query row is kept).  4 heads x 4, temperature 2,
    masked *query rows* filled with -1e9 (=> uniform attention), residual, LayerNorm eps 1e-6."""
    p = "nerf_dec.ray_attention."
    r, s, _ = a.shape
    q = (a @ sd[p + "w_qs.weight"].t()).reshape(r, s, 4, 4).permute(0, 2, 1, 3)
    k = (a @ sd[p + "w_ks.weight"].t()).reshape(r, s, 4, 4).permute(0, 2, 1, 3)
    v = (a @ sd[p + "w_vs.weight"].t()).reshape(r, s, 4, 4).permute(0, 2, 1, 3)
    scores = (q / 2.0) @ k.transpose(-1, -2)                                         # [R,4,S,S]
    scores = torch.where(valid[:, None, :, None] > 0, scores, torch.full_like(scores, -1e9))
    o = torch.softmax(scores, -1) @ v                                                # [R,4,S,4]
    o = o.permute(0, 2, 1, 3).reshape(r, s, 16) @ sd[p + "fc.weight"].t() + a
    return F.layer_norm(o, (16,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps=1e-6)


def decoder(cfg, sd, x_ref, dir_ref, cond, mask):
    """models/rfdecoder/cond_nerf.py:52-100.  x_ref [R,S,3] (u,v,z w.r.t. source view 0),
    dir_ref [R,3] unit view direction in source-0 camera frame, cond [R,S,Dc], mask [R,S,V].
    Returns rgb [R,S,3], sigma [R,S]."""
    nd = "nerf_dec."
    enc = posenc_3d(cfg, x_ref, cfg.L_3D)
    film = cond @ sd[nd + "pts_bias.weight"].t() + sd[nd + "pts_bias.bias"]
    h = enc
    for i in range(cfg.net_depth):
        h = F.relu((h @ sd[f"{nd}pts_linears.{i}.weight"].t() + sd[f"{nd}pts_linears.{i}.bias"]) * film)
        if i in cfg.skip:
            h = torch.cat([enc, h], -1)
    a = _act(cfg.raytrans_act, h @ sd[nd + "alpha_linear.0.weight"].t() + sd[nd + "alpha_linear.0.bias"])
    if cfg.raytrans_posenc:
        a = a + raytrans_table(a.shape[1])[None]
    n_valid = mask.sum(-1)                                                           # [R,S]
    o = ray_attention(sd, a, (n_valid > 1).float())
    o = _act(cfg.raytrans_act, o @ sd[nd + "out_alpha_linear.0.weight"].t() + sd[nd + "out_alpha_linear.0.bias"])
    sigma = F.relu(o @ sd[nd + "out_alpha_linear.2.weight"].t() + sd[nd + "out_alpha_linear.2.bias"])[..., 0]
    if cfg.density_maskfill:
        sigma = torch.where(n_valid < 1, torch.zeros_like(sigma), sigma)
    feat = h @ sd[nd + "feature_linear.weight"].t() + sd[nd + "feature_linear.bias"]
    d = dir_ref[:, None, :].expand(-1, x_ref.shape[1], -1)
    d_enc = posenc_3d(cfg, d, cfg.L_view) if cfg.L_view > 0 else d
    hv = F.relu(torch.cat([feat, d_enc], -1) @ sd[nd + "views_linears.0.weight"].t() + sd[nd + "views_linears.0.bias"])
    rgb = torch.sigmoid(hv @ sd[nd + "rgb_linear.weight"].t() + sd[nd + "rgb_linear.bias"])
    return rgb, sigma


# =============================================================================== K5 (a14)


def composite(cfg, ray, rgb, sigma, depth, setbg_opaque=False):
    """models/rfdecoder/nerf.py:101-124.  ray [R,3], rgb [R,S,3], sigma [R,S], depth [R,S]
    -> rgb [R,3], depth [R,1], opacity [R,1], prob [R,S,1] (the reference's four return values)."""
    if cfg.wo_render_interval:
        sd_ = sigma
    else:
        intv = torch.cat([depth[:, 1:] - depth[:, :-1], torch.full_like(depth[:, :1], 1e10)], 1)
        sd_ = sigma * intv * ray.norm(dim=-1, keepdim=True)
    alpha = 1 - torch.exp(-sd_)
    excl = torch.cat([torch.zeros_like(sd_[:, :1]), sd_[:, :-1]], 1).cumsum(1)
    w = torch.exp(-excl) * alpha
    out_rgb = (rgb * w[..., None]).sum(1)
    out_depth = (depth * w).sum(1, keepdim=True)
    opacity = w.sum(1, keepdim=True)
    if setbg_opaque:
        out_rgb = out_rgb + (1 - opacity)
    return out_rgb, out_depth, opacity, w[..., None]


# =============================================================================== a7 render


def render_rays(cfg, sd, ray_idx, tgt_extr, tgt_intr, tgt_nf, src_extr, src_intr, src_nf,
                src_images, pair_feats, setbg_opaque=False, stratified_u=None, return_stages=False):
    """models/matchnerf.py:88-143 for one batch element and a set of target-pixel indices."""
    v_n, _, height, width = src_images.shape
    center, ray = target_rays(height, width, tgt_extr, tgt_intr, cfg.legacy_coord)
    center, ray = center[ray_idx], ray[ray_idx]
    d = depth_samples(cfg, tgt_nf[0], tgt_nf[1], ray.shape[0], stratified_u)
    pts = center[:, None] + ray[:, None] * d[..., None]
    cond, mask = cost_volume_cond(cfg, pts, src_extr, src_intr, src_nf, src_images, pair_feats, height, width)
    x_ref = project_to_view(pts, src_extr[0], src_intr[0], width, height, src_nf[0, 0], src_nf[0, 1])
    dir_ref = F.normalize(ray, dim=-1) @ src_extr[0][:, :3].t()
    rgb_s, sigma = decoder(cfg, sd, x_ref, dir_ref, cond, mask)
    rgb, depth, opacity, _ = composite(cfg, ray, rgb_s, sigma, d, setbg_opaque)
    if return_stages:
        return dict(rgb=rgb, depth=depth, opacity=opacity, cond=cond, x_ref=x_ref, dir_ref=dir_ref,
                    rgb_samples=rgb_s, sigma=sigma, depth_samples=d, ray=ray, center=center)
    return rgb, depth, opacity


# =============================================================================== encoder (a3-a6)


def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


def backbone(sd, x):
    """models/gmflow/backbone.py:6-36, 101-122 with num_scales=1 (stride 8, 128 ch)."""
    p = "feat_enc.backbone."
    x = F.relu(_inorm(F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)))
    for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        for blk in (0, 1):
            q = f"{p}{layer}.{blk}."
            s = stride if blk == 0 else 1
            y = F.relu(_inorm(F.conv2d(x, sd[q + "conv1.weight"], stride=s, padding=1)))
            y = F.relu(_inorm(F.conv2d(y, sd[q + "conv2.weight"], padding=1)))
            if q + "downsample.0.weight" in sd:
                x = _inorm(F.conv2d(x, sd[q + "downsample.0.weight"], sd[q + "downsample.0.bias"], stride=s))
            x = F.relu(x + y)
    return F.conv2d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"])


def sine_position(h, w, channels=128):
    """models/gmflow/position.py:26-47 -> [channels,h,w] ([pos_y(64) | pos_x(64)])."""
    npf = channels // 2
    y = (torch.arange(1, h + 1, dtype=torch.float32) / (h + 1e-6) * (2 * math.pi))[:, None].expand(h, w)
    x = (torch.arange(1, w + 1, dtype=torch.float32) / (w + 1e-6) * (2 * math.pi))[None, :].expand(h, w)
    i = torch.arange(npf, dtype=torch.float32)
    dim_t = 10000.0 ** (2 * torch.div(i, 2, rounding_mode="trunc") / npf)

    def emb(t):
        a = t[..., None] / dim_t
        return torch.stack([a[..., 0::2].sin(), a[..., 1::2].cos()], -1).flatten(-2)

    return torch.cat([emb(y), emb(x)], -1).permute(2, 0, 1)


def add_window_position(feat, splits):
    """models/gmflow/utils.py:68-88: the sine PE of the *window* shape is tiled over all windows."""
    _, c, h, w = feat.shape
    pe = sine_position(h // splits, w // splits, c)
    return feat + pe.repeat(1, splits, splits)[None]


def window_attention(q, k, v, h, w, splits, shifted):
    """models/gmflow/transformer.py:8-16, 19-43, 46-105 restated per token.

    q,k,v [B,h*w,C].  Token (y,x) belongs to window ((y-sy) mod h // wh, (x-sx) mod w // ww)
    when ``shifted`` (the reference rolls by -shift).  Tokens of one window attend to each
    other; under shift, pairs whose *rolled* positions lie in different wrap regions get -100
    added to the score (regions: [0,h-wh), [h-wh,h-sh), [h-sh,h) per axis)."""
    b, n, c = q.shape
    wh, ww = h // splits, w // splits
    sh, sw = (wh // 2, ww // 2) if shifted else (0, 0)
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    ry, rx = (ys - sh) % h, (xs - sw) % w            # position after roll by -shift
    win = (ry // wh) * splits + (rx // ww)           # [h,w]

    def region(r, size, wsz, s):
        return (r >= size - wsz).long() + (r >= size - s).long()

    reg = region(ry, h, wh, sh) * 3 + region(rx, w, ww, sw) if shifted else torch.zeros_like(win)
    win, reg = win.reshape(-1), reg.reshape(-1)
    rolled_order = (ry * w + rx).reshape(-1)
    out = torch.empty_like(q)
    scale = 1.0 / math.sqrt(c)
    for wi in range(splits * splits):
        idx = torch.nonzero(win == wi).reshape(-1)
        idx = idx[torch.argsort(rolled_order[idx])]   # same summation order as the reference
        s = (q[:, idx] @ k[:, idx].transpose(1, 2)) * scale
        if shifted:
            r = reg[idx]
            s = s + torch.where(r[:, None] != r[None, :], -100.0, 0.0)[None]
        out[:, idx] = torch.softmax(s, -1) @ v[:, idx]
    return out


def transformer_layer(sd, prefix, source, target, h, w, splits, shifted, ffn):
    """models/gmflow/transformer.py:147-185."""
    q = source @ sd[prefix + "q_proj.weight"].t()
    k = target @ sd[prefix + "k_proj.weight"].t()
    v = target @ sd[prefix + "v_proj.weight"].t()
    if splits > 1:
        m = window_attention(q, k, v, h, w, splits, shifted)
    else:
        m = torch.softmax((q @ k.transpose(1, 2)) / math.sqrt(q.shape[-1]), -1) @ v
    m = m @ sd[prefix + "merge.weight"].t()
    m = F.layer_norm(m, (m.shape[-1],), sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"])
    if ffn:
        m = torch.cat([source, m], -1) @ sd[prefix + "mlp.0.weight"].t()
        m = F.gelu(m) @ sd[prefix + "mlp.2.weight"].t()
        m = F.layer_norm(m, (m.shape[-1],), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"])
    return source + m


def feature_transformer(cfg, sd, f0, f1):
    """models/gmflow/transformer.py:279-339.  f0,f1 [P,C,h,w] -> same."""
    p_n, c, h, w = f0.shape
    t0 = f0.flatten(2).transpose(1, 2)
    t1 = f1.flatten(2).transpose(1, 2)
    src = torch.cat([t0, t1], 0)
    tgt = torch.cat([t1, t0], 0)
    for i in range(cfg.num_transformer_layers):
        pre = f"feat_enc.transformer.layers.{i}."
        shifted = (i % 2 == 1) and cfg.attn_splits > 1
        if not cfg.wo_self_attn:
            src = transformer_layer(sd, pre + "self_attn.", src, src, h, w, cfg.attn_splits, shifted, ffn=False)
        src = transformer_layer(sd, pre + "cross_attn_ffn.", src, tgt, h, w, cfg.attn_splits, shifted, ffn=True)
        tgt = torch.cat([src[p_n:], src[:p_n]], 0)
    o0 = src[:p_n].transpose(1, 2).reshape(p_n, c, h, w)
    o1 = src[p_n:].transpose(1, 2).reshape(p_n, c, h, w)
    return o0, o1


def upsampler(cfg, sd, x):
    """models/gmflow/superres.py:27-38."""
    p = "feat_enc.featup_net."
    right = F.conv2d(x, sd[p + "conv_l2rs.0.weight"], sd[p + "conv_l2rs.0.bias"], padding=1)
    left = x
    for i in range(int(math.log2(cfg.upsample_factor))):
        left = F.interpolate(left, scale_factor=2.0, mode="nearest")
        left = F.leaky_relu(F.conv2d(left, sd[f"{p}conv_ls.{i}.weight"], sd[f"{p}conv_ls.{i}.bias"], padding=1), 0.2)
        mid = F.conv2d(left, sd[f"{p}conv_l2rs.{i + 1}.weight"], sd[f"{p}conv_l2rs.{i + 1}.bias"], padding=1)
        right = F.interpolate(right, scale_factor=2, mode="bilinear", align_corners=False) + mid
    return right


_IMAGENET_MEAN = torch.tensor([0.485, 0.456, 0.406]).reshape(1, 3, 1, 1)
_IMAGENET_STD = torch.tensor([0.229, 0.224, 0.225]).reshape(1, 3, 1, 1)


def encode_pairs(cfg, sd, images):
    """models/gmflow/gmflow.py:47-67, 82-150 for one batch element.  images [V,3,H,W] in [0,1]
    -> list over 2 scales of (f0 [P,C,h,w], f1 [P,C,h,w]) (raw 1/8 and up-sampled 1/4)."""
    v_n, _, hh, ww = images.shape
    x = images
    if hh == 756 and ww == 1008:  # gmflow.py:100-103
        x = F.interpolate(x, size=(768, 1024), mode="bilinear", align_corners=True)
    feat = backbone(sd, (x - _IMAGENET_MEAN) / _IMAGENET_STD)
    pairs = pair_list(v_n)
    f0 = torch.stack([feat[a] for a, _ in pairs], 0)
    f1 = torch.stack([feat[b] for _, b in pairs], 0)
    if cfg.attn_splits > 1:
        f0, f1 = add_window_position(f0, cfg.attn_splits), add_window_position(f1, cfg.attn_splits)
    else:
        pe = sine_position(f0.shape[2], f0.shape[3], f0.shape[1])[None]
        f0, f1 = f0 + pe, f1 + pe
    f0, f1 = feature_transformer(cfg, sd, f0, f1)
    up = upsampler(cfg, sd, torch.cat([f0, f1], 0))
    return [(f0, f1), (up[:len(pairs)], up[len(pairs):])]


def pair_feats_to_view_chunks(pair_feats, n_views):
    """Re-express pair-major features in the reference's per-view layout [V,(V-1)*C,h,w]
    (models/matchnerf.py:192-205) — used only to compare against reference goldens."""
    pairs = pair_list(n_views)
    out = []
    for f0, f1 in pair_feats:
        per_view = [[] for _ in range(n_views)]
        for p, (a, b) in enumerate(pairs):
            per_view[a].append(f0[p])
            per_view[b].append(f1[p])
        out.append(torch.stack([torch.cat(ch, 0) for ch in per_view], 0))
    return out


# =============================================================================== a2 forward


def forward_test(cfg, sd, batch, chunk=4096, setbg_opaque=False, ray_idx=None):
    """models/matchnerf.py:32-73 in mode='test' (full image, legacy/regular sampling) for a
    batch dict of torch tensors (images [B,V+1,3,H,W], extrinsics [B,V+1,4,4], intrinsics,
    near_fars).  Returns dict(rgb [B,N,3], depth [B,N,1], opacity [B,N,1])."""
    b_n = batch["images"].shape[0]
    v = cfg.n_src_views
    outs = dict(rgb=[], depth=[], opacity=[])
    for b in range(b_n):
        imgs = batch["images"][b, :v]
        height, width = imgs.shape[-2:]
        feats = encode_pairs(cfg, sd, imgs)
        te, ti, tn = batch["extrinsics"][b, -1, :3], batch["intrinsics"][b, -1], batch["near_fars"][b, -1]
        se, si, sn = batch["extrinsics"][b, :-1, :3], batch["intrinsics"][b, :-1], batch["near_fars"][b, :-1]
        idx_all = torch.arange(height * width) if ray_idx is None else ray_idx
        parts = [render_rays(cfg, sd, idx_all[c:c + chunk], te, ti, tn, se, si, sn, imgs, feats, setbg_opaque)
                 for c in range(0, idx_all.numel(), chunk)]
        for k, i in (("rgb", 0), ("depth", 1), ("opacity", 2)):
            outs[k].append(torch.cat([p[i] for p in parts], 0))
    return {k: torch.stack(val, 0) for k, val in outs.items()}


def psnr(pred, gt, mask=None):
    """misc/metrics.py:35-41: -10*log10(mean((pred-gt)^2)) over kept pixels."""
    err = (pred - gt) ** 2
    if mask is not None:
        err = err[mask]
    return float(-10.0 * torch.log10(err.mean()))
