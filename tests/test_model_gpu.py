"""End-to-end GPU parity of the drop-in module: MatchNeRF(opts).forward(batch, mode) on the HIP
path vs the reference goldens (encoder with the K6 kernel + render kernels), for the
BASELINE config[0] case and the option variants.  Gate: RGB L-inf <= 1e-4 (north_star)."""
import numpy as np
import pytest
import torch

from helpers import golden_case, linf, split_poses
from matchnerf_amd import options, synthetic as syn
from matchnerf_amd.edict import EasyDict
from oracle import matchnerf_oracle as O

pytestmark = pytest.mark.gpu

def build_model(meta, device="cuda"):
    from matchnerf_amd.models import models_dict
    opt = options.load_options(f"configs/{meta.get('yaml', 'test')}.yaml", verbose=False)
    opt.device = device
    for k, v in meta["opt_overrides"].items():
        node = opt
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    model = models_dict[opt.model](opt).to(device).eval()
    spec = syn.state_dict_spec(n_src_views=opt.n_src_views)
    model.load_state_dict(syn.to_torch(syn.seeded_state_dict(spec, meta["weight_seed"]), device))
    model.nerf_setbg_opaque = meta["setbg_opaque"]
    return opt, model


def to_batch(g, device="cuda"):
    return EasyDict({k: torch.from_numpy(g[k]).to(device) for k in ("images", "extrinsics", "intrinsics", "near_fars")})


@pytest.mark.parametrize("name", ["c1_default", "rect_wide", "nonlegacy", "v4", "inverse_depth", "demo_own_small", "demo_own"])
def test_forward_test_mode_matches_reference(name):
    g, cfg, sd, _ = golden_case(name)
    opt, model = build_model(g["meta"])
    batch = to_batch(g)
    with torch.no_grad():
        out = model(batch, mode="test")
    assert out.rgb.shape == g["rgb"].shape and out.depth.shape == g["depth"].shape
    assert linf(out.rgb, g["rgb"]) < 1e-4
    assert linf(out.opacity, g["opacity"]) < 1e-4
    assert linf(out.depth, g["depth"]) < 3e-4
    mse = float(((out.rgb.cpu() - torch.from_numpy(g["rgb"])) ** 2).mean())
    gt = batch.images[:, -1].permute(0, 2, 3, 1).reshape(1, -1, 3).cpu()
    d_psnr = abs(O.psnr(out.rgb.cpu(), gt) - O.psnr(torch.from_numpy(g["rgb"]), gt))
    assert mse < 1e-9 and d_psnr < 0.01  # north_star: PSNR delta < 0.01 dB


@pytest.mark.parametrize("name", ["c1_default", "rect_wide", "demo_own"])
def test_encoder_features_match_reference(name):
    from matchnerf_amd.gmflow import pair_major_to_view_chunks
    g, cfg, sd, _ = golden_case(name)
    opt, model = build_model(g["meta"])
    batch = to_batch(g)
    with torch.no_grad():
        feats = model.get_img_feat(batch.images[:, :cfg.n_src_views], cur_n_src_views=cfg.n_src_views)
    for i, f in enumerate(feats):
        ref_layout = pair_major_to_view_chunks(f)[0].cpu()
        got, want = (ref_layout, g[f"feat_scale{i}"]) if f"feat_scale{i}" in g else (ref_layout[:, ::16], g[f"feat_scale{i}_sub"])
        # 5e-4 absolute on features that reach ~30 (observed on MI355X: 1.4e-4 .. 2.5e-4 on the synthetic scenes).  The real
        # photographs at 256x160 are twice as sensitive for ANY fp32 evaluation (CPU oracle vs reference there: 7.6e-5 against
        # 2.3e-5 .. 5.9e-5) and measure 6.3e-4 / 5.7e-4 = 3.2e-5 of the largest feature (split-fp16 operands carry 22 bits, an fp32
        # product 24); that case is gated relative to the feature range.  The rendered frame of the same case holds the 1e-4 gate.
        tol = 4e-5 * float(np.abs(want).max()) if name == "demo_own" else 5e-4
        assert linf(got, want) < tol, (i, linf(got, want), tol)


@pytest.mark.parametrize("name", ["c1_default", "rect_wide", "nonlegacy"])
def test_cond_nerf_forward_matches_reference(name):
    """The L1b public interface: nerf_dec(opt, points_3D, ray_unit, cond_info) -> (rgb, density) with
    the reference's call signature and shapes (cond_nerf.py:52-100), fed with the reference's own decoder inputs;
    composite() then returns the reference's four values (nerf.py:124)."""
    g, cfg, sd, _ = golden_case(name)
    opt, model = build_model(g["meta"])
    n, s, dc = g["cond"].shape
    v = cfg.n_src_views
    sum_g = dc - 4 * v
    cond = torch.from_numpy(g["cond"]).cuda()[None]
    cond_info = dict(feat_info=cond[..., :sum_g], color_info=cond[..., sum_g:sum_g + 3 * v], mask_info=cond[..., sum_g + 3 * v:])
    x = torch.from_numpy(g["x_ref"]).cuda()[None]
    ray_unit = torch.from_numpy(g["dir_ref"]).cuda()[None, :, None, :].expand(1, n, s, 3)
    rgb, sigma = model.nerf_dec(opt, x, ray_unit=ray_unit, cond_info=cond_info)
    assert rgb.shape == (1, n, s, 3) and sigma.shape == (1, n, s)
    assert linf(rgb[0], g["rgb_samples"]) < 5e-5 and linf(sigma[0], g["sigma"]) < 5e-5
    # volume rendering through the reference-signature composite: matches the reference's rendered rays
    depth = torch.from_numpy(np.linalg.norm(g["pts"] - g["pts"][:, :1], axis=-1)).cuda()  # any increasing depths do for prob
    ray = torch.ones(1, n, 3, device="cuda")
    out = model.nerf_dec.composite(opt, ray, rgb, sigma, depth[None, ..., None], setbg_opaque=False)
    assert len(out) == 4 and out[3].shape == (1, n, s, 1)
    assert linf(out[3].sum(2), out[2]) < 2e-6
    if opt.nerf.wo_render_interval:
        sel = g["stage_rays"]
        assert linf(out[2][0], g["opacity"][0, sel]) < 1e-4


def test_random_ray_mode_matches_oracle():
    """mode='train' path (random pixel subset) evaluated without autograd, deterministic depths."""
    g, cfg, sd, batch_cpu = golden_case("c1_default")
    opt, model = build_model(g["meta"])
    opt.nerf.rand_rays_train = 777
    opt.nerf.sample_stratified = False
    batch = to_batch(g)
    with torch.no_grad():
        out = model(batch, mode="train")
    idx = out.ray_idx.cpu()
    assert idx.numel() == 777 and out.rgb.shape == (1, 777, 3)
    v = cfg.n_src_views
    with torch.no_grad():
        feats = O.encode_pairs(cfg, sd, batch_cpu["images"][0, :v])
        ref = O.render_rays(cfg, sd, idx, *split_poses(batch_cpu), batch_cpu["images"][0, :v], feats)
    assert linf(out.rgb[0], ref[0]) < 1e-4 and linf(out.opacity[0], ref[2]) < 1e-4


def test_train_mode_gradients_match_oracle_autograd(monkeypatch):
    """mode='train' under autograd: forward values from the HIP kernels; the ray chunk's backward is HIP end to end
    (compositing, conditional MLP + ray transformer on the forward's own sample coordinates, cost volume:
    matchnerf_amd/autograd.py) — compared with autograd through the CPU oracle on all nine probe parameters."""
    g, cfg, sd, batch_cpu = golden_case("c1_default")
    opt, model = build_model(g["meta"])
    model.train()
    # (until round 6 the convolutions' backward ran on MIOpen and its solver choice had to be pinned here - benchmark mode picked
    # solvers by timing, 7e-5 ... 3.7e-3 from box to box; the CNN's forward and backward are this library's kernels now:
    # csrc/conv_backward.hip, instance_norm.hip - nothing to pin)
    opt.nerf.rand_rays_train = 96
    opt.nerf.sample_stratified = False
    batch = to_batch(g)
    torch.manual_seed(0)
    out = model(batch, mode="train")
    idx = out.ray_idx
    gt = batch.images[:, -1].reshape(1, 3, -1).permute(0, 2, 1)[:, idx]
    loss = ((out.rgb - gt) ** 2).mean() + 0.1 * out.opacity.mean() + 0.05 * out.depth.mean()
    loss.backward()

    sd_req = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    v = cfg.n_src_views
    feats = O.encode_pairs(cfg, sd_req, batch_cpu["images"][0, :v])
    ref = O.render_rays(cfg, sd_req, idx.cpu(), *split_poses(batch_cpu), batch_cpu["images"][0, :v], feats)
    gt_c = batch_cpu["images"][0, -1].reshape(3, -1).t()[idx.cpu()]
    loss_ref = ((ref[0] - gt_c) ** 2).mean() + 0.1 * ref[2].mean() + 0.05 * ref[1].mean()
    loss_ref.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-5
    params = dict(model.named_parameters())
    checked, worst, worst_enc = 0, 0.0, 0.0
    for name in ("nerf_dec.pts_linears.0.weight", "nerf_dec.pts_bias.weight", "nerf_dec.rgb_linear.weight",
                 "nerf_dec.ray_attention.w_qs.weight", "nerf_dec.out_alpha_linear.2.weight",
                 "feat_enc.transformer.layers.5.cross_attn_ffn.mlp.2.weight",
                 "feat_enc.transformer.layers.1.self_attn.q_proj.weight", "feat_enc.backbone.conv1.weight",
                 "feat_enc.featup_net.conv_l2rs.1.weight", "feat_enc.backbone.layer1.0.conv2.weight",
                 "feat_enc.backbone.layer2.0.conv1.weight", "feat_enc.backbone.layer2.0.downsample.0.weight",
                 "feat_enc.backbone.layer3.1.conv1.weight",  # (a bias in front of an InstanceNorm has gradient zero: not probed)
                 "feat_enc.backbone.conv2.weight", "feat_enc.featup_net.conv_ls.0.bias"):
        a, b = params[name].grad.cpu(), sd_req[name].grad
        scale = float(b.abs().max()) + 1e-12
        rel = float((a - b).abs().max()) / scale
        print(f"grad {name}: max|ref| {scale:.3e} rel err {rel:.2e}")
        if name.startswith("nerf_dec."):
            worst = max(worst, rel)
        else:
            worst_enc = max(worst_enc, rel)
        checked += 1
    # Decoder parameters: HIP forward + HIP K5 / K1+K2 backward + the re-evaluated MLP on the forward's own (bit-exact)
    # sample coordinates: 1e-3 (observed <= 2e-4).  Encoder parameters: transformer layers' backward in HIP since round 4 (attention
    # backward, split-bf16 GEMMs), the backbone's and the up-sampler's convolutions and norms in HIP since round 6 (exact-f32 matrix
    # products): every probe - stem, stride-1 / stride-2 3x3, the 1x1 downsample, the last 1x1, the up-sampler and a bias of it - at 1e-3.
    assert checked == 15 and worst < 1e-3 and worst_enc < 1e-3, (worst, worst_enc)


def test_stratified_depths_match_oracle():
    """kernel-level: explicit U[0,1) offsets through mnerf_rays.strat_u vs the oracle."""
    from gpu_helpers import (images_rgba, make_decoder_struct, make_rays_struct, make_scene_struct,
                             ref_layout_to_pair_major)
    from matchnerf_amd import hip
    g, cfg, sd, batch = golden_case("c1_default")
    v = cfg.n_src_views
    feats_pm = [ref_layout_to_pair_major(torch.from_numpy(g[f"feat_scale{i}"]), v).cuda() for i in range(2)]
    img = images_rgba(batch["images"][0, :v]).cuda()
    sc = make_scene_struct(cfg, batch, feats_pm, img)
    dec, keep = make_decoder_struct(cfg, sd)
    n = 200
    u = torch.rand(n, cfg.sample_intvs, generator=torch.Generator().manual_seed(1))
    idx = torch.randperm(64 * 64, generator=torch.Generator().manual_seed(2))[:n]
    idx_gpu = idx.int().cuda()  # keep alive: the struct only holds the raw pointer
    rays = make_rays_struct(cfg, batch, n, ray_idx_gpu=idx_gpu)
    u_gpu = u.cuda()
    rays.strat_u = u_gpu.data_ptr()
    rgb, depth, opacity = (torch.empty(n, 3, device="cuda"), torch.empty(n, device="cuda"), torch.empty(n, device="cuda"))
    ws = torch.empty(hip.render_workspace_bytes(n, cfg.sample_intvs, dec.cond_stride) // 4, device="cuda")
    hip.render_chunk(sc, dec, rays, ws, rgb, depth, opacity)
    pair_feats = [(f[:, 0].permute(0, 3, 1, 2).cpu(), f[:, 1].permute(0, 3, 1, 2).cpu()) for f in feats_pm]
    with torch.no_grad():
        ref = O.render_rays(cfg, sd, idx, *split_poses(batch), batch["images"][0, :v], pair_feats, stratified_u=u)
    assert linf(rgb, ref[0]) < 1e-4 and linf(depth, ref[1][:, 0]) < 3e-4


def test_video_mode_renders_each_pose():
    """render_video: streamed frames against the CPU ORACLE rendering the same pose (two poses in full), plus shape /
    device / distinctness checks for all frames.  (Legacy-coordinate case: with render intervals the reference's
    1e10 last interval turns fp32 noise in the last density into 1e-4-class colour changes at some poses — in every
    matrix path alike — which makes that variant a poor oracle target away from its golden pose.)"""
    g, cfg, sd, batch_cpu = golden_case("c1_default")
    opt, model = build_model(g["meta"])
    opt.nerf.video_n_frames = 6
    batch = to_batch(g)
    with torch.no_grad():
        out = model(batch, mode="test", render_video=True, render_path_mode="interpolate")
    h, w = g["images"].shape[-2:]
    assert out.rgb.shape == (6, h * w, 3) and out.rgb.device.type == "cpu"
    assert out.depth.shape == (6, h * w, 1) and out.opacity.shape == (6, h * w, 1)
    assert torch.isfinite(out.rgb).all()
    tgt, ref_poses = model.extract_poses(batch)
    poses = model.get_video_rendering_path(tgt, ref_poses, "interpolate", 6, batch)
    assert all("_host" in p for p in poses)   # host originals ride along: no device->host copy per frame
    for i in (1, 4):
        b_i = {k: v.clone() for k, v in batch_cpu.items()}
        b_i["extrinsics"][:, -1, :3] = poses[i]["extrinsics"].cpu()
        with torch.no_grad():
            ref = O.forward_test(cfg, sd, b_i)
        assert linf(out.rgb[i], ref["rgb"][0]) < 1e-4 and linf(out.opacity[i], ref["opacity"][0]) < 1e-4
        assert linf(out.depth[i], ref["depth"][0]) < 3e-4
    assert len({float(out.rgb[i].sum()) for i in range(6)}) == 6  # six different poses


@pytest.mark.parametrize("name", ["demo_own_small", "demo_own"])
def test_video_frames_of_the_real_scene_match_reference(name):
    """configs/demo_own.yaml on the reference's own COLMAP scene: frames of the 24-pose 'interpolate' path rendered by
    forward(render_video=True) against the REFERENCE's frames at those poses (same 1e-4 gate as the still image)."""
    g, cfg, sd, _ = golden_case(name)
    opt, model = build_model(g["meta"])
    assert opt.nerf.video_n_frames == 24 and opt.nerf.sample_intvs == 128
    batch = to_batch(g)
    batch.c2ws_all = torch.from_numpy(g["c2ws_all"]).cuda()
    with torch.no_grad():
        out = model(batch, mode="test", render_video=True, render_path_mode="interpolate")
    h, w = g["images"].shape[-2:]
    assert out.rgb.shape == (24, h * w, 3)
    for j, f in enumerate(g["video_frames"]):
        assert linf(out.rgb[f], g["video_rgb"][j]) < 1e-4, (name, f)
        assert linf(out.opacity[f], g["video_opacity"][j]) < 1e-4
        assert linf(out.depth[f], g["video_depth"][j]) < 3e-4


def test_batch_of_two_equals_two_batches_of_one():
    g, *_ = golden_case("nonlegacy")
    opt, model = build_model(g["meta"])
    sc = syn.make_scene(32, 48, 3, seed=11, batch_size=2)
    full = EasyDict({k: torch.from_numpy(v).cuda() for k, v in sc.items()})
    with torch.no_grad():
        both = model(full, mode="test")
        for b in range(2):
            one = model(EasyDict({k: torch.from_numpy(v[b:b + 1]).cuda() for k, v in sc.items()}), mode="test")
            # MIOpen may pick a different conv algorithm for 6 vs 3 backbone images: ulp-level
            # feature differences, far below the parity gate
            assert linf(both.rgb[b], one.rgb[0]) < 2e-5


def test_coach_test_model_reports_psnr(tmp_path, monkeypatch):
    """thin harness (reference: test.py + Coach.test_model): options -> model -> PSNR report"""
    from matchnerf_amd.coach import Coach
    monkeypatch.chdir(tmp_path)
    cmd = options.parse_arguments(["--yaml=test", "--name=harness", "--nerf.sample_intvs=32",
                                   "--data_test.llff=", "--data_test.blender=", "--data_test.tnt=",
                                   "--data_test.dtu.img_wh=48,32", "--data_test.dtu.max_len=1"])
    opt = options.set(cmd, verbose=False)
    c = Coach(opt)
    c.build_networks()
    c.restore_checkpoint()
    c.load_dataset()
    rep = c.test_model()
    assert list(rep) == ["dtu"] and len(rep["dtu"]) == 1
    assert all(np.isfinite(v) and 0 < v < 60 for v in rep["dtu"].values())


def test_demo_own_yaml_renders_the_reference_scene_from_disk(tmp_path, monkeypatch):
    """`python test.py --yaml=demo_own --data_test.colmap.root_dir=<dir>`: the reference's demo (configs/demo_own.yaml, its own
    three photographs + COLMAP poses) through options -> Coach -> the COLMAP producer -> forward(render_video=True).  The frame
    the reference-model golden holds (frame 5 of the 24-pose path, tests/golden/demo_own.npz) must come out within the gate."""
    import os
    import test as entry
    from conftest import GOLDEN
    monkeypatch.chdir(tmp_path)
    g, *_ = golden_case("demo_own")
    videos = entry.run(["--yaml=demo_own", f"--data_test.colmap.root_dir={os.path.join(GOLDEN, 'demo_data')}",
                        "--data_test.colmap.num_workers=0", "--data_test.tnt=", f"--output_root={tmp_path}", "--load="])
    frames = videos["colmap"]
    assert frames.shape == (24, 160, 256, 3) and frames.dtype == np.uint8
    want = (torch.from_numpy(g["video_rgb"][0]).reshape(160, 256, 3).clamp(0, 1) * 255).to(torch.uint8).numpy()
    assert np.abs(frames[int(g["video_frames"][0])].astype(int) - want.astype(int)).max() <= 1
    # what the reference's test_model_video leaves on disk (coach.py:507-527), under its names: the GIF (nerf.save_gif) and
    # the strip of source views; no per-frame files (nerf.save_frames is off in this yaml)
    from PIL import Image
    out_dir = tmp_path / "test_video" / "demo" / "test_videos" / "colmap"
    with Image.open(out_dir / "printer_view00_src02_01_00.gif") as im:
        assert im.n_frames == 24 and im.size == (256, 160)
    with Image.open(out_dir / "printer_view00_src02_01_00.jpg") as im:
        assert im.size == (3 * 256, 160)
    assert not list(out_dir.glob("*_f0.jpg"))


def test_video_own_yaml_at_256_samples_runs_the_wide_ping_pong_decoder(tmp_path, monkeypatch):
    """configs/test_video_own.yaml (sample_intvs 256, attn_splits 4, IBRNet switches, the reference's scene at 960 x 640) through
    test.py with three poses: decoder_pp_kernel<256> (round 5) against the round-2 kernel it replaces (behind the knob) on real
    data with density_maskfill / raytrans_posenc / ELU on: the same frames within the parity gate."""
    import os
    import test as entry
    from conftest import GOLDEN
    from matchnerf_amd import hip
    monkeypatch.chdir(tmp_path)
    argv = ["--yaml=test_video_own", f"--data_test.colmap.root_dir={os.path.join(GOLDEN, 'demo_data')}", "--data_test.colmap.num_workers=0",
            "--data_test.tnt=", f"--output_root={tmp_path}", "--load=", "--nerf.video_n_frames=3"]
    new = entry.run(argv)["colmap"]
    assert new.shape == (3, 640, 960, 3) and new.dtype == np.uint8
    with hip.knob("decoder_pp_max_s", 128):
        old = entry.run(argv)["colmap"]
    assert np.abs(new.astype(int) - old.astype(int)).max() <= 1
    assert len({int(f.astype(np.int64).sum()) for f in new}) == 3


def test_two_same_shaped_batches_do_not_share_launch_context():
    """the per-source-set launch context (host cameras + RGBA source images) must never survive from
    one batch to the next.  Batch A is rendered and FREED, batch B of the same shape is then allocated (the caching
    allocator hands out the same addresses, version 0, same shapes: the old cache key matched) and rendered by the same
    model; the result must equal a fresh model's."""
    g, cfg, sd, _ = golden_case("c1_default")
    opt, model = build_model(g["meta"])
    batch_a = to_batch(g)
    with torch.no_grad():
        model(batch_a, mode="test")
    ptr_a = batch_a.images.data_ptr()
    del batch_a
    gb = {k: np.array(g[k]) for k in ("images", "extrinsics", "intrinsics", "near_fars")}
    gb["images"] = np.ascontiguousarray(gb["images"][..., ::-1, :]) * 0.5 + 0.25   # other colours
    gb["extrinsics"][0, :, 0, 3] += 0.05                                            # other cameras
    batch_b = to_batch(gb)
    same_address = batch_b.images.data_ptr() == ptr_a   # the situation the advisor describes (not guaranteed)
    with torch.no_grad():
        out_b = model(batch_b, mode="test")
        _, fresh = build_model(g["meta"])
        ref_b = fresh(to_batch(gb), mode="test")
    assert torch.equal(out_b.rgb, ref_b.rgb) and torch.equal(out_b.depth, ref_b.depth), same_address
    # and the context is dropped at the top of every forward
    model._frame = ("stale",)
    with torch.no_grad():
        again = model(batch_b, mode="test")
    assert torch.equal(again.rgb, ref_b.rgb)


def test_encoder_graph_replay_equals_the_eager_pass_and_follows_weight_updates():
    """get_img_feat replays a captured HIP graph of the encoder pass at inference (matchnerf.py: _encoder_graph_replay): same
    kernels, same bits as the eager pass; the outputs are clones (a second call does not overwrite the first result); a
    parameter update changes the key, so the next call re-captures and again equals the eager pass."""
    g, cfg, sd, _ = golden_case("c1_default")
    opt, model = build_model(g["meta"])
    batch = to_batch(g)
    imgs = batch.images[:, :3]
    with torch.no_grad():
        model.encoder_graph = False
        eager = [f.clone() for f in model.get_img_feat(imgs, cur_n_src_views=3)]
        model.encoder_graph = True
        first = model.get_img_feat(imgs, cur_n_src_views=3)      # warm-up + capture + replay
        assert model.encoder_graph and len(model._enc_graphs) == 1, "capture failed"
        second = model.get_img_feat(imgs.flip(-1).contiguous(), cur_n_src_views=3)  # another input through the same graph
        third = model.get_img_feat(imgs, cur_n_src_views=3)
        for a, b, c in zip(eager, first, third):
            assert torch.equal(a, b) and torch.equal(a, c)
        assert not torch.equal(first[0], second[0])
        # a weight update: new key, new capture, still the eager pass's bits
        model.feat_enc.transformer.layers[0].self_attn.q_proj.weight.mul_(1.01)
        upd = model.get_img_feat(imgs, cur_n_src_views=3)
        assert len(model._enc_graphs) == 1 and not torch.equal(upd[0], eager[0])  # the old weights' graph is dropped, not kept
        model.encoder_graph = False
        eager2 = model.get_img_feat(imgs, cur_n_src_views=3)
        for a, b in zip(upd, eager2):
            assert torch.equal(a, b)
