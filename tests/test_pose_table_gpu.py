"""Pose table (include/mnerf.h: mnerf_rays.pose_table): the video loop of the reference
(models/matchnerf.py:42-71, one full render per interpolated pose) with SEVERAL poses of a small frame in one launch.

The bar is bit-identity with the pose-by-pose launches: a table only changes WHERE a tile finds its camera constants (23 floats
from HBM instead of the kernel arguments), never the arithmetic.  The pose-by-pose path itself is pinned against the oracle in
tests/test_model_gpu.py::test_video_mode_renders_each_pose, which now runs through the table as well."""
import pytest
import torch

import matchnerf_amd.matchnerf as mn
from helpers import golden_case
from matchnerf_amd import camera, hip, synthetic as syn
from matchnerf_amd.edict import EasyDict
from test_model_gpu import build_model, to_batch

pytestmark = pytest.mark.gpu


def video(model, batch, batching, n_frames):
    """(cv_mm = 0: a pose table travels with the segment walk - the matrix form of the cost volume takes one pose per launch -
    so the pose-by-pose side is pinned to the walk as well: the claim under test is about the table, not about the kernel)"""
    model.opts.nerf.video_n_frames = n_frames
    model.pose_batching = batching
    with torch.no_grad(), hip.knob("cv_mm", 0):
        out = model(EasyDict(dict(batch)), mode="test", render_video=True, render_path_mode="interpolate")
    return {k: out[k].clone() for k in ("rgb", "depth", "opacity")}


def same_bits(a, b):
    return all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("name", ["c1_default", "nonlegacy", "inverse_depth"])
def test_video_through_the_pose_table_is_bit_identical_to_pose_by_pose(name):
    g, *_ = golden_case(name)
    opt, model = build_model(g["meta"])
    batch = to_batch(g)
    one = video(model, batch, False, 9)
    calls = []
    orig = hip.render_chunk
    try:
        hip.render_chunk = lambda sc, dec, rays, *a, **k: (calls.append((rays.n_rays, rays.rays_per_pose)), orig(sc, dec, rays, *a, **k))[1]
        tab = video(model, batch, True, 9)
    finally:
        hip.render_chunk = orig
    h, w = g["images"].shape[-2:]
    assert calls == [(9 * h * w, h * w)]  # ONE launch pair for the nine frames
    assert same_bits(one, tab)
    assert len({float(tab["rgb"][i].sum()) for i in range(9)}) == 9


def test_ragged_last_group_and_nonzero_first_pose(monkeypatch):
    """groups of 4 poses for 9 frames: 4 + 4 + 1, ray_begin = 4 and 8 frames into the table"""
    g, *_ = golden_case("c1_default")
    opt, model = build_model(g["meta"])
    batch = to_batch(g)
    h, w = g["images"].shape[-2:]
    one = video(model, batch, False, 9)
    monkeypatch.setattr(mn, "MAX_RAYS_PER_POSE_LAUNCH", 4 * h * w + 100)
    tab = video(model, batch, True, 9)
    assert same_bits(one, tab)


def test_batch_of_two_and_32_samples():
    """B = 2: a table per batch element, outputs scattered into the frame-major result; S = 32: the other table instance"""
    g, *_ = golden_case("c1_default")
    meta = dict(g["meta"])
    meta["opt_overrides"] = dict(meta["opt_overrides"], **{"nerf.sample_intvs": 32})
    opt, model = build_model(meta)
    sc = syn.make_scene(32, 48, 3, seed=5, batch_size=2)
    batch = EasyDict({k: torch.from_numpy(v).cuda() for k, v in sc.items()})
    one = video(model, batch, False, 6)
    tab = video(model, batch, True, 6)
    assert one["rgb"].shape == (12, 32 * 48, 3)
    assert same_bits(one, tab)


def test_render_poses_declines_where_no_table_instance_exists():
    """10 source views (pair-blocked cost volume, 4-step FiLM stage): no table; the video still renders, pose by pose"""
    from test_fullsize_gpu import build, gpu_batch
    opt, model, _ = build(n_views=10, S=32)
    scene = syn.make_scene(32, 48, 10, seed=3)
    batch = gpu_batch(scene)
    with torch.no_grad():
        ref_images = batch.images[:, :10]
        feats = model.get_img_feat(ref_images, cur_n_src_views=10)
        tgt, ref_poses = model.extract_poses(batch)
        poses = model.get_video_rendering_path(tgt, ref_poses, "interpolate", 6, batch)
        assert model.render_poses(opt, poses, ref_poses=ref_poses, ref_images=ref_images, ref_feats_list=feats) is None
    out = video(model, batch, True, 6)
    assert out["rgb"].shape == (6, 32 * 48, 3) and torch.isfinite(out["rgb"]).all()


def test_c_abi_refuses_what_a_table_cannot_do():
    g, *_ = golden_case("c1_default")
    opt, model = build_model(g["meta"])
    batch = to_batch(g)
    h, w = g["images"].shape[-2:]
    S = int(opt.nerf.sample_intvs)
    with torch.no_grad():
        ref_images = batch.images[:, :3]
        feats = model.get_img_feat(ref_images, cur_n_src_views=3)
        tgt, ref_poses = model.extract_poses(batch)
        ref_host, images_cl = model._frame_ctx(ref_poses, ref_images)
        sc = model._scene(0, ref_host, feats, images_cl)
        dec = model._decoder(S, ref_images.device)
    assert hip.render_takes_pose_table(sc, dec, S, h * w)
    assert not hip.render_takes_pose_table(sc, dec, S, h * w + 1)  # frames must be whole wavefronts
    assert hip.render_takes_pose_table(sc, dec, 128, h * w)        # round 5: the S = 128 instance (configs/demo_own.yaml)
    assert not hip.render_takes_pose_table(sc, dec, 129, h * w)    # S <= 128
    ex, it, nf = model._tgt_host(tgt)
    kinv, c2w = camera.target_ray_consts(ex[0], it[0], True)
    table = torch.from_numpy(hip.pose_table_rows([(kinv, c2w, nf[0, 0], nf[0, 1])] * 2)).cuda()
    n = 2 * h * w
    ws = torch.empty(hip.render_workspace_bytes(n, S, dec.cond_stride) // 4, device="cuda")
    outs = [torch.empty(n, c, device="cuda") for c in (3, 1, 1)]
    idx = torch.arange(n, dtype=torch.int32, device="cuda")

    def rays(**kw):
        args = dict(pose_table_ptr=table.data_ptr(), rays_per_pose=h * w)
        args.update(kw)
        return hip.make_rays(n, S, h, w, kinv, c2w, nf[0, 0], nf[0, 1], **args)

    hip.render_chunk(sc, dec, rays(), ws, *outs)  # the accepted form, as a control
    torch.cuda.synchronize()
    assert torch.equal(outs[0][:h * w], outs[0][h * w:])  # the same pose twice
    for bad, rc in ((rays(rays_per_pose=h * w - 4), hip.MNERF_E_RANGE), (rays(ray_idx_ptr=idx.data_ptr()), hip.MNERF_E_UNSUPPORTED)):
        with pytest.raises(hip.MnerfError) as e:
            hip.render_chunk(sc, dec, bad, ws, *outs)
        assert f"rc={rc}" in str(e.value) and "pose table" in str(e.value)
    with pytest.raises(hip.MnerfError):
        hip.render_chunk(sc, dec, rays(), None, *outs, fused=True)  # one-launch form: no table
    r = rays()
    r.pose_table = None  # rays_per_pose without a table
    with pytest.raises(hip.MnerfError):
        hip.render_chunk(sc, dec, r, ws, *outs)
    with pytest.raises(hip.MnerfError):
        hip.ray_samples(rays(), sc.views[0], device="cuda")


def test_small_frame_video_at_the_size_the_table_is_for():
    """128 x 160 frames (3 poses per launch), 9 poses (the interpolated path has 3 legs: a multiple of 3): bit-identical to
    pose by pose"""
    g, *_ = golden_case("c1_default")
    opt, model = build_model(g["meta"])
    sc = syn.make_scene(128, 160, 3, seed=9)
    batch = EasyDict({k: torch.from_numpy(v).cuda() for k, v in sc.items()})
    one = video(model, batch, False, 9)
    tab = video(model, batch, True, 9)
    assert same_bits(one, tab)
    assert torch.isfinite(tab["rgb"]).all() and float(tab["opacity"].min()) >= 0.0


def test_demo_own_video_goes_through_the_pose_table():
    """configs/demo_own.yaml on the reference's own scene: 24 frames of 256 x 160 at 128 samples per ray (40 960 rays per frame)
    leave as 4 launches of 6 poses (decoder_pp_kernel<128, 2, true>), bit-identical to pose by pose; frame 5 is the reference
    model's frame within the 1e-4 gate (tests/golden/demo_own.npz)."""
    from helpers import linf
    g, *_ = golden_case("demo_own")
    opt, model = build_model(g["meta"])
    batch = to_batch(g)
    one = video(model, batch, False, 24)
    calls = []
    orig = hip.render_chunk
    try:
        hip.render_chunk = lambda sc, dec, rays, *a, **k: (calls.append((rays.n_rays, rays.rays_per_pose)), orig(sc, dec, rays, *a, **k))[1]
        tab = video(model, batch, True, 24)
    finally:
        hip.render_chunk = orig
    assert calls == [(6 * 40960, 40960)] * 4
    assert same_bits(one, tab)
    f = int(g["video_frames"][0])
    assert linf(tab["rgb"][f], g["video_rgb"][0]) < 1e-4 and linf(tab["opacity"][f], g["video_opacity"][0]) < 1e-4
