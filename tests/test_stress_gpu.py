"""Stress tests of the two ray-chunk forms under co-residency (DESIGN.md section 4 "packed fp32
next to 16-bit MFMA").

Round 2 saw run-to-run different wrong rays when two workgroups of the one-launch (fused) form shared a CU.  The cause
(tools/exp/race_probe.py): packed-fp32 vector instructions of the cost-volume walk lose their result in lanes 48-63 of a
wave while another wave of the same SIMD issues 16-bit 32x32x16 matrix instructions.  The fused form is now built
without packed-fp32 instructions in the walk and runs two workgroups per CU again; these tests hammer exactly that
configuration and compare bit for bit."""
import pytest
import torch

from helpers import linf
from gpu_helpers import make_decoder_struct, make_rays_struct, make_scene_struct
from test_hip_kernels import _case_on_gpu  # noqa: F401  (golden case -> device tensors)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from matchnerf_amd import hip as H
    H.load()
    return H


@pytest.mark.parametrize("name", ["c1_default", "v4"])
def test_fused_two_workgroups_per_cu_60_frames_equal_staged_and_goldens(hip, name):
    """60 frames of a golden case through the one-launch form (two workgroups per CU: the kernel asks for its natural
    68 KiB of LDS): every frame is bit-identical to the staged form and within the golden tolerances."""
    g, cfg, sd, batch, feats_gpu, img_gpu = _case_on_gpu(name)
    sc = make_scene_struct(cfg, batch, feats_gpu, img_gpu)
    dec, keep = make_decoder_struct(cfg, sd, setbg_opaque=g["meta"]["setbg_opaque"], math="f16x3")
    h, w = batch["images"].shape[-2:]
    n = h * w
    rays = make_rays_struct(cfg, batch, n)
    assert hip.render_is_fused(sc, dec, rays)
    cond = hip.cost_volume(sc, rays, dec.cond_stride)
    with hip.knob("decoder_pp", 0):  # decoder_kernel: the kernel the one-launch form is built from, same summation order
        staged = hip.decoder_chunk(dec, sc.views[0], rays, cond)
    assert linf(staged[0], g["rgb"][0]) < 1e-4
    assert linf(staged[2], g["opacity"][0, :, 0]) < 1e-4 and linf(staged[1], g["depth"][0, :, 0]) < 3e-4
    default = hip.decoder_chunk(dec, sc.views[0], rays, cond)  # the ping-pong kernel (layer 5 summed activation half first)
    assert linf(default[0], staged[0]) < 3e-6 and linf(default[2], staged[2]) < 3e-6
    bad = 0
    for frame in range(60):
        out = [torch.full((n, 3), -1.0, device="cuda"), torch.full((n,), -1.0, device="cuda"), torch.full((n,), -1.0, device="cuda")]
        hip.render_chunk(sc, dec, rays, None, *out, fused=True)
        bad += int(((out[0] != staged[0]).any(1) | (out[1] != staged[1]) | (out[2] != staged[2])).sum())
    assert bad == 0, f"{bad} rays of the fused form differ from the staged form over 60 frames"


def test_fused_bench_frame_is_bit_identical_to_staged_over_50_launches():
    """BASELINE config[1] (512x640, 3 views, S=64): 10 frames x 5 launches of 65,536 rays through the one-launch form with
    two workgroups per CU against the staged form of the same launch, bit for bit (the configuration in which round 2
    saw up to 14 wrong rays per frame)."""
    import bench
    from matchnerf_amd import camera, hip
    dev = torch.device("cuda:0")
    opt, model, _ = bench.build_model(dev)
    _, batch = bench.make_batch(dev, 0)
    with torch.no_grad():
        feats = model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
    tgt_pose, ref_poses = model.extract_poses(batch)
    ref_host, images_cl = model._frame_ctx(ref_poses, batch.images[:, :3])
    tgt_ex, tgt_in, tgt_nf = model._tgt_host(tgt_pose)
    sc = model._scene(0, ref_host, feats, images_cl)
    dec = model._decoder(bench.S, dev)
    kinv, c2w = camera.target_ray_consts(tgt_ex[0], tgt_in[0], True)
    n_rays, chunk = bench.H * bench.W, 65536

    def rays_of(c):
        m = min(chunk, n_rays - c)
        return hip.make_rays(m, bench.S, bench.H, bench.W, kinv, c2w, tgt_nf[0, 0], tgt_nf[0, 1], ray_begin=c, legacy=True), m

    refs, wrong_pp = {}, 0
    for c in range(0, n_rays, chunk):
        rays, m = rays_of(c)
        cond = hip.cost_volume(sc, rays, dec.cond_stride)
        with hip.knob("decoder_pp", 0):
            refs[c] = hip.decoder_chunk(dec, sc.views[0], rays, cond)
        pp = hip.decoder_chunk(dec, sc.views[0], rays, cond)  # the default staged decoder: other summation order in layer 5
        wrong_pp += int(((pp[0] - refs[c][0]).abs().amax(1) > 3e-6).sum())
    assert wrong_pp == 0
    wrong = 0
    for frame in range(10):
        for c in range(0, n_rays, chunk):
            rays, m = rays_of(c)
            out = [torch.full((m, 3), -1.0, device=dev), torch.full((m,), -1.0, device=dev), torch.full((m,), -1.0, device=dev)]
            hip.render_chunk(sc, dec, rays, None, *out, fused=True)
            ref = refs[c]
            wrong += int(((out[0] != ref[0]).any(1) | (out[1] != ref[1]) | (out[2] != ref[2])).sum())
    assert wrong == 0, f"{wrong} wrong rays in 50 fused launches"


@pytest.mark.parametrize("pp", [1, 0])
def test_staged_decoder_is_reproducible_over_40_launches(hip, pp):
    """Both staged decoders (the ping-pong kernel: two teams of one 8-wave workgroup per CU; decoder_kernel: two workgroups per
    CU, same weight pipeline): 40 launches of one golden frame give the same bits every time and stay within the golden
    tolerances."""
    g, cfg, sd, batch, feats_gpu, img_gpu = _case_on_gpu("c1_default")
    sc = make_scene_struct(cfg, batch, feats_gpu, img_gpu)
    dec, keep = make_decoder_struct(cfg, sd, setbg_opaque=g["meta"]["setbg_opaque"], math="f16x3")
    h, w = batch["images"].shape[-2:]
    rays = make_rays_struct(cfg, batch, h * w)
    cond = hip.cost_volume(sc, rays, dec.cond_stride)
    with hip.knob("decoder_pp", pp):
        first = hip.decoder_chunk(dec, sc.views[0], rays, cond)
        assert linf(first[0], g["rgb"][0]) < 1e-4
        for _ in range(40):
            again = hip.decoder_chunk(dec, sc.views[0], rays, hip.cost_volume(sc, rays, dec.cond_stride))
            for a, b in zip(first, again):
                assert torch.equal(a, b)


@pytest.mark.parametrize("S", [200, 256])
def test_long_rays_full_launch_is_reproducible_and_matches_oracle(hip, S):
    """128 < S <= 256 ran decoder_kernel<8,256,*> until round 5: one 8-wave workgroup per CU, two waves per SIMD, VALU ray attention
    next to its own split-fp16 matrix phases.  Round 3 shipped it with packed-fp32 instructions behind ds_read_b128 (the erratum's
    pattern: loop-vectoriser output) and covered it with one 96-ray shot.  Here: one 65,536-ray launch (the golden scene's 4,096
    rays through an index list, 16 times over) x 20 - every launch bit-identical to the first and all 16 copies of a ray
    identical -, and a 256-ray slab against the CPU oracle."""
    from helpers import split_poses
    from oracle import matchnerf_oracle as O
    g, cfg, sd, batch, feats_gpu, img_gpu = _case_on_gpu("c1_default")
    cfg.sample_intvs = S
    v = cfg.n_src_views
    sc = make_scene_struct(cfg, batch, feats_gpu, img_gpu)
    dec, keep = make_decoder_struct(cfg, sd, math="f16x3")
    h, w = batch["images"].shape[-2:]
    n = 65536
    idx = (torch.arange(n, device="cuda") % (h * w)).int()
    rays = make_rays_struct(cfg, batch, n, ray_idx_gpu=idx)
    cond = hip.cost_volume(sc, rays, dec.cond_stride)
    first = [t.clone() for t in hip.decoder_chunk(dec, sc.views[0], rays, cond)]
    for t in first:  # the 16 copies of the frame agree with each other
        t16 = t.reshape(16, h * w, -1)
        assert torch.equal(t16, t16[:1].expand_as(t16))
    for _ in range(20):
        again = hip.decoder_chunk(dec, sc.views[0], rays, cond)
        for a, b in zip(first, again):
            assert torch.equal(a, b)
    pair_feats = [(f[:, 0].permute(0, 3, 1, 2).cpu(), f[:, 1].permute(0, 3, 1, 2).cpu()) for f in feats_gpu]
    with torch.no_grad():
        ref = O.render_rays(cfg, sd, torch.arange(1024, 1280), *split_poses(batch), batch["images"][0, :v], pair_feats)
    assert linf(first[0][1024:1280], ref[0]) < 1e-4 and linf(first[2][1024:1280], ref[2][:, 0]) < 1e-4
    assert linf(first[1][1024:1280], ref[1][:, 0]) < 3e-4
    # Since round 5 the launches above run decoder_pp_kernel<256> (one ray per tile across both teams, the ray attention in two
    # key halves merged flash style).  The round-2 kernel (decoder_kernel<8,256,*>, VALU ray attention) stays behind the knob:
    # same frame within the parity gate (other summation orders), and reproducible itself.
    with hip.knob("decoder_pp_max_s", 128):
        old = [t.clone() for t in hip.decoder_chunk(dec, sc.views[0], rays, cond)]
        again = hip.decoder_chunk(dec, sc.views[0], rays, cond)
    for a, b in zip(old, again):
        assert torch.equal(a, b)
    assert linf(old[0], first[0]) < 5e-5 and linf(old[2], first[2]) < 5e-5 and linf(old[1], first[1]) < 3e-4
    assert linf(old[0][1024:1280], ref[0]) < 1e-4
    # per-sample colours / densities of the two kernels (the outputs CondNeRF.forward hands back), a ragged launch of 1 001 rays
    rays_s = make_rays_struct(cfg, batch, 1001, ray_begin=333)
    cond_s = hip.cost_volume(sc, rays_s, dec.cond_stride)
    new_s = hip.decoder_chunk(dec, sc.views[0], rays_s, cond_s, want_samples=True)
    with hip.knob("decoder_pp_max_s", 128):
        old_s = hip.decoder_chunk(dec, sc.views[0], rays_s, cond_s, want_samples=True)
    assert linf(new_s[3], old_s[3]) < 5e-5 and linf(new_s[4], old_s[4]) < 5e-5 * max(1.0, float(old_s[4].abs().max()))
    assert linf(new_s[0], old_s[0]) < 5e-5


def test_other_kernels_next_to_the_f16_decoder_on_a_second_stream():
    """tools/exp/race_probe.py E7 as a test: the possible VICTIMS of the erratum - the encoder's kernels (split-fp16
    convolutions, InstanceNorm, q|k|v, window attention, K7) and mnerf_cost_volume_backward - run on one stream while the
    f16x3 decoder issues 16-bit 32x32x16 matrix instructions on another.  Feature maps: bit-identical to the maps computed
    alone, every round.  Cost-volume backward: its scatter-adds are float atomics (order differs from run to run even
    alone: measured up to 3.5e-6 of the largest gradient), so the gate is 2e-5 - a lost term of a sum shows up two orders above."""
    import bench
    from matchnerf_amd import camera, hip
    dev = torch.device("cuda:0")
    opt, model, _ = bench.build_model(dev)
    _, batch = bench.make_batch(dev, 0)
    with torch.no_grad():
        solo = [f.clone() for f in model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)]
    tgt_pose, ref_poses = model.extract_poses(batch)
    ref_host, images_cl = model._frame_ctx(ref_poses, batch.images[:, :3])
    tgt_ex, tgt_in, tgt_nf = model._tgt_host(tgt_pose)
    sc = model._scene(0, ref_host, solo, images_cl)
    dec = model._decoder(bench.S, dev)
    kinv, c2w = camera.target_ray_consts(tgt_ex[0], tgt_in[0], True)
    rays = hip.make_rays(65536, bench.S, bench.H, bench.W, kinv, c2w, tgt_nf[0, 0], tgt_nf[0, 1], ray_begin=0, legacy=True)
    cond = hip.cost_volume(sc, rays, dec.cond_stride)
    ref_out = [t.clone() for t in hip.decoder_chunk(dec, sc.views[0], rays, cond)]
    # victims' inputs for the backward: a 4,096-ray chunk and a fixed upstream gradient
    rays_b = hip.make_rays(4096, bench.S, bench.H, bench.W, kinv, c2w, tgt_nf[0, 0], tgt_nf[0, 1], ray_begin=100000, legacy=True)
    gen = torch.Generator(device="cpu").manual_seed(3)
    g_cond = torch.randn(4096 * bench.S, dec.cond_stride, generator=gen).to(dev)
    solo_grads = hip.cost_volume_backward(sc, rays_b, dec.cond_stride, g_cond, [torch.zeros_like(f) for f in solo])
    solo_grads = [t.clone() for t in solo_grads]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    feat_diffs, worst_grad, dec_diffs = 0, 0.0, 0
    for it in range(8):
        torch.cuda.synchronize()
        with torch.cuda.stream(s2):
            outs = [hip.decoder_chunk(dec, sc.views[0], rays, cond, stream=s2) for _ in range(3)]
        with torch.cuda.stream(s1), torch.no_grad():
            f = model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
            grads = hip.cost_volume_backward(sc, rays_b, dec.cond_stride, g_cond, [torch.zeros_like(x) for x in solo], stream=s1)
        torch.cuda.synchronize()
        for k in range(2):
            feat_diffs += int((f[k].view(torch.int32) != solo[k].view(torch.int32)).sum())
            worst_grad = max(worst_grad, float((grads[k] - solo_grads[k]).abs().max() / solo_grads[k].abs().max()))
        for o in outs:
            dec_diffs += sum(int((a != b).sum()) for a, b in zip(o, ref_out))
    assert feat_diffs == 0, f"{feat_diffs} feature values differ next to the decoder"
    assert dec_diffs == 0, f"{dec_diffs} decoder outputs differ next to the encoder"
    assert worst_grad < 2e-5, worst_grad
