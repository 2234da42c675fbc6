"""Round-2 GPU diagnostics (one-off): determinism / chunk invariance of both ray-chunk forms on the bench frame,
per-math-path error of the video test case vs the oracle, and a two-stream overlap experiment
(cost volume of chunk k+1 under the decoder of chunk k)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from matchnerf_amd import camera, hip, matchnerf as M  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "all"


def diff(a, b, tag):
    d = (a - b).abs()
    bad = (d.reshape(d.shape[0], -1).max(1).values > 0).nonzero().flatten()
    print(f"  {tag}: equal={bool(torch.equal(a, b))} max|d|={float(d.max()):.3e} differing rows={bad.numel()}"
          + (f" first={bad[:8].tolist()} last={bad[-4:].tolist()}" if bad.numel() else ""), flush=True)


if which in ("all", "det"):
    opt, model, _ = bench.build_model(dev)
    _, batch = bench.make_batch(dev, 0)
    with torch.no_grad():
        feats = model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
        model.get_img_feat = lambda *a, **k: feats

        def frame(fused, chunk=65536):
            M.MAX_RAYS_PER_LAUNCH = chunk
            model.fused_render = fused
            o = model(batch, mode="test")
            torch.cuda.synchronize()
            return torch.cat([o.rgb[0], o.depth[0], o.opacity[0]], -1).clone()
        s1, s2 = frame(False), frame(False)
        diff(s1, s2, "staged vs staged (same chunking)")
        diff(s1, frame(False, 4096 * 7 + 13), "staged vs staged chunk 28685")
        f1, f2 = frame(True), frame(True)
        diff(f1, f2, "fused vs fused (same chunking)")
        diff(f1, s1, "fused vs staged")
        diff(frame(True, 4096 * 7 + 13), s1, "fused chunk 28685 vs staged")
        diff(frame(True, 512), s1, "fused chunk 512 (one tile per workgroup) vs staged")
        model.fused_render = False
        M.MAX_RAYS_PER_LAUNCH = 65536

if which in ("all", "video"):
    from helpers import golden_case
    from test_model_gpu import build_model, to_batch
    from oracle import matchnerf_oracle as O
    g, cfg, sd, batch_cpu = golden_case("nonlegacy")
    for math in ("f16x3", "bf16x6", "f32"):
        os.environ["MNERF_DECODER_MATH"] = math
        opt, model = build_model(g["meta"])
        opt.nerf.video_n_frames = 6
        batch = to_batch(g)
        with torch.no_grad():
            out = model(batch, mode="test", render_video=True, render_path_mode="interpolate")
            tgt, ref_poses = model.extract_poses(batch)
            poses = model.get_video_rendering_path(tgt, ref_poses, "interpolate", 6, batch)
            for i in (1, 4):
                b_i = {k: v.clone() for k, v in batch_cpu.items()}
                b_i["extrinsics"][:, -1, :3] = poses[i]["extrinsics"].cpu()
                ref = O.forward_test(cfg, sd, b_i)
                # same pose rendered by the oracle from the GPU encoder's features: isolates the render path
                feats = model.get_img_feat(batch.images[:, :-1], cur_n_src_views=3)
                pf = [(f[0, :, 0].permute(0, 3, 1, 2).cpu(), f[0, :, 1].permute(0, 3, 1, 2).cpu()) for f in feats]
                h, w = g["images"].shape[-2:]
                te, ti, tn = b_i["extrinsics"][0, -1, :3], b_i["intrinsics"][0, -1], b_i["near_fars"][0, -1]
                se, si, sn = b_i["extrinsics"][0, :-1, :3], b_i["intrinsics"][0, :-1], b_i["near_fars"][0, :-1]
                r2 = O.render_rays(cfg, sd, torch.arange(h * w), te, ti, tn, se, si, sn, b_i["images"][0, :3], pf)
                print(f"  video pose {i} [{math}]: rgb err vs oracle (own encoder) {float((out.rgb[i] - ref['rgb'][0]).abs().max()):.3e}"
                      f"  vs oracle on GPU features {float((out.rgb[i] - r2[0]).abs().max()):.3e}", flush=True)
    os.environ.pop("MNERF_DECODER_MATH", None)

if which in ("all", "overlap"):
    opt, model, _ = bench.build_model(dev)
    _, batch = bench.make_batch(dev, 0)
    n_samples, H, W = 64, 512, 640
    with torch.no_grad():
        feats = model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
        tgt, ref = model.extract_poses(batch)
        ref_host, images_cl = model._frame_ctx(ref, batch.images[:, :3])
        sc = model._scene(0, ref_host, feats, images_cl)
        dec = model._decoder(n_samples, dev)
        t_ex, t_in, t_nf = model._tgt_host(tgt)
        kinv, c2w = camera.target_ray_consts(t_ex[0], t_in[0], True)
        n_rays = H * W
        rgb, dep, opa = torch.empty(n_rays, 3, device=dev), torch.empty(n_rays, device=dev), torch.empty(n_rays, device=dev)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        for chunk in (65536, 32768, 16384, 8192):
            ws = [torch.empty(chunk * n_samples * dec.cond_stride, device=dev) for _ in range(2)]

            def run(overlap):
                n = (n_rays + chunk - 1) // chunk
                cv_done = [torch.cuda.Event() for _ in range(n)]
                dec_done = [torch.cuda.Event() for _ in range(n)]
                lib = hip.load()
                import ctypes as C
                for k in range(n):
                    c = k * chunk
                    m = min(chunk, n_rays - c)
                    rays = hip.make_rays(m, n_samples, H, W, kinv, c2w, t_nf[0, 0], t_nf[0, 1], ray_begin=c)
                    s_cv = sb if overlap else sa
                    if k >= 2:
                        s_cv.wait_event(dec_done[k - 2])
                    hip.check(lib.mnerf_cost_volume(C.byref(sc), C.byref(rays), dec.cond_stride, C.c_void_p(ws[k % 2].data_ptr()),
                                                    C.c_void_p(s_cv.cuda_stream)), "cv")
                    cv_done[k].record(s_cv)
                    sa.wait_event(cv_done[k])
                    hip.check(lib.mnerf_decoder_chunk(C.byref(dec), C.byref(sc.views[0]), C.byref(rays), C.c_void_p(ws[k % 2].data_ptr()),
                                                      C.c_void_p(rgb[c:].data_ptr()), C.c_void_p(dep[c:].data_ptr()),
                                                      C.c_void_p(opa[c:].data_ptr()), None, None, C.c_void_p(sa.cuda_stream)), "dec")
                    dec_done[k].record(sa)
            for overlap in (False, True):
                torch.cuda.synchronize()
                run(overlap)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    run(overlap)
                torch.cuda.synchronize()
                print(f"  render kernels, chunk {chunk:6d}, {'two streams (CV of chunk k+1 under decoder of chunk k)' if overlap else 'one stream'}: "
                      f"{(time.perf_counter() - t0) / 3 * 1e3:.2f} ms/frame (grid {os.environ.get('MNERF_DECODER_GRID', '512')})", flush=True)
