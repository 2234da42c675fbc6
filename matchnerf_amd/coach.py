"""Thin orchestration counterpart of the reference's Coach for INFERENCE (coach.py:27-146,
368-529): build the network from the registry, restore a checkpoint per child, iterate test
batches, call the hot path, report PSNR.  The training loop, TensorBoard, SSIM/LPIPS and the
on-disk datasets are callers / data formats outside the hot path — a dataset is
anything that yields batches with the reference's contract (images, extrinsics, intrinsics,
near_fars[, depth, scene, view_ids]); ``synthetic`` is built in so the tool runs offline."""
import os

import numpy as np
import torch

from . import checkpoint, datasets, metrics, synthetic
from .edict import EasyDict as edict
from .models import models_dict


class SyntheticScenes:
    """Seeded stand-in for a dataset (no DTU/LLFF/Blender data offline)."""

    def __init__(self, name, cfg, n_src_views):
        self.name = name
        w, h = cfg.get("img_wh", [64, 64])
        self.kw = dict(height=h, width=w, n_src_views=n_src_views, wide=(name == "blender"),
                       near_far=(2.0, 6.0) if name == "blender" else (2.125, 4.525))
        n = cfg.get("max_len", -1)
        self.n = 2 if n in (None, -1) else n

    def get_name(self):
        return self.name

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            sc = synthetic.make_scene(seed=100 + i, **self.kw)
            batch = {k: torch.from_numpy(v) for k, v in sc.items()}
            batch["scene"] = [f"synthetic{i}"]
            batch["view_ids"] = torch.arange(self.kw["n_src_views"] + 1)[None]
            yield batch


class Coach:
    def __init__(self, opts):
        self.opts = opts
        self.n_src_views = opts.n_src_views
        self.device = opts.device

    def build_networks(self):
        self.model = models_dict[self.opts.model](self.opts).to(self.opts.device)  # coach.py:77

    def restore_checkpoint(self):
        path = self.opts.load
        if path and os.path.isfile(path):
            checkpoint.restore_checkpoint(self.model, path, self.opts.device)
        else:
            print(f"[coach] checkpoint {path!r} not found: using seeded random weights (offline run)")
            spec = synthetic.state_dict_spec(n_src_views=self.n_src_views)
            self.model.load_state_dict(synthetic.to_torch(synthetic.seeded_state_dict(spec, 1), self.opts.device))

    def load_dataset(self, splits=("test",), loaders=None):
        """``loaders``: optional list of iterables of batches (objects with get_name()); otherwise
        every ``data_test`` entry whose ``root_dir`` exists is read from disk (datasets.py) and the others are served by the
        synthetic generator at that entry's img_wh."""
        if loaders is not None:
            self.test_loaders = list(loaders)
            return
        self.test_loaders = []
        for name, cfg in self.opts.data_test.items():
            if cfg is None:
                continue
            root, kind = cfg.get("root_dir"), cfg.get("dataset_name", name)
            if root and os.path.isdir(root) and kind in datasets.datas_dict:  # coach.py:52-69: the on-disk producer
                extra = {k: cfg[k] for k in ("meta_dir", "pairs_file") if cfg.get(k)}  # where the scan / pair lists live
                ds = datasets.datas_dict[kind](root, "test", n_views=self.n_src_views, img_wh=cfg.get("img_wh"),
                                               max_len=cfg.get("max_len", -1), scene_list=cfg.get("scene_list"),
                                               test_views_method=cfg.get("test_views_method", "nearest"),
                                               nf_mode=cfg.get("nf_mode", "avg"), eval_mode=cfg.get("eval_mode", "mvsnerf"),
                                               n_add_train_views=cfg.get("n_add_train_views", 2), **extra)
                loader = torch.utils.data.DataLoader(ds, shuffle=False, num_workers=cfg.get("num_workers", 0),
                                                     batch_size=self.opts.batch_size, pin_memory=True)
                loader.get_name = ds.get_name
                self.test_loaders.append(loader)
            else:
                self.test_loaders.append(SyntheticScenes(name, cfg, self.n_src_views))

    @torch.no_grad()
    def test_model(self, save_images=False, **kwargs):
        """coach.py:368-453 -> {dataset: {image_id: psnr}}; the results file also lists SSIM and, when the two weight files of the
        `lpips` package are on disk (metrics.load_lpips: $MNERF_LPIPS_VGG16 / $MNERF_LPIPS_LIN or torch hub's cache), LPIPS."""
        self.model.eval()
        try:
            lpips_fn = metrics.load_lpips(device=self.opts.device)
        except FileNotFoundError:
            lpips_fn = None
        out_root = os.path.join(self.opts.output_path, "test")
        os.makedirs(out_root, exist_ok=True)
        report = {}
        for loader in self.test_loaders:
            name = loader.get_name()
            report[name] = {}
            ssims, lpipss = [], []
            self.model.nerf_setbg_opaque = (name == "blender")  # coach.py:382-383
            for bi, batch in enumerate(loader):
                var = edict({k: (v.to(self.opts.device) if torch.is_tensor(v) else v) for k, v in batch.items()})
                gt_depth = var.pop("depth") if "depth" in var else None  # forward overwrites 'depth'
                var = self.model(var, mode="test")
                b, _, _, h, w = var.images.shape
                pred = var.rgb.reshape(b, h, w, 3).cpu().numpy()
                gt = var.images[:, -1].permute(0, 2, 3, 1).cpu().numpy()
                for i in range(b):
                    mask = None if gt_depth is None else (gt_depth[i].cpu().numpy() == 0)
                    report[name][f"{name}_{bi:03d}_{i}"] = metrics.psnr(pred[i], gt[i], mask)
                    tools = metrics.EvalTools(lpips_fn=lpips_fn)
                    tools.set_inputs(pred[i], gt[i], mask)
                    got = tools.get_metrics(["SSIM"] + (["LPIPS"] if lpips_fn else []))
                    ssims.append(got["SSIM"])
                    lpipss.append(got.get("LPIPS"))
                    if save_images:
                        from PIL import Image
                        vis = np.concatenate([pred[i], gt[i]], 1)
                        Image.fromarray((vis.clip(0, 1) * 255).astype("uint8")).save(
                            os.path.join(out_root, f"{name}_{bi:03d}_{i}.png"))
            self.model.nerf_setbg_opaque = False
            vals = list(report[name].values())
            with open(os.path.join(out_root, f"0results_{name}.txt"), "w") as f:
                for (k, v), sv, lv in zip(report[name].items(), ssims, lpipss):
                    f.write(f"{k}: PSNR {v:.4f} SSIM {sv:.4f}" + (f" LPIPS {lv:.4f}" if lv is not None else "") + "\n")
                f.write(f"mean PSNR {np.mean(vals):.4f} SSIM {np.mean(ssims):.4f}" +
                        (f" LPIPS {np.mean(lpipss):.4f}" if lpips_fn else "") + "\n")
            print(f"[coach] {name}: mean PSNR {np.mean(vals):.2f} over {len(vals)} images")
        return report

    @torch.no_grad()
    def test_model_video(self, **kwargs):
        """coach.py:455-529: every batch of every test set rendered along its video path (dtu / blender / colmap-by-config:
        interpolate, llff: spiral; white background for blender), written under <output_path>/test_videos/<set>/ as the reference
        names them: `<scene>_view<tgt>_src<ids>.gif` at 12 fps when nerf.save_gif, `..._f<i>.jpg` frames when nerf.save_frames, and
        the strip of source views `<name>.jpg` (PIL instead of imageio; the reference's .mp4 needs scikit-video / ffmpeg, neither in
        this image, and is skipped).  Returns {set: frames [F,H,W,3] uint8 of its FIRST batch element}."""
        from PIL import Image
        self.model.eval()
        out_root = os.path.join(self.opts.output_path, "test_videos")
        videos = {}
        for loader in self.test_loaders:
            name = loader.get_name()
            out_dir = os.path.join(out_root, name)
            os.makedirs(out_dir, exist_ok=True)
            if name == "llff":
                mode = "spiral"
            elif name == "colmap":
                mode = getattr(getattr(self.opts.data_test, "colmap", {}), "render_path_mode", "interpolate")
            elif name in ("dtu", "blender"):
                mode = "interpolate"
            else:  # coach.py:480-481: the reference has no video rule for the other sets (ibrnet, tnt, ...)
                raise Exception(f"Unknown dataset for rendering video {name}")
            self.model.nerf_setbg_opaque = (name == "blender")
            n_frames = int(self.opts.nerf.video_n_frames)
            for batch in loader:
                var = edict({k: (v.to(self.opts.device) if torch.is_tensor(v) else v) for k, v in batch.items()})
                # (coach.py:487-488 passes opts.nerf.render_video; this method IS the video test, so a config that leaves the
                # switch off still gets the frames - and is told so once)
                if not getattr(self.opts.nerf, "render_video", True) and not getattr(self, "_warned_render_video", False):
                    import warnings
                    warnings.warn("test_model_video: nerf.render_video is off in the options; rendering the video path anyway")
                    self._warned_render_video = True
                if getattr(self.opts, "vis_depth", False) and not getattr(self, "_warned_vis_depth", False):
                    import warnings
                    warnings.warn("test_model_video: vis_depth (depth maps next to the frames, coach.py:497-505) and the .mp4 "
                                  "container (skvideo, coach.py:511-525) are not written: frames / GIF / source strip only")
                    self._warned_vis_depth = True
                var = self.model(var, mode="test", render_video=True, render_path_mode=mode)
                b, _, _, h, w = var.images.shape
                # forward returns the reference's frame-major layout [n_frames * B, HW, 3]
                frames = (var.rgb.reshape(n_frames, b, h, w, 3).clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
                for bi in range(b):
                    clip = frames[:, bi]
                    videos.setdefault(name, clip)
                    ids = [int(x) for x in batch["view_ids"][bi]] if "view_ids" in batch else list(range(self.n_src_views + 1))
                    scene = batch["scene"][bi] if "scene" in batch else f"scene{bi}"
                    stem = f"{scene}_view{ids[-1]:02d}_src" + "_".join(f"{x:02d}" for x in ids[:self.n_src_views])
                    if getattr(self.opts.nerf, "save_frames", False):
                        for fi, fr in enumerate(clip):
                            Image.fromarray(fr).save(os.path.join(out_dir, f"{stem}_f{fi}.jpg"))
                    if getattr(self.opts.nerf, "save_gif", False):
                        imgs = [Image.fromarray(fr) for fr in clip]
                        imgs[0].save(os.path.join(out_dir, f"{stem}.gif"), save_all=True, append_images=imgs[1:], duration=1000 // 12, loop=0)
                    src = (var.images[bi, :self.n_src_views].permute(0, 2, 3, 1).cpu().numpy() * 255).astype("uint8")
                    Image.fromarray(np.concatenate(list(src), axis=1)).save(os.path.join(out_dir, f"{stem}.jpg"))
            self.model.nerf_setbg_opaque = False
        return videos
