"""Batch producers for the five test / training sets next to DTU: real forward-facing scenes with LLFF-style
`poses_bounds.npy` (LLFF, the user's own COLMAP captures, the IBRNet training collection), NeRF-synthetic (Blender) and
Tanks-and-Temples in MVSNet camera files.  Each yields the sample dict `MatchNeRF.forward` consumes (images [V+1,3,H,W] with
the target LAST, world->camera extrinsics, intrinsics, near_fars, view_ids, scene, img_wh and — where the reference keeps them
for the spiral video path — the camera->world matrices of every training view as `c2ws_all`).

One table-driven base instead of the reference's five parallel classes: a producer fills `self.views[(scene, view)]` with a
`PosedView` and `self.metas` with (scene, target, ordered sources, all training views); image loading, source ordering and the
near/far merge are shared.  Restated from /root/reference/datasets/llff.py:71-242 (+ `center_poses` 12-68), colmap.py:12-173,
blender.py:10-177, ibrnet.py:72-232 and tnt.py:11-190, and PINNED against them: the reference's dataset modules import in the
build container with a three-line torchvision stand-in (`tools/ref_import.py`), so `tools/gen_dataset_golden.py` runs them on
the reference's own shipped scene (`docs/demo_data/printer`) and on small seeded on-disk trees (`tests/dataset_trees.py`) and
commits every sample they produce as `tests/golden/datasets.npz`; `tests/test_scene_sets.py` rebuilds the trees and demands
the same poses, intrinsics and bounds bit for bit and the same images.

Arithmetic notes that matter for bit-equality (the model's geometry is pinned bit-exactly downstream): pose algebra runs in
float64 and is cast to float32 at the end, EXCEPT the world->camera matrix of the poses_bounds.npy sets and the
camera->world matrix of Tanks-and-Temples, which the reference inverts in float32 (llff.py:196-197, tnt.py:114)."""
import json
import os
from collections import namedtuple

import numpy as np
import torch


def read_view_split(pairs_file):
    """`configs/pairs.th`: {<scene>_{train,val,test}: view ids} (looked up like the DTU lists: datasets.resolve_meta)."""
    from .datasets import load_pairs, resolve_meta  # datasets.py imports this module for its registry
    return load_pairs(resolve_meta(pairs_file, "train/test view split (pairs.th)"))

PosedView = namedtuple("PosedView", "intrinsic w2c c2w near_far image")

IMAGE_SUFFIXES = (".jpg", ".JPG", ".jpeg", ".JPEG", ".png", ".PNG", ".ppm", ".PPM", ".bmp", ".BMP", ".tif", ".TIF", ".tiff", ".TIFF")
_FLIP_YZ = np.diag([1, -1, -1, 1])  # NeRF/Blender camera axes (x right, y up, z back) -> OpenCV (x right, y down, z forward)


def list_all_images(root_dir):
    """misc/utils.py:265-275: image file names of a directory, sorted as strings ("10.png" < "2.png")."""
    return sorted(f for f in os.listdir(root_dir) if f.endswith(IMAGE_SUFFIXES))


def load_image(path, img_wh, blend_alpha=False):
    """PIL LANCZOS resize to (w, h), then 8-bit HWC -> float32 CHW / 255 (what torchvision's ToTensor does to an 8-bit image);
    `blend_alpha`: RGBA over white, in float (blender.py:37-41)."""
    from PIL import Image
    img = Image.open(path).resize(tuple(int(x) for x in img_wh), Image.LANCZOS)
    if img.mode not in ("RGB", "RGBA"):
        img = img.convert("RGB")
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(img, np.uint8).transpose(2, 0, 1))).float().div(255.0)
    if blend_alpha:
        if x.shape[0] != 4:
            raise ValueError(f"{path}: expected an RGBA image")
        x = x[:3] * x[-1:] + (1 - x[-1:])
    return x[:3]


def average_pose(poses):
    """llff.py:17-46: centre = mean position, z = normalised mean z axis, x = normalise(mean y x z), y = z x x -> [3,4]."""
    center = poses[..., 3].mean(0)
    z = poses[..., 2].mean(0)
    z = z / np.linalg.norm(z)
    x = np.cross(poses[..., 1].mean(0), z)
    x = x / np.linalg.norm(x)
    return np.stack([x, np.cross(z, x), z, center], 1)


def center_poses(poses):
    """llff.py:49-68: express every [3,4] camera->world in the frame of the average pose, then flip to OpenCV axes."""
    avg = np.eye(4)
    avg[:3] = average_pose(poses)
    bottom = np.tile(np.array([0, 0, 0, 1]), (len(poses), 1, 1))
    homo = np.concatenate([poses, bottom], 1)
    return ((np.linalg.inv(avg) @ homo) @ _FLIP_YZ)[:, :3]


def read_poses_bounds(path, img_wh, center, near_scale):
    """One `poses_bounds.npy` ([N,17]: a 3x5 [R|t|h,w,f] block in LLFF's (down, right, back) axes + near, far) -> per view
    (K float64 [3,3], c2w float64 [4,4], w2c float32 [4,4], bounds float64 [2]).  The scene is rescaled so that the nearest bound
    sits at 1 / near_scale (llff.py:160-186: centred, 0.75; colmap.py:85-107: not centred — the model works in the first source
    view's frame anyway — and 0.47058824)."""
    raw = np.load(path)
    block = raw[:, :15].copy().reshape(-1, 3, 5)
    poses = np.concatenate([block[..., 1:2], -block[..., :1], block[..., 2:4]], -1)  # (down,right,back) -> (right,up,back)
    poses = center_poses(poses) if center else poses @ _FLIP_YZ
    bounds = raw[:, -2:].copy()
    scale = bounds.min() * near_scale
    bounds /= scale
    poses[..., 3] /= scale
    w, h = img_wh
    out = []
    for i in range(len(raw)):
        raw_h, raw_w, focal = raw[:, :15].copy().reshape(-1, 3, 5)[i, :, -1]
        k = np.array([[focal * w / raw_w, 0, w / 2], [0, focal * h / raw_h, h / 2], [0, 0, 1]])
        c2w = np.eye(4)
        c2w[:3] = poses[i]
        out.append((k, c2w, np.linalg.inv(c2w.astype(np.float32)), bounds[i]))
    return out


def read_mvsnet_cam(path):
    """tnt.py:124-137: `extrinsic` 4x4 on lines 1-4, `intrinsic` 3x3 on lines 7-9, line 11 = depth_min ... depth_max
    (first and LAST number; DTU's files carry an interval second — datasets.read_cam_file — these carry the far plane last)."""
    with open(path) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extr = np.array(" ".join(lines[1:5]).split(), np.float32).reshape(4, 4)
    intr = np.array(" ".join(lines[7:10]).split(), np.float32).reshape(3, 3)
    d = lines[11].split()
    return intr, extr, float(d[0]), float(d[-1])


class PosedImageSet(torch.utils.data.Dataset):
    """Shared machinery; subclasses fill `self.views` and `self.metas` in `__init__`."""

    name = "posed"
    nf_merge = None  # None: per-view bounds; "avg" / "minmax": one pair for all views of a sample (llff.py:236, colmap.py:156-165)
    keep_all_c2ws = False
    blend_alpha = False

    def __init__(self, root_dir, split, n_views=3, img_wh=None, max_len=-1):
        self.root_dir, self.split, self.n_views, self.img_wh, self.max_len = root_dir, split, n_views, img_wh, max_len
        self.views, self.metas = {}, []

    def get_name(self):
        return self.name

    def __len__(self):
        return len(self.metas) if self.max_len <= 0 else self.max_len

    # ---- helpers for subclasses
    def scene_dirs(self, scene_list):
        if scene_list is not None:
            return list(scene_list)
        return sorted(x for x in os.listdir(self.root_dir) if os.path.isdir(os.path.join(self.root_dir, x)))

    def order_sources(self, scene, target, train_views, method):
        """'nearest': training views by L1 distance of the camera centres to the target's; 'fixed': as listed."""
        if method == "fixed":
            return train_views
        if method != "nearest":
            raise ValueError(f"Unknown evaluate method [{method}]")
        pos = np.stack([self.views[scene, v].c2w for v in train_views])[:, :3, 3]
        dist = np.sum(np.abs(pos - self.views[scene, target].c2w[:3, 3]), axis=-1)
        return [train_views[i] for i in np.argsort(dist)]

    def add_targets(self, scene, train_views, test_views, method):
        for t in test_views:
            self.metas.append((scene, t, self.order_sources(scene, t, train_views, method), train_views))

    # ---- sample assembly
    def image_path(self, scene, view):
        return os.path.join(self.root_dir, scene, "images", self.views[scene, view].image)

    def view_intrinsic(self, scene, view, original_size):
        return self.views[scene, view].intrinsic

    def pick_views(self, sources, target):
        return [sources[i] for i in range(self.n_views)] + [target]

    def view_id_array(self, view_ids):
        return np.array(view_ids)

    def scene_label(self, scene):
        return scene

    def __getitem__(self, idx):
        from PIL import Image
        scene, target, sources, train_views = self.metas[idx]
        view_ids = self.pick_views(sources, target)
        img_wh = np.array(self.img_wh).astype("int")
        imgs, ks = [], []
        for v in view_ids:
            path = self.image_path(scene, v)
            with Image.open(path) as probe:
                original = probe.size
            imgs.append(load_image(path, img_wh, self.blend_alpha))
            ks.append(self.view_intrinsic(scene, v, original))
        nfs = np.stack([self.views[scene, v].near_far for v in view_ids])
        if self.nf_merge == "avg":
            nfs = np.expand_dims(np.average(nfs, axis=0), axis=0).repeat(len(view_ids), axis=0)
        elif self.nf_merge == "minmax":  # widened hull of all views' bounds
            nfs = np.expand_dims(np.array([nfs.min() * 0.8, nfs.max() * 1.2]), axis=0).repeat(len(view_ids), axis=0)
        elif self.nf_merge is not None:
            raise ValueError(f"Unknown near far mode {self.nf_merge}")
        sample = {
            "images": torch.stack(imgs).float(),
            "extrinsics": np.stack([self.views[scene, v].w2c for v in view_ids]).astype(np.float32),
            "intrinsics": np.stack(ks).astype(np.float32),
            "near_fars": nfs.astype(np.float32),
            "view_ids": self.view_id_array(view_ids),
            "scene": self.scene_label(scene),
            "img_wh": img_wh,
        }
        if self.keep_all_c2ws:
            sample["c2ws_all"] = np.stack([self.views[scene, v].c2w for v in train_views]).astype(np.float32)
        return sample


class MVSDatasetRealFF(PosedImageSet):
    """llff.py:71-242.  Real forward-facing scenes: `<root>/<scene>/{poses_bounds.npy, images/}`; train / test views from
    `configs/pairs.th` (`eval_mode='mvsnerf'`) or every 8th image held out (`'gpnr'`); one near/far pair per sample = the mean."""

    name = "llff"
    nf_merge = "avg"
    keep_all_c2ws = True
    center, near_scale = True, 0.75

    def __init__(self, root_dir, split, n_views=3, img_wh=None, downSample=1.0, max_len=-1, scene_list=None,
                 test_views_method="nearest", eval_mode="mvsnerf", pairs_file=os.path.join("configs", "pairs.th"), **kwargs):
        if split != "test":
            raise ValueError('Only support "test" split for this dataset!')
        super().__init__(root_dir, split, n_views, img_wh, max_len)
        self.eval_mode = eval_mode
        pairs = self.load_view_split(pairs_file)
        for scene in self.scene_dirs(scene_list):
            train_views, test_views = self.split_views(scene, pairs)
            self.add_scene_cameras(scene)
            self.add_targets(scene, train_views, test_views, test_views_method)

    def load_view_split(self, pairs_file):
        return read_view_split(pairs_file) if self.eval_mode == "mvsnerf" else None

    def split_views(self, scene, pairs):
        if self.eval_mode == "mvsnerf":  # MVSNeRF's protocol: 'val' ids are the test targets
            return pairs[f"{scene}_train"], pairs[f"{scene}_val"]
        if self.eval_mode == "gpnr":
            n = len(list_all_images(os.path.join(self.root_dir, scene, "images")))
            test = list(range(0, n, 8))
            return [x for x in range(n) if x not in test], test
        raise ValueError(f"Unknown eval_mode {self.eval_mode}.")

    def add_scene_cameras(self, scene):
        scene_dir = os.path.join(self.root_dir, scene)
        files = list_all_images(os.path.join(scene_dir, "images"))
        cams = read_poses_bounds(os.path.join(scene_dir, "poses_bounds.npy"), self.img_wh, self.center, self.near_scale)
        for i, (k, c2w, w2c, nf) in enumerate(cams):
            if i < len(files):
                self.views[scene, i] = PosedView(k, w2c, c2w, nf, files[i])


def colmap_view_split(root_dir, n_select=20, n_interval=6):
    """colmap.py:12-48: per scene directory, the `n_select` cameras nearest (L1) to the mean position; every `n_interval`-th of
    them is a test target, the rest are sources.  Up to three images: target 0, sources (2, 1, 0)."""
    pairs = {}
    for scene in sorted(os.listdir(root_dir)):
        meta = os.path.join(root_dir, scene, "poses_bounds.npy")
        if not os.path.isdir(os.path.join(root_dir, scene)):
            continue
        if not os.path.isfile(meta):
            raise FileNotFoundError(f"Please run COLMAP for {os.path.join(root_dir, scene)} first, using imgs2pose from LLFF project.")
        block = np.load(meta)[:, :15].reshape(-1, 3, 5)
        n = block.shape[0]
        if n <= 3:
            pairs[f"{scene}_test"], pairs[f"{scene}_val"], pairs[f"{scene}_train"] = [0], [0], [2, 1, 0]
            continue
        n_sel, n_int = min(n, int(n_select)), min(n, int(n_interval))
        pos = np.concatenate([block[..., 1:2], -block[..., :1], block[..., 2:4]], -1)[..., 3]
        near = np.argsort(np.sum(np.abs(pos - np.mean(pos, axis=0, keepdims=True)), axis=-1))[:n_sel]
        pairs[f"{scene}_test"] = pairs[f"{scene}_val"] = [int(x) for x in near[::n_int]]
        pairs[f"{scene}_train"] = [int(x) for x in np.delete(near, range(0, n_sel, n_int))]
    return pairs


class MVSDatasetCOLMAP(MVSDatasetRealFF):
    """colmap.py:51-173: the user's own captures (`configs/demo_own.yaml`).  Views are split on the fly, poses are not centred,
    the near bound lands at 1 / 0.47058824 = 2.125 (DTU's), `nf_mode` picks mean or widened hull; `test_views_method='fixed'`
    keeps a single target per scene (video rendering)."""

    name = "colmap"
    center, near_scale = False, 0.47058824

    def __init__(self, root_dir, split, n_views=3, img_wh=None, downSample=1.0, max_len=-1, scene_list=None,
                 test_views_method="nearest", nf_mode="avg", **kwargs):
        self.nf_merge = nf_mode
        self._fixed = test_views_method == "fixed"
        super().__init__(root_dir, split, n_views, img_wh, downSample, max_len, scene_list, test_views_method, "mvsnerf")

    def load_view_split(self, pairs_file):
        pairs = colmap_view_split(self.root_dir, 20, 6)
        if self._fixed:
            pairs = {k: (v[:1] if k.endswith("_val") else v) for k, v in pairs.items()}
        return pairs


class MVSDatasetIBRNet(MVSDatasetRealFF):
    """ibrnet.py:72-232: `<root>/<collection>/<scene>/{poses_bounds.npy, images/}`; every image is a target in turn ('train')
    or only image 0 ('val'); sources = the others by distance; training draws n_views of the nearest n_views + 3 at random."""

    name = "ibrnet"
    keep_all_c2ws = False

    def __init__(self, root_dir, split, n_views=3, img_wh=None, downSample=1.0, max_len=-1, scene_list=None,
                 test_views_method="nearest", **kwargs):
        if split not in ("train", "val"):
            raise ValueError('Only support "train" and "val" split for IBRNet dataset!')
        if test_views_method != "nearest":
            raise ValueError(f"Unknown evaluate method [{test_views_method}]")
        PosedImageSet.__init__(self, root_dir, split, n_views, img_wh, max_len)
        from glob import glob
        for group in glob(os.path.join(root_dir, "*/")):
            for scene_path in glob(os.path.join(group, "*/")):
                self.add_scene_cameras(scene_path)
                n = np.load(os.path.join(scene_path, "poses_bounds.npy")).shape[0]
                for t in (range(n) if split == "train" else [0]):
                    others = [x for x in range(n) if x != t]
                    self.metas.append((scene_path, t, self.order_sources(scene_path, t, others, "nearest"), others))

    def add_scene_cameras(self, scene_path):
        files = list_all_images(os.path.join(scene_path, "images"))
        for i, (k, c2w, w2c, nf) in enumerate(read_poses_bounds(os.path.join(scene_path, "poses_bounds.npy"), self.img_wh, True, 0.75)):
            self.views[scene_path, i] = PosedView(k, w2c, c2w, nf, files[i])

    def image_path(self, scene, view):
        return os.path.join(scene, "images", self.views[scene, view].image)

    def scene_label(self, scene_path):
        return "_".join(scene_path.strip("/").split("/")[-2:])

    def pick_views(self, sources, target):
        if self.split == "train":
            ids = torch.sort(torch.randperm(self.n_views + 3)[:self.n_views])[0]
            return [sources[int(i)] for i in ids] + [target]
        return sources[:self.n_views] + [target]


class MVSDatasetBlender(PosedImageSet):
    """blender.py:10-177: NeRF-synthetic `<root>/<scene>/transforms_{train,test}.json` + RGBA PNGs (blended over white), one
    pinhole shared by all views (800-pixel renders), near / far = 2 / 6.  'mvsnerf': sources AND targets are frames of the raw
    train split picked by `pairs.th`; 'gpnr': sources = the train/ folder, targets = the test/ folder (views named by split)."""

    name = "blender"
    blend_alpha = True

    def __init__(self, root_dir, split, n_views=3, img_wh=None, downSample=1.0, max_len=-1, scene_list=None,
                 test_views_method="nearest", eval_mode="mvsnerf", pairs_file=os.path.join("configs", "pairs.th"), **kwargs):
        if split != "test":
            raise ValueError('Only support "test" split for blender dataset!')
        if eval_mode not in ("mvsnerf", "gpnr"):
            raise ValueError("Only support mvsnerf and gpnr test mode.")
        if img_wh is not None and (img_wh[0] % 32 or img_wh[1] % 32):
            raise ValueError("img_wh must both be multiples of 32!")
        super().__init__(root_dir, split, n_views, img_wh, max_len)
        self.eval_mode = eval_mode
        pairs = read_view_split(pairs_file)
        for scene in self.scene_dirs(scene_list):
            if eval_mode == "mvsnerf":
                train_views, test_views = pairs[f"{scene}_train"], pairs[f"{scene}_val"]
                self.add_frames(scene, "transforms_train.json", [*train_views, *test_views], lambda v: v)
            else:
                train_views = [f"train_{i}" for i in self.folder_indices(scene, "train")]
                test_views = [f"test_{i}" for i in self.folder_indices(scene, "test")]
                self.add_frames(scene, "transforms_train.json", train_views, lambda v: int(v.split("_")[-1]))
                self.add_frames(scene, "transforms_test.json", test_views, lambda v: int(v.split("_")[-1]))
            for t in test_views:
                self.metas.append((scene, t, self.order_sources(scene, t, train_views, test_views_method), train_views))

    def folder_indices(self, scene, folder):
        names = [x for x in os.listdir(os.path.join(self.root_dir, scene, folder)) if x.endswith("png")]
        return sorted({int(x.split(".")[0].split("_")[-1]) for x in names})

    def add_frames(self, scene, meta_name, views, frame_of):
        with open(os.path.join(self.root_dir, scene, meta_name)) as f:
            meta = json.load(f)
        w, h = self.img_wh
        focal = 0.5 * 800.0 / np.tan(0.5 * meta["camera_angle_x"])
        focal = focal * w / 800.0
        k = np.array([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1]])
        for v in views:
            frame = meta["frames"][frame_of(v)]
            c2w = np.array(frame["transform_matrix"]) @ _FLIP_YZ
            self.views[scene, v] = PosedView(k, np.linalg.inv(c2w), c2w, [2.0, 6.0], f"{frame['file_path']}.png")

    def image_path(self, scene, view):
        return os.path.join(self.root_dir, scene, self.views[scene, view].image)

    def view_id_array(self, view_ids):
        if isinstance(view_ids[0], str):
            view_ids = [int(x.split("_")[-1]) for x in view_ids]
        return np.array(view_ids)


class MVSDatasetTNT(PosedImageSet):
    """tnt.py:11-190: Tanks and Temples in MVSNet layout `<root>/<scene>/{cams_1/%08d_cam.txt, images/%08d.jpg}`; translations
    and depth bounds are multiplied by 500; the intrinsics are given for the ORIGINAL image and rescaled per image to img_wh."""

    name = "tnt"
    keep_all_c2ws = True
    scale_factor = 500.0

    def __init__(self, root_dir, split, n_views=3, img_wh=None, downSample=1.0, max_len=-1, scene_list=None,
                 test_views_method="nearest", eval_mode="mvsnerf", nf_mode="avg",
                 pairs_file=os.path.join("configs", "pairs.th"), **kwargs):
        if split != "test":
            raise ValueError('Only support "test" split for TNT dataset!')
        super().__init__(root_dir, split, n_views, img_wh, max_len)
        self.nf_merge = nf_mode
        pairs = read_view_split(pairs_file) if eval_mode == "mvsnerf" else None
        for scene in self.scene_dirs(scene_list):
            if eval_mode == "mvsnerf":
                train_views, test_views = pairs[f"TNT_{scene}_train"], pairs[f"TNT_{scene}_val"]
            elif eval_mode == "gpnr":
                n = len(list_all_images(os.path.join(root_dir, scene, "images")))
                test_views = list(range(0, n, 8))
                train_views = [x for x in range(n) if x not in test_views]
            else:
                raise ValueError(f"Unknown eval_mode {eval_mode}.")
            for v in [*train_views, *test_views]:
                intr, extr, dmin, dmax = read_mvsnet_cam(os.path.join(root_dir, scene, "cams_1", f"{v:08d}_cam.txt"))
                extr[:3, 3] *= self.scale_factor
                self.views[scene, v] = PosedView(intr, extr, np.linalg.inv(extr.astype(np.float32)),
                                                 np.array([dmin * self.scale_factor, dmax * self.scale_factor]), f"{v:08d}.jpg")
            self.add_targets(scene, train_views, test_views, test_views_method)

    def view_intrinsic(self, scene, view, original_size):
        k = self.views[scene, view].intrinsic.copy()
        wh = np.array(self.img_wh).astype("int")
        k[0] *= wh[0] / original_size[0]  # a float64 numpy scalar: the float32 row is scaled in double, then rounded (tnt.py:163-165)
        k[1] *= wh[1] / original_size[1]
        return k
