"""Batch producers and on-disk formats of the DTU benchmark: the MVSNet camera file, the PFM depth map,
the view-pair lists and the `MVSDatasetDTU` sample layout that `MatchNeRF.forward` consumes.

Restates /root/reference/datasets/dtu.py (class MVSDatasetDTU, lines 12-209) and misc/utils.py:278-313 (read_pfm) on
numpy + PIL only — OpenCV and torchvision are not part of this image, so the two places the reference uses them are written
out: `T.ToTensor()` is uint8 HWC -> float CHW / 255, and `cv2.resize(.., fx=0.5, fy=0.5, INTER_NEAREST)` of an even-sized map
picks every second row and column (source index floor(dst * 2)).

PARITY: the reference's own `MVSDatasetDTU` imports in the build container (tools/ref_import.py: in-memory stand-ins for the
absent cv2 / torchvision modules), so `tools/gen_dataset_golden.py` runs it on a seeded MVSNet-layout tree (tests/dataset_trees.py:
camera files, list files, 7-light images, 1200x1600 PFM depth maps with holes) for the test, val and train splits and
`tests/test_scene_sets.py` demands the same samples from this class bit for bit.  The one thing that pin cannot reach is OpenCV's
INTER_NEAREST rule itself: the reference run uses `nearest_resize` below as its `cv2.resize` (there is no OpenCV here), so that
function is checked against hand-built arrays only.  The list files themselves (`configs/dtu_meta/*.txt`, `configs/pairs.th`) are
the user's data and are read from the paths the reference reads them from, relative to the working directory, unless `meta_dir` / `pairs_file` say otherwise."""
import os
import re

import numpy as np
import torch

DTU_SCALE = 1.0 / 200  # dtu.py:29: millimetres -> the unit the model was trained in


def read_pfm(filename):
    """misc/utils.py:278-313 -> (array [H,W] or [H,W,3], bottom row first in the file => flipped to top first; scale)."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise ValueError(f"{filename}: not a PFM file (header {header!r})")
        dims = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not dims:
            raise ValueError(f"{filename}: malformed PFM header")
        width, height = int(dims.group(1)), int(dims.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.fromfile(f, endian + "f")
    shape = (height, width, 3) if header == "PF" else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)


def write_pfm(filename, array, scale=1.0):
    """Inverse of read_pfm (little-endian), for tests and for dumping predicted depth in the same format."""
    a = np.flipud(np.asarray(array, np.float32))
    with open(filename, "wb") as f:
        f.write(b"PF\n" if a.ndim == 3 else b"Pf\n")
        f.write(f"{a.shape[1]} {a.shape[0]}\n".encode())
        f.write(f"{-abs(scale)}\n".encode())
        a.astype("<f4").tofile(f)


def read_cam_file(filename, scale_factor=DTU_SCALE, n_depth_planes=192):
    """dtu.py:108-123: MVSNet `*_cam.txt` -> (K [3,3] f32, world->camera [4,4] f32, [near, far]).  Lines 1-4 hold the
    extrinsic, 7-9 the intrinsic, 11 `depth_min depth_interval`; far = near + interval * 192 planes, both scaled."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsic = np.array(" ".join(lines[1:5]).split(), np.float32).reshape(4, 4)
    intrinsic = np.array(" ".join(lines[7:10]).split(), np.float32).reshape(3, 3)
    d = lines[11].split()
    depth_min = float(d[0]) * scale_factor
    depth_max = depth_min + float(d[1]) * n_depth_planes * scale_factor
    return intrinsic, extrinsic, [depth_min, depth_max]


def read_view_pairs(filename):
    """MVSNet `view_pairs.txt` (dtu.py:75-86): count, then per reference view its id and `n id score id score ..`."""
    pairs = []
    with open(filename) as f:
        n = int(f.readline())
        for _ in range(n):
            ref = int(f.readline().rstrip())
            pairs.append((ref, [int(x) for x in f.readline().rstrip().split()[1::2]]))
    return pairs


def load_pairs(filename):
    """`configs/pairs.th` (dtu.py:45-47): a pickled dict `{<scene>_{train,test,val}: ids}` -> dict of int lists."""
    d = torch.load(filename, weights_only=False)
    return {k: [int(x) for x in v] for k, v in d.items()}


def nearest_resize_half(a):
    """cv2.resize(a, None, fx=0.5, fy=0.5, interpolation=INTER_NEAREST) (dtu.py:127)."""
    h, w = a.shape[:2]
    return a[(np.arange(int(round(h * 0.5))) * 2).clip(max=h - 1)][:, (np.arange(int(round(w * 0.5))) * 2).clip(max=w - 1)]


def nearest_resize(a, fx, fy):
    """cv2 INTER_NEAREST for a scale factor: destination pixel d reads source floor(d / f)."""
    h, w = a.shape[:2]
    nh, nw = int(round(h * fy)), int(round(w * fx))
    ys = np.minimum(np.floor(np.arange(nh) / fy).astype(int), h - 1)
    xs = np.minimum(np.floor(np.arange(nw) / fx).astype(int), w - 1)
    return a[ys][:, xs]


def resolve_meta(path, what):
    """The scene / pair lists (`configs/dtu_meta/*.txt`, `configs/pairs.th`) are DATA the reference keeps in its repository, not
    part of this package.  A relative path is looked up in the working directory first, then next to this package's configs
    (matchnerf_amd/configs/..., the same order options.load_options uses for yaml files), then under $MNERF_META_ROOT.  A miss
    names every place that was tried and the constructor arguments that override the defaults."""
    if os.path.isabs(path):
        tried = [path]
    else:
        pkg = os.path.dirname(os.path.abspath(__file__))
        rel = path[len("configs" + os.sep):] if path.startswith("configs" + os.sep) else path
        tried = [os.path.abspath(path), os.path.join(pkg, "configs", rel)]
        if os.environ.get("MNERF_META_ROOT"):
            tried.append(os.path.join(os.environ["MNERF_META_ROOT"], path))
    for t in tried:
        if os.path.exists(t):
            return t
    raise FileNotFoundError(
        f"{what} not found (tried: {', '.join(tried)}).  These lists ship with the reference repository "
        "(configs/dtu_meta/{train_all,val_all,view_pairs}.txt, configs/pairs.th): copy them next to the working directory, "
        "point MNERF_META_ROOT at a checkout, or pass meta_dir= / pairs_file= (data_test.<name>.meta_dir / .pairs_file in the yaml).")


class MVSDatasetDTU(torch.utils.data.Dataset):
    """dtu.py:12-209.  `img_wh` = (640, 512) for the benchmark; a sample is the dict `MatchNeRF.forward` takes
    (images [V+1,3,H,W] with the target LAST, extrinsics world->camera [V+1,4,4], intrinsics [V+1,3,3], near_fars [V+1,2],
    view_ids, scene, img_wh and, for val/test, the target's depth [H,W])."""

    def __init__(self, root_dir, split, n_views=3, img_wh=None, downSample=1.0, max_len=-1, test_views_method="nearest",
                 n_add_train_views=2, meta_dir=os.path.join("configs", "dtu_meta"), pairs_file=os.path.join("configs", "pairs.th"),
                 **kwargs):
        if split not in ("train", "val", "test"):
            raise ValueError('split must be either "train", "val" or "test"!')
        if img_wh is not None and (img_wh[0] % 32 or img_wh[1] % 32):
            raise ValueError("img_wh must both be multiples of 32!")
        self.root_dir, self.split, self.n_views, self.img_wh = root_dir, split, n_views, img_wh
        self.downSample, self.max_len = downSample, max_len
        self.scale_factor = DTU_SCALE
        self.val_light_idx, self.val_view_idx = 3, 24
        self.n_add_train_views = n_add_train_views
        self.permute_train_src = True
        if split in ("train", "val"):
            self.metas, id_list = self.build_train_metas(resolve_meta(os.path.join(meta_dir, "train_all.txt"), "DTU scan list"),
                                                         resolve_meta(os.path.join(meta_dir, "view_pairs.txt"), "DTU view-pair list"))
            self.build_camera_info(id_list)
        else:
            pairs = load_pairs(resolve_meta(pairs_file, "train/test view split (pairs.th)"))
            train_views, test_views = pairs["dtu_train"], pairs["dtu_test"]
            val_list = resolve_meta(os.path.join(meta_dir, "val_all.txt"), "DTU test scan list")
            self.build_camera_info([*train_views, *test_views])
            self.metas = self.build_test_metas(val_list, train_views, test_views, method=test_views_method)

    def get_name(self):
        return "dtu"

    @staticmethod
    def _scans(path):
        with open(path) as f:
            return [line.rstrip() for line in f.readlines()]

    def build_train_metas(self, scene_list_filepath, view_pairs_filepath):
        """dtu.py:62-90: every scan x reference view x light (7 lights for train; light 3 and view 24 only for val)."""
        metas, id_list = [], []
        lights = range(7) if self.split == "train" else [self.val_light_idx]
        pairs = read_view_pairs(view_pairs_filepath)
        for scan in self._scans(scene_list_filepath):
            for ref_view, src_views in pairs:
                for light in lights:
                    if self.split == "val" and ref_view != self.val_view_idx:
                        continue
                    metas.append((scan, light, ref_view, src_views))
                    id_list.append([ref_view] + src_views)
        return metas, np.unique(id_list)

    def build_camera_info(self, id_list):
        """dtu.py:92-106: intrinsics are given at 1/4 resolution (x4), translations in mm (x 1/200)."""
        self.intrinsics_dict, self.world2cams_dict, self.cam2worlds_dict, self.near_fars_dict = {}, {}, {}, {}
        for vid in id_list:
            vid = int(vid)
            k, e, nf = read_cam_file(os.path.join(self.root_dir, f"Cameras/train/{vid:08d}_cam.txt"), self.scale_factor)
            k[:2] *= 4
            k[:2] = k[:2] * self.downSample
            e[:3, 3] *= self.scale_factor
            self.intrinsics_dict[vid], self.world2cams_dict[vid] = k, e
            self.cam2worlds_dict[vid] = np.linalg.inv(e)
            self.near_fars_dict[vid] = nf

    def read_depth(self, filename):
        """dtu.py:125-130: 1200x1600 -> nearest half -> crop to 512x640 -> nearest downSample."""
        depth = np.array(read_pfm(filename)[0], dtype=np.float32)
        depth = nearest_resize_half(depth)[44:556, 80:720]
        return nearest_resize(depth, self.downSample, self.downSample)

    def build_test_metas(self, scene_list_filepath, train_views, test_views, method="nearest"):
        return [(scan, 3, tv, self.sorted_test_src_views(tv, train_views, method))
                for scan in self._scans(scene_list_filepath) for tv in test_views]

    def sorted_test_src_views(self, target_view, train_views, method="nearest"):
        """dtu.py:146-157: source views by L1 distance of the camera centres."""
        if method == "fixed":
            return list(train_views)
        if method != "nearest":
            raise ValueError(f"Unknown evaluate method [{method}]")
        pos = np.stack([self.cam2worlds_dict[x] for x in train_views])[:, :3, 3]
        dist = np.sum(np.abs(pos - self.cam2worlds_dict[target_view][:3, 3]), axis=-1)
        return [train_views[i] for i in np.argsort(dist)]

    def __len__(self):
        return len(self.metas) if self.max_len <= 0 else self.max_len

    def __getitem__(self, idx):
        from PIL import Image
        scan, light, target_view, src_views = self.metas[idx]
        if self.permute_train_src and self.split == "train":
            ids = torch.sort(torch.randperm(self.n_views + self.n_add_train_views)[:self.n_views])[0]
            view_ids = [src_views[int(i)] for i in ids] + [target_view]
        else:
            view_ids = [src_views[i] for i in range(self.n_views)] + [target_view]
        img_wh = np.round(np.array(self.img_wh) * self.downSample).astype("int")
        imgs, depth = [], None
        for vid in view_ids:
            name = os.path.join(self.root_dir, f"Rectified/{scan}_train/rect_{vid + 1:03d}_{light}_r5000.png")  # files count from 1
            img = Image.open(name).resize(tuple(int(x) for x in img_wh), Image.BILINEAR)
            a = np.asarray(img.convert("RGB"), np.uint8)
            imgs.append(torch.from_numpy(a.transpose(2, 0, 1).astype(np.float32) / 255.0))
            if self.split in ("test", "val") and vid == target_view:
                dname = os.path.join(self.root_dir, f"Depths/{scan}/depth_map_{vid:04d}.pfm")
                if not os.path.exists(dname):
                    raise FileNotFoundError(f"{dname}: the target's depth is needed for evaluation")
                depth = self.read_depth(dname) * self.scale_factor
        sample = {
            "images": torch.stack(imgs).float(),
            "extrinsics": np.stack([self.world2cams_dict[v] for v in view_ids]).astype(np.float32),
            "intrinsics": np.stack([self.intrinsics_dict[v] for v in view_ids]).astype(np.float32),
            "near_fars": np.stack([self.near_fars_dict[v] for v in view_ids]).astype(np.float32),
            "view_ids": np.array(view_ids),
            "scene": scan,
            "img_wh": img_wh,
        }
        if depth is not None:
            sample["depth"] = depth.astype(np.float32)
        return sample


from . import scene_sets as _ss  # noqa: E402  (the other five producers; they reach back into this module lazily)

datas_dict = {"dtu": MVSDatasetDTU, "blender": _ss.MVSDatasetBlender, "llff": _ss.MVSDatasetRealFF, "colmap": _ss.MVSDatasetCOLMAP,
              "ibrnet": _ss.MVSDatasetIBRNet, "tnt": _ss.MVSDatasetTNT}  # datasets/__init__.py:9-16
