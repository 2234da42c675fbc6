"""Small seeded on-disk dataset trees in the five layouts `matchnerf_amd/scene_sets.py` reads (LLFF, COLMAP captures, IBRNet
collection, NeRF-synthetic, Tanks-and-Temples) — test INPUT data, written by code so that nothing binary is committed.

`tools/gen_dataset_golden.py` builds the tree in a scratch directory, runs the REFERENCE's dataset classes on it (build container
only) and commits what they return as tests/golden/datasets.npz; tests/test_scene_sets.py builds the same tree again and runs
this repository's producers on it.  Everything is drawn from one PCG64 stream, PNG is lossless, and the JPEG files of the
Tanks-and-Temples layout are decoded by the same libjpeg that wrote them, so both sides see identical pixels."""
import json
import os

import numpy as np
import torch
from PIL import Image
from scipy.spatial.transform import Rotation


def _image(rng, w, h, channels=3):
    img = rng.random((h, w, channels))
    for _ in range(2):
        img = (np.roll(img, 1, 0) + 2 * img + np.roll(img, -1, 0)) / 4
        img = (np.roll(img, 1, 1) + 2 * img + np.roll(img, -1, 1)) / 4
    img = (img - img.min()) / (img.max() - img.min())
    return (img * 255).astype(np.uint8)


def _camera_to_world(rng, n, spread=0.4):
    """n forward-facing cameras in NeRF axes (x right, y up, z back): small random rotations, positions on a jittered grid."""
    out = []
    for i in range(n):
        rot = Rotation.from_euler("xyz", rng.normal(0, 0.12, 3)).as_matrix()
        pos = np.array([(i % 4 - 1.5) * spread, (i // 4 - 1.0) * spread, 0.0]) + rng.normal(0, 0.05, 3)
        out.append(np.concatenate([rot, pos[:, None]], 1))
    return np.stack(out)


def write_poses_bounds_scene(rng, scene_dir, n, wh=(80, 60), prefix="IMG_", ext="png"):
    """`poses_bounds.npy` + images/.  LLFF stores the 3x5 block with columns (down, right, back, position, [h, w, f])."""
    os.makedirs(os.path.join(scene_dir, "images"), exist_ok=True)
    c2w = _camera_to_world(rng, n)
    right, up, back, pos = c2w[..., 0], c2w[..., 1], c2w[..., 2], c2w[..., 3]
    hwf = np.array([wh[1], wh[0], 0.9 * wh[0] + rng.random()])
    block = np.stack([-up, right, back, pos, np.tile(hwf, (n, 1))], -1)  # [n,3,5]
    bounds = np.stack([1.5 + rng.random(n), 6.0 + 3.0 * rng.random(n)], -1)
    np.save(os.path.join(scene_dir, "poses_bounds.npy"), np.concatenate([block.reshape(n, 15), bounds], 1))
    for i in range(n):
        Image.fromarray(_image(rng, *wh)).save(os.path.join(scene_dir, "images", f"{prefix}{i:03d}.{ext}"))


def write_blender_scene(rng, scene_dir, n_train, n_test, side=40):
    for split, n in (("train", n_train), ("test", n_test)):
        os.makedirs(os.path.join(scene_dir, split), exist_ok=True)
        frames = []
        for i in range(n):
            ang = rng.random(3) * np.array([0.6, 2 * np.pi, 0.2])
            rot = Rotation.from_euler("xyz", ang).as_matrix()
            c2w = np.eye(4)
            c2w[:3, :3] = rot
            c2w[:3, 3] = rot[:, 2] * 4.0  # on a sphere of radius 4, looking at the origin (camera looks along -z)
            frames.append(dict(file_path=f"./{split}/r_{i}", rotation=0.012, transform_matrix=c2w.tolist()))
            Image.fromarray(_image(rng, side, side, 4), "RGBA").save(os.path.join(scene_dir, split, f"r_{i}.png"))
        with open(os.path.join(scene_dir, f"transforms_{split}.json"), "w") as f:
            json.dump(dict(camera_angle_x=0.6911112070083618, frames=frames), f)


def write_tnt_scene(rng, scene_dir, view_ids, wh=(96, 54)):
    os.makedirs(os.path.join(scene_dir, "cams_1"), exist_ok=True)
    os.makedirs(os.path.join(scene_dir, "images"), exist_ok=True)
    for j, v in enumerate(view_ids):
        rot = Rotation.from_euler("xyz", rng.normal(0, 0.2, 3)).as_matrix()
        t = rng.normal(0, 0.002, 3)
        k = np.array([[1.1 * wh[0], 0, wh[0] / 2 + rng.normal()], [0, 1.1 * wh[0], wh[1] / 2 + rng.normal()], [0, 0, 1]])
        with open(os.path.join(scene_dir, "cams_1", f"{v:08d}_cam.txt"), "w") as f:
            f.write("extrinsic\n")
            for r in range(3):
                f.write(" ".join(f"{x:.9g}" for x in [*rot[r], t[r]]) + "\n")
            f.write("0.0 0.0 0.0 1.0\n\nintrinsic\n")
            for r in range(3):
                f.write(" ".join(f"{x:.9g}" for x in k[r]) + "\n")
            f.write(f"\n{0.004 + 0.001 * rng.random():.9g} {0.0001:.9g} 192 {0.02 + 0.004 * rng.random():.9g}\n")
        # the first image is larger than the rest: the intrinsics are rescaled per image (tnt.py:160-166)
        size = (wh[0] * 2, wh[1] * 2) if j == 0 else wh
        Image.fromarray(_image(rng, *size)).save(os.path.join(scene_dir, "images", f"{v:08d}.jpg"), quality=95)


def write_dtu_tree(rng, root, scans_train, scans_val, ref_views, n_cams, wh=(80, 64), depth_for=()):
    """MVSNet-style DTU layout (datasets/dtu.py): Cameras/train/%08d_cam.txt, Rectified/<scan>_train/rect_%03d_<light>_r5000.png
    (view ids count from 1 in the file names, 7 lights), Depths/<scan>/depth_map_%04d.pfm at 1200 x 1600 for the (scan, view)
    pairs of `depth_for`, and the list files the reference keeps under configs/dtu_meta/."""
    from matchnerf_amd.datasets import write_pfm
    meta = os.path.join(root, "configs", "dtu_meta")
    os.makedirs(meta, exist_ok=True)
    os.makedirs(os.path.join(root, "dtu", "Cameras", "train"), exist_ok=True)
    with open(os.path.join(meta, "train_all.txt"), "w") as f:
        f.write("".join(s + "\n" for s in scans_train))
    with open(os.path.join(meta, "val_all.txt"), "w") as f:
        f.write("".join(s + "\n" for s in scans_val))
    with open(os.path.join(meta, "view_pairs.txt"), "w") as f:
        f.write(f"{len(ref_views)}\n")
        for ref, srcs in ref_views:
            f.write(f"{ref}\n{len(srcs)} " + " ".join(f"{s} {1000.0 / (j + 1):.3f}" for j, s in enumerate(srcs)) + " \n")
    for v in range(n_cams):
        rot = Rotation.from_euler("xyz", rng.normal(0, 0.25, 3)).as_matrix()
        t = rng.normal(0, 120.0, 3) + np.array([0.0, 0.0, 600.0])
        k = np.array([[361.5 + rng.normal(), 0, 82.9 + rng.normal()], [0, 360.7 + rng.normal(), 66.4 + rng.normal()], [0, 0, 1]])
        with open(os.path.join(root, "dtu", "Cameras", "train", f"{v:08d}_cam.txt"), "w") as f:
            f.write("extrinsic\n")
            for r in range(3):
                f.write(" ".join(f"{x:.9g}" for x in [*rot[r], t[r]]) + " \n")
            f.write("0.0 0.0 0.0 1.0\n\nintrinsic\n")
            for r in range(3):
                f.write(" ".join(f"{x:.9g}" for x in k[r]) + " \n")
            f.write(f"\n{425.0 + rng.random() * 10:.9g} {2.5 + rng.random() * 0.1:.9g}\n")
    for scan in sorted({*scans_train, *scans_val}):
        d = os.path.join(root, "dtu", "Rectified", f"{scan}_train")
        os.makedirs(d, exist_ok=True)
        for v in range(n_cams):
            for light in range(7):
                Image.fromarray(_image(rng, *wh)).save(os.path.join(d, f"rect_{v + 1:03d}_{light}_r5000.png"))
    for scan, v in depth_for:
        d = os.path.join(root, "dtu", "Depths", scan)
        os.makedirs(d, exist_ok=True)
        depth = (600.0 + 200.0 * rng.random((1200, 1600))).astype(np.float32)
        depth[rng.random((1200, 1600)) < 0.3] = 0.0  # holes: the evaluation mask
        write_pfm(os.path.join(d, f"depth_map_{v:04d}.pfm"), depth)


DTU_REF_VIEWS = [(0, [10, 1, 9, 12, 11, 13, 2, 8, 14, 5]), (24, [3, 7, 1, 0, 13, 12, 9, 5, 6, 11]), (6, [5, 7, 13, 0, 1, 2, 3, 4, 8, 9])]

PAIRS = {
    "dtu_train": [3, 7, 1, 0, 13, 12, 9], "dtu_test": [24, 5], "dtu_val": [24, 5],
    "fernlike_train": [7, 2, 9, 0, 4, 11, 5], "fernlike_val": [3, 8], "fernlike_test": [3, 8],
    "roomlike_train": [1, 0, 6, 4, 8], "roomlike_val": [5], "roomlike_test": [5],
    "legolike_train": [6, 3, 1, 8, 0, 9], "legolike_val": [2, 7], "legolike_test": [2, 7],
    "TNT_Yard_train": [1, 2, 17, 71, 73, 74], "TNT_Yard_val": [0, 72], "TNT_Yard_test": [0, 72],
}


def build_trees(root, seed=11):
    """-> root with configs/pairs.th, llff/, colmap/, ibrnet/, blender/, tnt/ underneath."""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "configs"), exist_ok=True)
    torch.save({k: list(v) for k, v in PAIRS.items()}, os.path.join(root, "configs", "pairs.th"))
    write_poses_bounds_scene(rng, os.path.join(root, "llff", "fernlike"), 12)
    write_poses_bounds_scene(rng, os.path.join(root, "llff", "roomlike"), 17, wh=(64, 48), prefix="", ext="png")
    write_poses_bounds_scene(rng, os.path.join(root, "colmap", "desk"), 9, wh=(72, 54))
    write_poses_bounds_scene(rng, os.path.join(root, "colmap", "shelf"), 3, wh=(72, 54))
    write_poses_bounds_scene(rng, os.path.join(root, "ibrnet", "groupA", "s1"), 8, wh=(64, 48))
    write_poses_bounds_scene(rng, os.path.join(root, "ibrnet", "groupB", "s2"), 7, wh=(64, 48))
    write_blender_scene(rng, os.path.join(root, "blender", "legolike"), 10, 3)
    write_tnt_scene(rng, os.path.join(root, "tnt", "Yard"), sorted({*PAIRS["TNT_Yard_train"], *PAIRS["TNT_Yard_val"]}))
    write_tnt_scene(rng, os.path.join(root, "tnt", "Lot"), list(range(10)))  # contiguous ids: the hold-out protocol counts images
    write_dtu_tree(rng, root, ["scanA", "scanB"], ["scanB"], DTU_REF_VIEWS, 25,
                   depth_for=[("scanA", 24), ("scanB", 24), ("scanB", 5)])
    return root


# (case name, registry key, root below the tree, split, constructor keywords) — the same list drives the golden generator
# (reference classes) and the tests (this repository's producers)
CASES = [
    ("dtu_test_nearest", "dtu", "dtu", "test", dict(img_wh=[64, 32], test_views_method="nearest")),
    ("dtu_val", "dtu", "dtu", "val", dict(img_wh=[64, 32])),
    ("dtu_train_first6", "dtu", "dtu", "train", dict(img_wh=[32, 32], max_len=6, n_add_train_views=2)),
    ("llff_nearest", "llff", "llff", "test", dict(img_wh=[48, 32], test_views_method="nearest", eval_mode="mvsnerf")),
    ("llff_fixed_one_scene", "llff", "llff", "test", dict(img_wh=[40, 32], scene_list=["fernlike"], test_views_method="fixed")),
    ("llff_gpnr", "llff", "llff", "test", dict(img_wh=[48, 32], scene_list=["roomlike"], eval_mode="gpnr")),
    ("colmap_nearest_avg", "colmap", "colmap", "test", dict(img_wh=[48, 40], test_views_method="nearest", nf_mode="avg")),
    ("colmap_fixed_minmax", "colmap", "colmap", "test", dict(img_wh=[48, 40], test_views_method="fixed", nf_mode="minmax")),
    ("ibrnet_val", "ibrnet", "ibrnet", "val", dict(img_wh=[48, 32])),
    ("ibrnet_train", "ibrnet", "ibrnet", "train", dict(img_wh=[48, 32])),
    ("blender_mvsnerf", "blender", "blender", "test", dict(img_wh=[32, 32], eval_mode="mvsnerf")),
    ("blender_gpnr_fixed", "blender", "blender", "test", dict(img_wh=[64, 32], eval_mode="gpnr", test_views_method="fixed")),
    ("tnt_avg", "tnt", "tnt", "test", dict(img_wh=[48, 32], scene_list=["Yard"], nf_mode="avg")),
    ("tnt_gpnr_minmax", "tnt", "tnt", "test", dict(img_wh=[48, 32], scene_list=["Lot"], eval_mode="gpnr", nf_mode="minmax",
                                                   test_views_method="fixed")),
]
FIELDS = ("images", "extrinsics", "intrinsics", "near_fars", "view_ids", "img_wh", "c2ws_all", "depth")
