"""Golden vectors for the dataset producers from the IMPORTED reference — BUILD CONTAINER ONLY.

    python tools/gen_dataset_golden.py        # writes tests/golden/datasets.npz and tests/golden/demo_data/printer/

Runs the reference's own dataset classes (/root/reference/datasets/{llff,colmap,ibrnet,blender,tnt}.py, imported in place with
the torchvision stand-in of tools/ref_import.py) on
  * the seeded on-disk trees of tests/dataset_trees.py (test input written by code), every case of its CASES list — the DTU
    cases with a stand-in for the one OpenCV call of that loader (cv2_nearest_resize below) —, and
  * the one real scene the reference ships, docs/demo_data/printer, with the dataset options of configs/demo_own.yaml —
    its three photographs and poses_bounds.npy are DATA and are copied next to the goldens so that the producer test and
    `python test.py --yaml=demo_own --data_test.colmap.root_dir=tests/golden/demo_data` run wherever the repository is,
and stores what each `__getitem__` returns (all camera quantities for every sample; pixel data for the first samples of a case,
a SHA-256 of the pixel bytes for the rest).  Only data is written; no reference source is copied."""
import hashlib
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))

from ref_import import import_reference, REF_ROOT  # noqa: E402
import dataset_trees as DT  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
FULL_SAMPLES = 2  # per case: samples whose pixels are stored in full


def record(store, case, ds, seed=None):
    store[f"{case}/len"] = np.array(len(ds))
    for i in range(len(ds)):
        if seed is not None:
            torch.manual_seed(seed + i)  # the IBRNet training split draws its sources from torch's generator
        s = ds[i]
        for k in DT.FIELDS:
            if k not in s:
                continue
            v = np.asarray(s[k])
            if k == "depth":  # 512 x 640 floats per sample: a digest of all of it + every 8th pixel
                store[f"{case}/{i}/depth_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest(), np.uint8)
                store[f"{case}/{i}/depth_sub8"] = v[::8, ::8].copy()
                continue
            if k == "images":
                store[f"{case}/{i}/images_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest(), np.uint8)
                if i >= FULL_SAMPLES:
                    continue
            store[f"{case}/{i}/{k}"] = v
        store[f"{case}/{i}/scene"] = np.frombuffer(str(s["scene"]).encode(), np.uint8)


def cv2_nearest_resize(src, dsize, fx=None, fy=None, interpolation=None):
    """Stand-in for the ONE OpenCV call of the reference's DTU loader (datasets/dtu.py:126-129, `cv2.resize(depth, None, fx, fy,
    INTER_NEAREST)`), which this image lacks: OpenCV's nearest rule (destination size round(size * f), source index
    floor(dst / f) clamped) as matchnerf_amd.datasets.nearest_resize states it.  So the DTU goldens pin everything of that loader
    EXCEPT this rule itself (camera files, list files, view selection, image decoding / resizing, the depth crop and scale)."""
    from matchnerf_amd.datasets import nearest_resize
    assert dsize is None and interpolation == 0
    return nearest_resize(src, fx, fy)


def main():
    import_reference()
    sys.modules["cv2"].resize = cv2_nearest_resize
    sys.modules["cv2"].INTER_NEAREST = 0
    import datasets as ref_datasets  # the reference's package (cwd and sys.path point into /root/reference)
    store = {}
    tmp = tempfile.mkdtemp(prefix="mnerf_trees_")
    try:
        DT.build_trees(tmp)
        os.chdir(tmp)  # the reference opens 'configs/pairs.th' relative to the working directory
        for case, kind, sub, split, kw in DT.CASES:
            ds = ref_datasets.datas_dict[kind](os.path.join(tmp, sub), split, n_views=3, **kw)
            record(store, case, ds, seed=5 if split == "train" else None)
            print(f"[dataset golden] {case}: {len(ds)} samples")
    finally:
        os.chdir(REF_ROOT)
        shutil.rmtree(tmp, ignore_errors=True)
    # the reference's real scene
    dst = os.path.join(OUT, "demo_data", "printer")
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    shutil.copytree(os.path.join(REF_ROOT, "docs", "demo_data", "printer"), dst)
    for root, _, files in os.walk(dst):
        os.chmod(root, 0o755)
        for f in files:
            os.chmod(os.path.join(root, f), 0o644)
    for case, wh in (("printer_256x160", [256, 160]), ("printer_96x64", [96, 64])):
        ds = ref_datasets.datas_dict["colmap"]("docs/demo_data", "test", n_views=3, img_wh=wh, max_len=-1, scene_list=["printer"],
                                              test_views_method="fixed", nf_mode="minmax")
        record(store, case, ds)
        print(f"[dataset golden] {case}: {len(ds)} samples")
    path = os.path.join(OUT, "datasets.npz")
    np.savez_compressed(path, **store)
    print(f"[dataset golden] -> {path} {os.path.getsize(path) / 1e6:.2f} MB, {len(store)} arrays")


if __name__ == "__main__":
    main()
