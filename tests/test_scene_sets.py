"""The five producers of matchnerf_amd/scene_sets.py against what the REFERENCE's dataset classes returned on the same files
(tests/golden/datasets.npz, written by tools/gen_dataset_golden.py from /root/reference/datasets/*.py): camera quantities bit
for bit, pixels exactly; plus the reference's real COLMAP scene (tests/golden/demo_data/printer) and the coach wiring."""
import hashlib
import os

import numpy as np
import pytest
import torch

import dataset_trees as DT
from conftest import GOLDEN
from matchnerf_amd import datasets, scene_sets


@pytest.fixture(scope="module")
def trees(tmp_path_factory):
    return DT.build_trees(str(tmp_path_factory.mktemp("trees")))


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(GOLDEN, "datasets.npz")))


def check_case(gold, case, ds, seed=None):
    assert len(ds) == int(gold[f"{case}/len"]), case
    for i in range(len(ds)):
        if seed is not None:
            torch.manual_seed(seed + i)
        s = ds[i]
        assert str(s["scene"]) == bytes(gold[f"{case}/{i}/scene"]).decode()
        for k in DT.FIELDS:
            key = f"{case}/{i}/{k}"
            if k == "depth":
                if f"{case}/{i}/depth_sha256" in gold:
                    got = np.ascontiguousarray(np.asarray(s[k]))
                    assert got.dtype == np.float32 and np.array_equal(got[::8, ::8], gold[f"{case}/{i}/depth_sub8"])
                    assert hashlib.sha256(got.tobytes()).digest() == bytes(gold[f"{case}/{i}/depth_sha256"]), (case, i)
                else:
                    assert k not in s
                continue
            if k == "images":
                got = np.ascontiguousarray(np.asarray(s[k]))
                assert got.dtype == np.float32
                assert hashlib.sha256(got.tobytes()).digest() == bytes(gold[f"{case}/{i}/images_sha256"]), (case, i)
            if key not in gold:
                assert k == "images" or k not in s, (case, i, k)
                continue
            got, want = np.asarray(s[k]), gold[key]
            assert got.dtype == want.dtype and got.shape == want.shape, (case, i, k, got.dtype, want.dtype)
            assert np.array_equal(got, want), (case, i, k, np.abs(got.astype(np.float64) - want).max())


@pytest.mark.parametrize("case", [c[0] for c in DT.CASES])
def test_producer_equals_the_reference_class(trees, gold, case, monkeypatch):
    _, kind, sub, split, kw = next(c for c in DT.CASES if c[0] == case)
    monkeypatch.chdir(trees)  # 'configs/pairs.th' is looked up relative to the working directory first, like the reference
    ds = datasets.datas_dict[kind](os.path.join(trees, sub), split, n_views=3, **kw)
    check_case(gold, case, ds, seed=5 if split == "train" else None)


@pytest.mark.parametrize("case,wh", [("printer_256x160", [256, 160]), ("printer_96x64", [96, 64])])
def test_real_colmap_scene_of_the_reference(gold, case, wh):
    """docs/demo_data/printer with the dataset options of configs/demo_own.yaml:24-37."""
    ds = scene_sets.MVSDatasetCOLMAP(os.path.join(GOLDEN, "demo_data"), "test", n_views=3, img_wh=wh, scene_list=["printer"],
                                     test_views_method="fixed", nf_mode="minmax")
    check_case(gold, case, ds)
    s = ds[0]
    assert list(s["view_ids"]) == [2, 1, 0, 0]
    k = s["intrinsics"][0]
    assert k[0, 0] != k[1, 1] and k[0, 2] == wh[0] / 2  # 4:3 photographs squeezed to 8:5: fx != fy


def test_producer_batch_is_the_demo_own_golden_batch(golden):
    """The batch the reference MODEL was run on for tests/golden/demo_own*.npz is what this producer yields."""
    for name, wh in (("demo_own_small", [96, 64]),):
        g = golden(name)
        s = scene_sets.MVSDatasetCOLMAP(os.path.join(GOLDEN, "demo_data"), "test", n_views=3, img_wh=wh, scene_list=["printer"],
                                        test_views_method="fixed", nf_mode="minmax")[0]
        for k in ("extrinsics", "intrinsics", "near_fars", "c2ws_all", "images"):
            assert np.array_equal(np.asarray(s[k]), g[k][0]), k


def test_colmap_view_split_small_and_large(trees):
    pairs = scene_sets.colmap_view_split(os.path.join(trees, "colmap"))
    assert pairs["shelf_train"] == [2, 1, 0] and pairs["shelf_val"] == [0]
    assert len(pairs["desk_val"]) == 2 and len(pairs["desk_train"]) == 7
    assert sorted(pairs["desk_val"] + pairs["desk_train"]) == list(range(9))
    os.makedirs(os.path.join(trees, "colmap_bad", "nopose"))
    with pytest.raises(FileNotFoundError, match="COLMAP"):
        scene_sets.colmap_view_split(os.path.join(trees, "colmap_bad"))


def test_argument_errors(trees, monkeypatch):
    monkeypatch.chdir(trees)
    with pytest.raises(ValueError):
        scene_sets.MVSDatasetRealFF(os.path.join(trees, "llff"), "train", img_wh=[48, 32])
    with pytest.raises(ValueError, match="multiples of 32"):
        scene_sets.MVSDatasetBlender(os.path.join(trees, "blender"), "test", img_wh=[40, 32])
    with pytest.raises(ValueError):
        scene_sets.MVSDatasetIBRNet(os.path.join(trees, "ibrnet"), "test", img_wh=[48, 32])
    ds = scene_sets.MVSDatasetRealFF(os.path.join(trees, "llff"), "test", img_wh=[48, 32], test_views_method="nearest")
    with pytest.raises(ValueError, match="evaluate method"):
        ds.order_sources("fernlike", 3, [1, 2], "farthest")
    monkeypatch.chdir(os.path.join(trees, "llff"))  # no configs/pairs.th here
    with pytest.raises(FileNotFoundError, match="pairs.th"):
        scene_sets.MVSDatasetRealFF(os.path.join(trees, "llff"), "test", img_wh=[48, 32])


def test_max_len_and_loader_collation(trees, monkeypatch):
    monkeypatch.chdir(trees)
    ds = scene_sets.MVSDatasetTNT(os.path.join(trees, "tnt"), "test", img_wh=[48, 32], scene_list=["Yard"], max_len=1)
    assert len(ds) == 1
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=1)))
    assert batch["images"].shape == (1, 4, 3, 32, 48) and batch["extrinsics"].dtype == torch.float32
    assert batch["c2ws_all"].shape == (1, 6, 4, 4)
