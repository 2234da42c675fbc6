"""f4: DTU on-disk formats and the SSIM restatement, on hand-built files with known answers (CPU)."""
import os

import numpy as np
import pytest
import torch

from matchnerf_amd import datasets, metrics


def _write_cam(path, extr, intr, dmin, dint):
    with open(path, "w") as f:
        f.write("extrinsic\n")
        for r in extr:
            f.write(" ".join(f"{x:.6f}" for x in r) + "\n")
        f.write("\nintrinsic\n")
        for r in intr:
            f.write(" ".join(f"{x:.6f}" for x in r) + "\n")
        f.write(f"\n{dmin} {dint}\n")


def test_pfm_round_trip_and_orientation(tmp_path):
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    p = str(tmp_path / "d.pfm")
    datasets.write_pfm(p, a, scale=2.0)
    with open(p, "rb") as f:
        assert f.readline() == b"Pf\n" and f.readline() == b"4 3\n"
        f.readline()
        first_row_in_file = np.frombuffer(f.read(16), "<f4")
    assert np.array_equal(first_row_in_file, a[-1])          # the file stores the BOTTOM row first
    b, scale = datasets.read_pfm(p)
    assert scale == 2.0 and np.array_equal(b, a)
    big = np.array(a, dtype=">f4")                           # a big-endian file (positive scale)
    with open(p, "wb") as f:
        f.write(b"Pf\n4 3\n1.0\n")
        np.flipud(big).tofile(f)
    assert np.array_equal(datasets.read_pfm(p)[0], a)
    with open(p, "wb") as f:
        f.write(b"P6\n4 3\n1.0\n")
    with pytest.raises(ValueError, match="not a PFM"):
        datasets.read_pfm(p)


def test_cam_file_fields(tmp_path):
    extr = np.eye(4)
    extr[:3, 3] = [100.0, -200.0, 400.0]
    intr = np.array([[361.5, 0, 82.9], [0, 360.4, 66.4], [0, 0, 1]])
    p = str(tmp_path / "00000007_cam.txt")
    _write_cam(p, extr, intr, 425.0, 2.5)
    k, e, nf = datasets.read_cam_file(p)
    assert k.dtype == np.float32 and np.allclose(k, intr) and np.allclose(e, extr)
    assert nf == pytest.approx([425.0 / 200, 425.0 / 200 + 2.5 * 192 / 200])


def test_nearest_resizes_pick_the_opencv_samples():
    a = np.arange(36).reshape(6, 6)
    assert np.array_equal(datasets.nearest_resize_half(a), a[::2, ::2])
    assert np.array_equal(datasets.nearest_resize(a, 0.5, 0.5), a[::2, ::2])
    assert np.array_equal(datasets.nearest_resize(a, 1.0, 1.0), a)


def _make_dtu(root, tmp, n_cams=6):
    from PIL import Image
    os.makedirs(root / "Cameras" / "train")
    rng = np.random.default_rng(0)
    for v in range(n_cams):
        extr = np.eye(4)
        extr[:3, 3] = [-100.0 * v, 0.0, 0.0]                  # camera centre at x = 100 v mm
        _write_cam(str(root / "Cameras" / "train" / f"{v:08d}_cam.txt"), extr, [[100, 0, 20], [0, 100, 16], [0, 0, 1]], 400.0, 2.0)
    os.makedirs(root / "Rectified" / "scan1_train")
    for v in range(n_cams):
        img = (rng.random((48, 64, 3)) * 255).astype(np.uint8)
        Image.fromarray(img).save(str(root / "Rectified" / "scan1_train" / f"rect_{v + 1:03d}_3_r5000.png"))
    os.makedirs(root / "Depths" / "scan1")
    depth = np.full((1200, 1600), 600.0, np.float32)
    depth[:100] = 0.0
    for v in range(n_cams):
        datasets.write_pfm(str(root / "Depths" / "scan1" / f"depth_map_{v:04d}.pfm"), depth)
    meta = tmp / "meta"
    os.makedirs(meta)
    (meta / "val_all.txt").write_text("scan1\n")
    (meta / "train_all.txt").write_text("scan1\n")
    (meta / "view_pairs.txt").write_text("2\n0\n3 1 9.0 2 8.0 3 7.0\n1\n3 0 9.0 2 8.0 3 7.0\n")
    torch.save({"dtu_train": [0, 1, 3, 4, 5], "dtu_test": [2], "dtu_val": [2]}, str(tmp / "pairs.th"))
    return str(meta), str(tmp / "pairs.th")


def test_dtu_test_split_sample(tmp_path):
    root = tmp_path / "dtu"
    meta, pairs = _make_dtu(root, tmp_path)
    ds = datasets.MVSDatasetDTU(str(root), "test", n_views=3, img_wh=(64, 32), meta_dir=meta, pairs_file=pairs)
    assert len(ds) == 1 and ds.get_name() == "dtu"
    scan, light, target, src = ds.metas[0]
    assert (scan, light, target) == ("scan1", 3, 2)
    assert src == [1, 3, 0, 4, 5]                              # L1 distance of the camera centres to view 2 (ties: argsort order)
    s = ds[0]
    assert s["images"].shape == (4, 3, 32, 64) and s["images"].dtype == torch.float32
    assert 0.0 <= float(s["images"].min()) and float(s["images"].max()) <= 1.0
    assert list(s["view_ids"]) == [1, 3, 0, 2]                # target last
    assert s["extrinsics"].shape == (4, 4, 4) and s["intrinsics"].shape == (4, 3, 3) and s["near_fars"].shape == (4, 2)
    assert np.allclose(s["intrinsics"][0], [[400, 0, 80], [0, 400, 64], [0, 0, 1]])        # x4: files are at 1/4 resolution
    assert np.allclose(s["extrinsics"][0][:3, 3], [-100.0 / 200, 0, 0])                   # mm -> model units
    assert np.allclose(s["near_fars"][0], [2.0, 2.0 + 2.0 * 192 / 200])
    assert s["depth"].shape == (512, 640) and s["depth"].dtype == np.float32
    assert np.allclose(s["depth"][10], 3.0) and np.all(s["depth"][:6] == 0.0)             # rows 88..99 of the half map are invalid
    assert tuple(s["img_wh"]) == (64, 32) and s["scene"] == "scan1"


def test_dtu_train_and_val_metas(tmp_path):
    root = tmp_path / "dtu"
    meta, pairs = _make_dtu(root, tmp_path)
    tr = datasets.MVSDatasetDTU(str(root), "train", n_views=2, img_wh=(64, 32), n_add_train_views=1, meta_dir=meta, pairs_file=pairs)
    assert len(tr) == 2 * 7 and tr.metas[0] == ("scan1", 0, 0, [1, 2, 3]) and tr.metas[7][2] == 1
    va = datasets.MVSDatasetDTU(str(root), "val", n_views=2, img_wh=(64, 32), meta_dir=meta, pairs_file=pairs)
    assert len(va.metas) == 0                                 # (only reference view 24 at light 3 is a validation sample)
    with pytest.raises(ValueError):
        datasets.MVSDatasetDTU(str(root), "test", img_wh=(60, 32), meta_dir=meta, pairs_file=pairs)


def _ssim_direct(x, y, data_range=2.0, win=7):
    """the published definition, window by window"""
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    vals = []
    for c in range(x.shape[-1]):
        for i in range(x.shape[0] - win + 1):
            for j in range(x.shape[1] - win + 1):
                a, b = x[i:i + win, j:j + win, c].astype(np.float64).ravel(), y[i:i + win, j:j + win, c].astype(np.float64).ravel()
                ma, mb = a.mean(), b.mean()
                va, vb, cab = a.var(ddof=1), b.var(ddof=1), np.cov(a, b, ddof=1)[0, 1]
                vals.append((2 * ma * mb + c1) * (2 * cab + c2) / ((ma ** 2 + mb ** 2 + c1) * (va + vb + c2)))
    return float(np.mean(vals))


def test_ssim_matches_direct_definition():
    rng = np.random.default_rng(3)
    x = rng.random((20, 24, 3))
    y = np.clip(x + 0.1 * rng.standard_normal(x.shape), 0, 1)
    assert metrics.ssim(x, x) == pytest.approx(1.0, abs=1e-12)
    assert metrics.ssim(x, y) == pytest.approx(_ssim_direct(x, y), abs=1e-10)
    assert metrics.ssim(x.astype(np.float32), y.astype(np.float32)) == pytest.approx(_ssim_direct(x, y), abs=2e-5)
    assert metrics.ssim(x, y, data_range=1.0) < metrics.ssim(x, y)   # the float default (2) is the more forgiving one
    with pytest.raises(ValueError):
        metrics.ssim(x[:5], y[:5])


def test_ssim_matches_the_hand_derived_golden_vectors():
    """tests/golden/ssim_hand_derived.json (tools/gen_ssim_golden.py): images whose 7x7 window statistics are known in closed form —
    constant images, opposite stripes, stripes at half contrast — with the SSIM value derived on paper and evaluated in exact
    rationals.  Independent of any filter implementation, this one's and scikit-image's alike."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ssim_hand_derived.json")))
    assert len(g["cases"]) == 3
    for c in g["cases"]:
        h, w, ch = c["shape"]
        s = np.where(np.arange(w) % 2 == 0, 1.0, -1.0)[None, :, None]
        if c["kind"] == "constant":
            x, y = np.full((h, w, ch), c["a"]), np.full((h, w, ch), c["b"])
        elif c["kind"] == "stripes":
            x, y = np.broadcast_to(c["m"] + c["amp"] * s, (h, w, ch)), np.broadcast_to(c["m"] - c["amp"] * s, (h, w, ch))
        else:
            x, y = np.broadcast_to(c["m"] + c["amp"] * s, (h, w, ch)), np.broadcast_to(c["m"] + 0.5 * c["amp"] * s, (h, w, ch))
        assert metrics.ssim(np.array(x), np.array(y)) == pytest.approx(c["ssim"], abs=1e-12), c["name"]
        assert metrics.ssim(np.array(x, np.float32), np.array(y, np.float32)) == pytest.approx(c["ssim"], abs=5e-6), c["name"]
        assert c["data_range"] == 2.0  # what a float image gets when the caller passes none (misc/metrics.py:43-45 passes none)


def test_eval_tools_mask_and_crop():
    rng = np.random.default_rng(4)
    gt = rng.random((40, 50, 3)).astype(np.float32)
    pred = np.clip(gt + 0.05 * rng.standard_normal(gt.shape).astype(np.float32), 0, 1)
    mask = np.zeros((40, 50), bool)
    mask[:10] = True
    tools = metrics.EvalTools()
    tools.set_inputs(pred, gt, mask)
    m = tools.get_metrics(return_full=True)
    assert list(m) == ["PSNR", "PSNR_Full", "SSIM", "SSIM_Full"]
    assert m["PSNR"] == pytest.approx(metrics.psnr(pred, gt, mask), abs=1e-4)
    zp, zg = pred.copy(), gt.copy()
    zp[mask] = 0
    zg[mask] = 0
    assert m["SSIM"] == pytest.approx(metrics.ssim(zp, zg))
    tools.set_inputs(pred, gt)
    m = tools.get_metrics(["PSNR"])
    assert m["PSNR"] == pytest.approx(metrics.psnr(pred, gt), abs=1e-4)
    with pytest.raises(ValueError, match="only support"):
        tools.get_metrics(["LPIPS"])
    assert metrics.EvalTools(lpips_fn=lambda a, b: 0.25).get_metrics.__self__.support_metrics[-1] == "LPIPS"


def test_missing_list_files_are_named_and_coach_passes_the_overrides(tmp_path, monkeypatch):
    """with data on disk but without the reference's list files the producer used to die on a bare
    FileNotFoundError for configs/pairs.th.  Now the error names the files, every place tried and the overrides, the lists are
    also found under $MNERF_META_ROOT, and Coach.load_dataset hands data_test.<name>.meta_dir / pairs_file through."""
    import pytest
    from matchnerf_amd import coach, options
    root = tmp_path / "dtu"
    meta, pairs = _make_dtu(root, tmp_path)
    monkeypatch.chdir(tmp_path)  # no configs/ here
    monkeypatch.delenv("MNERF_META_ROOT", raising=False)
    with pytest.raises(FileNotFoundError, match=r"pairs\.th.*meta_dir= / pairs_file="):
        datasets.MVSDatasetDTU(str(root), "test", n_views=3, img_wh=(64, 32))
    with pytest.raises(FileNotFoundError, match="DTU scan list"):
        datasets.MVSDatasetDTU(str(root), "train", n_views=3, img_wh=(64, 32))
    # the reference's layout under a checkout root
    ref = tmp_path / "checkout"
    os.makedirs(ref / "configs" / "dtu_meta")
    for n in ("val_all.txt", "train_all.txt", "view_pairs.txt"):
        (ref / "configs" / "dtu_meta" / n).write_text(open(os.path.join(meta, n)).read())
    torch.save(torch.load(pairs, weights_only=False), str(ref / "configs" / "pairs.th"))
    monkeypatch.setenv("MNERF_META_ROOT", str(ref))
    assert len(datasets.MVSDatasetDTU(str(root), "test", n_views=3, img_wh=(64, 32))) == 1
    monkeypatch.delenv("MNERF_META_ROOT")
    # Coach.load_dataset with an existing root_dir: the yaml entry carries the locations
    opt = options.load_options("configs/test.yaml", verbose=False)
    opt.device = "cpu"
    for name in list(opt.data_test.keys()):
        if name != "dtu":
            opt.data_test[name] = None
    opt.data_test.dtu.root_dir = str(root)
    opt.data_test.dtu.img_wh = [64, 32]
    opt.data_test.dtu.meta_dir, opt.data_test.dtu.pairs_file = meta, pairs
    c = coach.Coach.__new__(coach.Coach)  # load_dataset only needs the options and the view count
    c.opts, c.n_src_views = opt, 3
    c.load_dataset()
    assert len(c.test_loaders) == 1 and c.test_loaders[0].get_name() == "dtu"
    batch = next(iter(c.test_loaders[0]))
    assert batch["images"].shape == (1, 4, 3, 32, 64)


def test_lpips_network_structure_and_weight_files(tmp_path, monkeypatch):
    """metrics.LPIPSVGG (PARITY UNPINNED: no lpips weights offline): files with torchvision's / lpips's key names and shapes load
    strictly, the forward equals an independent functional evaluation of the published network, d(x, x) = 0, and a missing file
    is a clear error."""
    import torch
    import torch.nn.functional as F
    from matchnerf_amd import metrics as M
    g = torch.Generator().manual_seed(0)
    vgg = {}
    for i, (ci, co) in M.LPIPS_VGG_CONVS.items():
        vgg[f"features.{i}.weight"] = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
        vgg[f"features.{i}.bias"] = torch.randn(co, generator=g) * 0.01
    vgg["classifier.0.weight"] = torch.zeros(2, 2)  # the torchvision file also holds the classifier: ignored
    lin = {f"lin{l}.model.1.weight": torch.rand(1, c, 1, 1, generator=g) for l, c in enumerate(M.LPIPS_CHANNELS)}
    torch.save(vgg, tmp_path / "vgg16.pth")
    torch.save(lin, tmp_path / "lin.pth")
    fn = M.load_lpips(str(tmp_path / "vgg16.pth"), str(tmp_path / "lin.pth"))
    rng = np.random.default_rng(1)
    a, b = rng.random((40, 48, 3), dtype=np.float32), rng.random((40, 48, 3), dtype=np.float32)
    assert fn(a, a) == 0.0 and fn(a, b) > 0.0 and abs(fn(a, b) - fn(b, a)) < 1e-6

    def direct(x, y):  # independent evaluation: a plain layer list, features at the published taps
        taps, feats = {3, 8, 15, 22, 29}, ([], [])
        shift, scale = torch.tensor([-0.030, -0.088, -0.188]).view(1, 3, 1, 1), torch.tensor([0.458, 0.448, 0.450]).view(1, 3, 1, 1)
        for k, t in enumerate((x, y)):
            t = (t - shift) / scale
            for i in range(30):
                if f"features.{i}.weight" in vgg:
                    t = F.conv2d(t, vgg[f"features.{i}.weight"], vgg[f"features.{i}.bias"], padding=1)
                elif i in (4, 9, 16, 23):
                    t = F.max_pool2d(t, 2)
                else:
                    t = F.relu(t)
                if i in taps:
                    feats[k].append(t)
        d = 0.0
        for l, (fa, fb) in enumerate(zip(*feats)):
            na = fa / (fa.norm(dim=1, keepdim=True) + 1e-10)
            nb = fb / (fb.norm(dim=1, keepdim=True) + 1e-10)
            d += float((((na - nb) ** 2) * lin[f"lin{l}.model.1.weight"]).sum(1).mean())
        return d

    to_t = lambda z: torch.from_numpy(z)[None].permute(0, 3, 1, 2) * 2 - 1
    assert abs(fn(a, b) - direct(to_t(a), to_t(b))) < 1e-5 * max(1.0, fn(a, b))
    tools = M.EvalTools(lpips_fn=fn)
    tools.set_inputs(a, b)
    assert set(tools.get_metrics()) == {"PSNR", "SSIM", "LPIPS"}
    monkeypatch.delenv("MNERF_LPIPS_VGG16", raising=False)
    with pytest.raises(FileNotFoundError, match="no network"):
        M.load_lpips(str(tmp_path / "absent.pth"), str(tmp_path / "lin.pth"))
