"""CPU-only checks of the host side: state_dict layout, registry, video paths, ABI exports."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO
from matchnerf_amd import hip, options, synthetic as syn, video
from matchnerf_amd.edict import EasyDict


def _opts(**over):
    opt = options.load_options("configs/test.yaml", verbose=False)
    opt.device = "cpu"
    for k, v in over.items():
        node = opt
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return opt


def test_state_dict_keys_and_shapes_match_reference_layout():
    """153 tensors, same names / order / shapes as the reference module
    (the spec itself is asserted equal to the reference's state_dict in tools/gen_golden.py)."""
    from matchnerf_amd.models import models_dict
    for v in (3, 4):
        model = models_dict["matchnerf"](_opts(n_src_views=v))
        sd = model.state_dict()
        spec = syn.state_dict_spec(n_src_views=v)
        assert list(sd.keys()) == list(spec.keys())
        for k, t in sd.items():
            assert tuple(t.shape) == spec[k], k
        assert [n for n, _ in model.named_children()] == ["feat_enc", "nerf_dec"]
        assert [n for n, _ in model.feat_enc.named_children()] == ["backbone", "transformer", "featup_net"]
        model.load_state_dict(syn.to_torch(syn.seeded_state_dict(spec, 1)), strict=True)


def test_checkpoint_roundtrip_per_child(tmp_path):
    """misc/utils.py:183-222 semantics: the caller hands over {model, optim} only (coach.py:290-300), epoch / iter
    are stamped by save_checkpoint; restore is per top-level child, strict; resume returns (epoch, iter)."""
    from matchnerf_amd import checkpoint
    from matchnerf_amd.models import models_dict
    m1 = models_dict["matchnerf"](_opts())
    m1.load_state_dict(syn.to_torch(syn.seeded_state_dict(syn.state_dict_spec(), 5)))
    opt1 = torch.optim.SGD(m1.parameters(), lr=0.25)
    path = checkpoint.save_checkpoint(str(tmp_path), dict(model=m1.state_dict(), optim=opt1.state_dict()), ep=3, it=7)
    slim = torch.load(tmp_path / "models" / "ep3_it7.pth")
    assert slim["epoch"] == 3 and slim["iter"] == 7 and "optim" not in slim
    m2 = models_dict["matchnerf"](_opts())
    ep, it = checkpoint.restore_checkpoint(m2, path, "cpu")
    assert (ep, it) == (None, None)
    for (k1, a), (k2, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(a, b)
    opt2 = torch.optim.SGD(m2.parameters(), lr=1.0)
    assert checkpoint.restore_checkpoint(m2, path, "cpu", resume=True, optims_scheds=dict(optim=opt2)) == (3, 7)
    assert opt2.param_groups[0]["lr"] == 0.25
    # children filter (misc/utils.py:211-212): decoder-only checkpoint leaves the encoder untouched
    path = checkpoint.save_checkpoint(str(tmp_path / "dec"), dict(model=m1.state_dict()), 1, 2, children="nerf_dec")
    assert all(k.startswith("nerf_dec.") for k in torch.load(path)["model"])
    m3 = models_dict["matchnerf"](_opts())
    enc_before = {k: v.clone() for k, v in m3.feat_enc.state_dict().items()}
    checkpoint.restore_checkpoint(m3, path, "cpu")
    assert all(torch.equal(v, enc_before[k]) for k, v in m3.feat_enc.state_dict().items())
    assert torch.equal(m3.nerf_dec.pts_bias.weight, m1.nerf_dec.pts_bias.weight)


def test_unsupported_architectures_fail_loudly():
    from matchnerf_amd.models import models_dict
    with pytest.raises(NotImplementedError):
        models_dict["matchnerf"](_opts(**{"decoder.net_width": 256}))
    with pytest.raises(NotImplementedError):
        models_dict["matchnerf"](_opts(**{"decoder.posenc.L_view": 4}))


def test_decoder_packing_follows_the_math_switch(monkeypatch):
    """CondNeRF.packed: split-fp16 stream by default (every S), split-bf16 / exact-f32 streams on request; the cache is
    keyed on the switch and on S; bad values are rejected."""
    from matchnerf_amd import cond_nerf as CN
    from matchnerf_amd.models import models_dict
    dec = models_dict["matchnerf"](_opts()).nerf_dec
    monkeypatch.delenv("MNERF_DECODER_MATH", raising=False)
    ws, small, cs, fmt = dec.packed(64, "cpu")
    assert fmt == 2 and ws.numel() == CN.decoder_schedule_h(dec.cond_dim, dec.L_3D)[1]
    assert dec.packed(64, "cpu")[0] is ws                       # cached
    ws256, _, _, fmt256 = dec.packed(256, "cpu")
    assert fmt256 == 2 and ws256.numel() == ws.numel() and ws256 is not ws
    monkeypatch.setenv("MNERF_DECODER_MATH", "bf16x6")
    ws16, _, _, fmt16 = dec.packed(64, "cpu")
    assert fmt16 == 1 and ws16.numel() == CN.decoder_schedule16(dec.cond_dim, dec.L_3D)[1]
    monkeypatch.setenv("MNERF_DECODER_MATH", "f32")
    ws32, _, _, fmt32 = dec.packed(64, "cpu")
    assert fmt32 == 0 and ws32.numel() == CN.decoder_schedule(cs, dec.L_3D)[1]
    monkeypatch.setenv("MNERF_DECODER_MATH", "fp8")
    with pytest.raises(ValueError, match="MNERF_DECODER_MATH"):
        dec.packed(64, "cpu")


def test_view_limit_is_the_kernels(monkeypatch):
    """MNERF_MAX_VIEWS = 16 is reachable with the split streams (cond_stride 88 <= 96); the exact-f32 stream keeps
    its 64-float limit and says so when it is asked for more."""
    from matchnerf_amd.models import models_dict
    monkeypatch.delenv("MNERF_DECODER_MATH", raising=False)
    m16 = models_dict["matchnerf"](_opts(n_src_views=16))
    assert m16.nerf_dec.packed(64, "cpu")[2] == 80
    monkeypatch.setenv("MNERF_DECODER_MATH", "f32")
    with pytest.raises(NotImplementedError, match="supports 64"):
        m16.nerf_dec.packed(64, "cpu")
    with pytest.raises(NotImplementedError):
        models_dict["matchnerf"](_opts(n_src_views=17))


def test_data_parallel_wrapping_is_transparent():
    """coach.py:83-85 wraps feat_enc / nerf_dec in nn.DataParallel when len(gpu_ids) > 1: the module keeps
    finding its decoder (packing, cond_dim) through the wrapper."""
    from matchnerf_amd.models import models_dict
    model = models_dict["matchnerf"](_opts())
    plain = model._dec()
    model.nerf_dec = torch.nn.DataParallel(model.nerf_dec, [0, 1])
    assert model._dec() is plain and model._dec().cond_dim == 22
    d = model._decoder(64, torch.device("cpu"))
    assert d.cond_dim == 22 and d.wstream_format == 2 and d.wstream_floats == plain.packed(64, "cpu")[0].numel()
    assert list(model.state_dict())[-1].startswith("nerf_dec.module.")


def test_render_refuses_cpu_tensors():
    from matchnerf_amd.models import models_dict
    model = models_dict["matchnerf"](_opts())
    sc = syn.make_scene(32, 32, 3)
    batch = EasyDict({k: torch.from_numpy(v) for k, v in sc.items()})
    with pytest.raises((RuntimeError, hip.MnerfError)):
        with torch.no_grad():
            model(batch, mode="test")


def test_video_paths_match_reference(golden):
    g = golden("video_paths")
    assert np.allclose(video.interpolate_render_path(g["c2ws"], 8), g["interpolate"], atol=1e-6)
    assert np.allclose(video.spiral_render_path(g["c2ws"], [2.0, 6.0], rads_scale=0.3, n_views=8), g["spiral"], atol=1e-6)


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports exactly what include/mnerf.h declares."""
    lib = hip.load()
    assert lib.mnerf_abi_version() == hip.MNERF_ABI_VERSION
    header = open(os.path.join(REPO, "include", "mnerf.h")).read()
    declared = sorted(set(re.findall(r"\b(mnerf_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mnerf_last_error() is not None
    # host-only entry points are callable without a GPU
    assert lib.mnerf_render_workspace_bytes(4096, 64, 24) == 4096 * 64 * 24 * 4
    from matchnerf_amd import cond_nerf as CN
    for cd, cs, L in ((22, 24, 10), (50, 56, 10), (22, 24, 6), (74, 80, 10)):
        assert lib.mnerf_decoder_wstream_floats(cd, cs, L, hip.WSTREAM_F32) == CN.decoder_schedule(cs, L)[1]
        assert lib.mnerf_decoder_wstream_floats(cd, cs, L, hip.WSTREAM_BF16X3) == CN.decoder_schedule16(cd, L)[1]
        assert lib.mnerf_decoder_wstream_floats(cd, cs, L, hip.WSTREAM_F16X2) == CN.decoder_schedule_h(cd, L)[1]
    assert lib.mnerf_decoder_wstream_floats(22, 24, 10, 7) == -1


def test_struct_layouts_match_the_header():
    """ctypes mirrors of the by-value structs have the sizes the C side was compiled with."""
    lib = hip.load()
    for which, st in enumerate((hip.View, hip.Rays, hip.Scene, hip.Decoder, hip.EncoderLayer, hip.ConvLayer, hip.DecoderTrain,
                                hip.EncoderLayerTrain)):
        assert lib.mnerf_struct_size(which) == ctypes.sizeof(st), st.__name__
    assert lib.mnerf_struct_size(99) == -1
    assert ctypes.sizeof(hip.View) == 23 * 4


def test_pose_table_rows_and_ray_struct_fields():
    """mnerf_rays.pose_table (ABI 7): row layout [kinv 9 | c2w 12 | near | far | pad] as the header states it; make_rays only sets
    rays_per_pose together with a table"""
    header = open(os.path.join(REPO, "include", "mnerf.h")).read()
    assert int(re.search(r"#define MNERF_POSE_FLOATS (\d+)", header).group(1)) == hip.MNERF_POSE_FLOATS == 24
    assert int(re.search(r"#define MNERF_ABI_VERSION (\d+)", header).group(1)) == hip.MNERF_ABI_VERSION
    kinv = np.arange(9, dtype=np.float32).reshape(3, 3)
    c2w = 100 + np.arange(12, dtype=np.float32).reshape(3, 4)
    rows = hip.pose_table_rows([(kinv, c2w, 2.0, 6.0), (kinv * 2, c2w * 2, 1.0, 3.0)])
    assert rows.shape == (2, 24) and rows.dtype == np.float32
    assert (rows[0, :9] == kinv.reshape(-1)).all() and (rows[0, 9:21] == c2w.reshape(-1)).all()
    assert tuple(rows[0, 21:]) == (2.0, 6.0, 0.0) and tuple(rows[1, 21:23]) == (1.0, 3.0)
    r = hip.make_rays(128, 64, 8, 16, kinv, c2w, 2.0, 6.0, rays_per_pose=128)
    assert r.pose_table is None and r.rays_per_pose == 0
    r = hip.make_rays(256, 64, 8, 16, kinv, c2w, 2.0, 6.0, ray_begin=128, pose_table_ptr=0x1000, rays_per_pose=128)
    assert r.pose_table == 0x1000 and r.rays_per_pose == 128 and r.ray_begin == 128
    lib = hip.load()  # the predicate is host-only: NULL / malformed arguments answer 0 without a GPU
    assert lib.mnerf_render_takes_pose_table(None, None, 64, 128) == 0


def test_decoder_train_tensor_order_matches_the_header_and_the_module():
    """mnerf_decoder_train indexes the decoder's parameters by the MNERF_DT_* enum; the binding's name table must list them
    in that order, name every parameter of the CondNeRF module exactly once, and the knob hook must know its knobs."""
    header = open(os.path.join(REPO, "include", "mnerf.h")).read()
    body = header[header.index("MNERF_DT_PTS_W0"):header.index("MNERF_DEC_TENSORS")]
    enum = re.findall(r"\b(MNERF_DT_[A-Z0-9_]+)", body)
    expect = {"MNERF_DT_BIAS_W": "pts_bias.weight", "MNERF_DT_BIAS_B": "pts_bias.bias", "MNERF_DT_ALPHA_W": "alpha_linear.0.weight",
              "MNERF_DT_ALPHA_B": "alpha_linear.0.bias", "MNERF_DT_WQ": "ray_attention.w_qs.weight", "MNERF_DT_WK": "ray_attention.w_ks.weight",
              "MNERF_DT_WV": "ray_attention.w_vs.weight", "MNERF_DT_FC": "ray_attention.fc.weight",
              "MNERF_DT_LN_W": "ray_attention.layer_norm.weight", "MNERF_DT_LN_B": "ray_attention.layer_norm.bias",
              "MNERF_DT_OA0_W": "out_alpha_linear.0.weight", "MNERF_DT_OA0_B": "out_alpha_linear.0.bias",
              "MNERF_DT_OA2_W": "out_alpha_linear.2.weight", "MNERF_DT_OA2_B": "out_alpha_linear.2.bias",
              "MNERF_DT_FEAT_W": "feature_linear.weight", "MNERF_DT_FEAT_B": "feature_linear.bias",
              "MNERF_DT_VIEWS_W": "views_linears.0.weight", "MNERF_DT_VIEWS_B": "views_linears.0.bias",
              "MNERF_DT_RGB_W": "rgb_linear.weight", "MNERF_DT_RGB_B": "rgb_linear.bias"}
    assert enum[0] == "MNERF_DT_PTS_W0" and "MNERF_DT_BIAS_W = 12" in header
    names = [f"pts_linears.{i}.{k}" for i in range(6) for k in ("weight", "bias")] + [expect[e] for e in enum[1:]]
    assert tuple(names) == hip.DEC_TRAIN_TENSORS and len(names) == 32
    opt = options.load_options("configs/test.yaml", verbose=False)
    opt.device = "cpu"
    from matchnerf_amd.models import models_dict
    dec = models_dict[opt.model](opt).nerf_dec
    assert sorted(n for n, _ in dec.named_parameters()) == sorted(hip.DEC_TRAIN_TENSORS)
    lib = hip.load()
    old, back = ctypes.c_int(-7), ctypes.c_int(-7)
    assert lib.mnerf_debug_set_knob(b"decoder_pp", 0, ctypes.byref(old)) == 0 and old.value in (0, 1)
    assert lib.mnerf_debug_set_knob(b"decoder_pp", old.value, ctypes.byref(back)) == 0 and back.value == 0
    assert lib.mnerf_debug_set_knob(b"no_such_knob", 1, None) == hip.MNERF_E_RANGE and b"unknown knob" in lib.mnerf_last_error()
    # a knob whose legitimate value is -1, after a failed call left its message in the error channel: no spurious error
    with hip.knob("cv_uvpair", 1):
        pass
    assert lib.mnerf_debug_set_knob(b"cv_uvpair", -1, ctypes.byref(old)) == 0 and old.value == -1
