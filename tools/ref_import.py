"""Import the reference implementation as a CPU oracle — BUILD CONTAINER ONLY.

This is test infrastructure.  It imports
``/root/reference`` *in place* (nothing is copied), after registering in-memory stubs
for modules the reference imports at module scope but never executes on the hot path
(easydict, ipdb, termcolor, skvideo, cv2, torchvision).  ``/root/reference`` does not
exist on the GPU box; nothing under ``tests/ -m gpu``, ``bench.py`` or
``__graft_entry__.smoke()`` may import this file.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Compose:
    """Stand-in for torchvision.transforms.Compose (absent from this image): apply in order."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class _ToTensor:
    """Stand-in for torchvision.transforms.ToTensor on an 8-bit PIL image: HWC uint8 -> CHW float32 / 255 (its documented
    behaviour).  The reference's dataset modules build `Compose([ToTensor()])` at construction (datasets/llff.py:95-98)."""

    def __call__(self, img):
        import numpy as np
        import torch
        a = np.asarray(img.convert("RGB") if img.mode != "RGB" and img.mode != "RGBA" else img, np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255.0)


def import_reference():
    """Returns the reference's (options, MatchNeRF, EasyDict) with cwd switched to REF_ROOT
    (its options.py opens 'configs/...' relatively, options.py:54,64)."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (only available in the build container)")
    if REPO_ROOT not in sys.path:
        sys.path.insert(0, REPO_ROOT)
    from matchnerf_amd.edict import EasyDict

    if "easydict" not in sys.modules:
        _stub("easydict", EasyDict=EasyDict)
    _stub("ipdb", set_trace=lambda *a, **k: None)
    _stub("termcolor", colored=lambda s, *a, **k: s)
    sk = _stub("skvideo")
    sk.io = _stub("skvideo.io")
    _stub("cv2", COLORMAP_JET=2)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", Compose=_Compose, ToTensor=_ToTensor,
                                Lambda=lambda fn: fn)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    os.chdir(REF_ROOT)
    import options as ref_options  # noqa: E402
    from models.matchnerf import MatchNeRF as RefMatchNeRF  # noqa: E402
    return ref_options, RefMatchNeRF, EasyDict


def reference_options(yaml_name="test", **overrides):
    """Merged reference options for ``configs/<yaml_name>.yaml`` with device forced to cpu.
    ``overrides`` use dotted keys, e.g. {'nerf.sample_intvs': 64}."""
    ref_options, _, EasyDict = import_reference()
    opt = ref_options.load_options(f"configs/{yaml_name}.yaml")
    opt.device = "cpu"
    for k, v in overrides.items():
        node = opt
        parts = k.split(".")
        for p in parts[:-1]:
            if p not in node or node[p] is None:
                node[p] = EasyDict()
            node = node[p]
        node[parts[-1]] = v
    return opt
