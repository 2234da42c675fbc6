"""Matrix form of the cost volume (ABI v9: mnerf_cost_volume_operands + mnerf_scene.feat_op, csrc/cost_volume_mm.hip) against
the reference goldens and against the segment walk.

The goldens hold the reference's conditioning vectors for a list of pixels (``stage_rays``); the matrix form takes contiguous
pixel ranges, so a case renders the rows of the WHOLE image and the golden pixels are picked out of them.
Tolerance: 2e-5 on the cosines against the reference (tests/test_hip_kernels.py's gate); colours, masks and the constant column
are the walk kernel's arithmetic and must equal it bit for bit.
"""
import numpy as np
import pytest
import torch

from gpu_helpers import make_rays_struct, make_scene_struct
from helpers import linf
from test_hip_kernels import _case_on_gpu

pytestmark = pytest.mark.gpu

CASES = ["c1_default", "rect_wide", "nonlegacy", "v4", "inverse_depth", "demo_own_small", "demo_own"]


@pytest.fixture(scope="module")
def hip():
    from matchnerf_amd import hip as h
    h.load()
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return h


def _rows(hip, sc, cfg, batch, n, begin, cs, mm):
    rays = make_rays_struct(cfg, batch, n, ray_begin=begin)
    with hip.knob("cv_mm", int(mm)):
        out = hip.cost_volume(sc, rays, cs)
    return out.reshape(n, cfg.sample_intvs, cs)


@pytest.mark.parametrize("name", CASES)
def test_matrix_form_matches_reference_and_walk(hip, name):
    g, cfg, sd, batch, feats_gpu, img_gpu = _case_on_gpu(name)
    sc = make_scene_struct(cfg, batch, feats_gpu, img_gpu)
    h, w = batch["images"].shape[-2:]
    n = h * w
    dc = g["cond"].shape[-1]
    cs = ((dc + 1 + 7) // 8) * 8
    walk = _rows(hip, sc, cfg, batch, n, 0, cs, mm=0)
    keep = hip.cost_volume_operands(sc)  # noqa: F841  (sets sc.feat_op; the tensor must outlive the launches)
    assert sc.feat_op
    mm = _rows(hip, sc, cfg, batch, n, 0, cs, mm=1)
    torch.cuda.synchronize()
    sum_g = sum(cfg.cos_n_group)
    idx = torch.from_numpy(g["stage_rays"]).long().cuda()
    got = mm[idx].cpu()
    assert linf(got[..., :dc], g["cond"]) < 2e-5
    # against the walk on every pixel of the image: cosines to a few ulp of 1, everything else identical
    assert float((mm[..., :sum_g] - walk[..., :sum_g]).abs().max()) < 5e-6
    assert torch.equal(mm[..., sum_g:], walk[..., sum_g:])
    assert float(mm[..., :sum_g].abs().max()) <= 1 + 1e-5


@pytest.mark.parametrize("name", ["c1_default", "rect_wide", "demo_own_small"])
def test_matrix_form_is_chunk_invariant_and_reproducible(hip, name):
    """Launches over ragged pixel ranges (not aligned to the 8 x 4 tiles or to image rows) give the rows of one launch, bit
    for bit, and repeated launches are bit-identical."""
    g, cfg, sd, batch, feats_gpu, img_gpu = _case_on_gpu(name)
    sc = make_scene_struct(cfg, batch, feats_gpu, img_gpu)
    keep = hip.cost_volume_operands(sc)  # noqa: F841
    h, w = batch["images"].shape[-2:]
    n = h * w
    cs = ((g["cond"].shape[-1] + 1 + 7) // 8) * 8
    whole = _rows(hip, sc, cfg, batch, n, 0, cs, mm=1)
    for _ in range(3):
        assert torch.equal(_rows(hip, sc, cfg, batch, n, 0, cs, mm=1), whole)
    cuts = [0, 1, w - 3, 5 * w + 7, n // 2 + 11, n - 1, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = _rows(hip, sc, cfg, batch, hi - lo, lo, cs, mm=1)
        assert torch.equal(part, whole[lo:hi]), (lo, hi)


@pytest.mark.parametrize("focal", [6.0, 40.0])
def test_wide_footprints_take_the_general_loop(hip, focal):
    """A target camera with a very short focal length: neighbouring target pixels land many texels apart in the source maps, so
    a tile's footprints exceed the 8 x 8 chunk window (`big`) and the kernel walks the rays' whole chunk range with operands
    loaded on demand.  The segment walk (pinned to the reference by the goldens) is the yardstick; samples behind or beside the
    source cameras exercise the border clamp on the way."""
    g, cfg, sd, batch, feats_gpu, img_gpu = _case_on_gpu("c1_default")
    batch = {k: v.clone() for k, v in batch.items()}
    h, w = batch["images"].shape[-2:]
    batch["intrinsics"][0, -1] = torch.tensor([[focal, 0.0, w / 2.0], [0.0, focal, h / 2.0], [0.0, 0.0, 1.0]])
    sc = make_scene_struct(cfg, batch, feats_gpu, img_gpu)
    n = h * w
    dc = g["cond"].shape[-1]
    cs = ((dc + 1 + 7) // 8) * 8
    walk = _rows(hip, sc, cfg, batch, n, 0, cs, mm=0)
    keep = hip.cost_volume_operands(sc)  # noqa: F841
    mm = _rows(hip, sc, cfg, batch, n, 0, cs, mm=1)
    sum_g = sum(cfg.cos_n_group)
    assert float((mm[..., :sum_g] - walk[..., :sum_g]).abs().max()) < 5e-6
    assert torch.equal(mm[..., sum_g:], walk[..., sum_g:])
    assert not torch.equal(mm[..., :sum_g], torch.zeros_like(mm[..., :sum_g]))


def test_operand_image_round_trips_the_maps(hip):
    """hi + lo of the operand image, divided by the map's gain, gives back every texel to 2^-21 of the map's largest magnitude;
    the padding cells are zero."""
    g, cfg, sd, batch, feats_gpu, img_gpu = _case_on_gpu("c1_default")
    sc = make_scene_struct(cfg, batch, feats_gpu, img_gpu)
    buf = hip.cost_volume_operands(sc)
    torch.cuda.synchronize()
    raw = buf.cpu().numpy()
    maxmaps = 16 * 15
    gain = raw[:2 * maxmaps * 4].view(np.float32).reshape(2, maxmaps)
    off = 8192
    for s, f in enumerate(feats_gpu):
        P, _, fh, fw, C = f.shape
        nrp, rs = (fh + 1) // 2 + 1, fw + 3  # row pairs (+ a zero one), columns per row pair (+ three zero ones)
        map_bytes = nrp * rs * 1024
        for m in range(2 * P):
            cells = raw[off:off + map_bytes].view(np.float16).reshape(nrp, rs, 2, 128, 2).astype(np.float64)
            val = cells[:, :, 0] + cells[:, :, 1]                                # [rp, x, channel, row in pair]
            img = val.transpose(0, 3, 1, 2).reshape(nrp * 2, rs, 128)            # [y, x, channel]
            ref = f[m // 2, m % 2].cpu().numpy().astype(np.float64)
            gm = float(gain[s, m])
            assert gm > 0 and np.log2(gm) == np.round(np.log2(gm))
            amax = np.abs(ref).max()
            assert 2.0 ** 14 <= amax * gm < 2.0 ** 15
            assert np.abs(img[:fh, :fw] / gm - ref).max() <= amax * 2.0 ** -21
            assert np.all(img[fh:] == 0) and np.all(img[:, fw:] == 0)
            off += map_bytes
