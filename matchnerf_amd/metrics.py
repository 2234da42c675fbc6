"""Image metrics as the reference's evaluation computes them (misc/metrics.py): PSNR over the pixels kept by a mask (DTU:
ground-truth depth != 0) or, without a mask, over the 80 % centre crop; SSIM as scikit-image 0.19.2's
`structural_similarity(pred, gt, channel_axis=-1)` (the version requirements.txt pins) — restated here because scikit-image is
not part of this image: 7x7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03 and, for FLOAT images with no
`data_range` given, data_range = 2 (that version takes the dtype's nominal range [-1, 1]; the reference passes none).
LPIPS: the reference calls `lpips.LPIPS(net='vgg')` (misc/metrics.py:16,48-52).  Neither the `lpips` package nor the two weight
files it downloads (torchvision's ImageNet VGG-16 and the learned linear heads, `vgg.pth` of lpips v0.1) are available offline, so
`LPIPSVGG` below restates the published network (Zhang et al. 2018, lpips v0.1 'vgg' variant) and loads those two files from
where the user put them (`load_lpips`); PARITY UNPINNED — without the weights there is no number to compare, the tests check the
structure against an independent functional evaluation with the files' key names and shapes.  `EvalTools` takes any callable.
PARITY (rest): PSNR is checked against the reference's formula; the SSIM restatement against a direct per-window evaluation of the
published definition and against hand-derived closed-form vectors (tests/golden/ssim_hand_derived.json, generator
tools/gen_ssim_golden.py; tests/test_datasets.py) — not against scikit-image itself, which this image does not have."""
from collections import OrderedDict

import numpy as np


def psnr(pred, gt, invalid_mask=None):
    """pred, gt: float arrays [H,W,3] in [0,1]; invalid_mask: bool [H,W] of pixels to DROP."""
    pred = np.asarray(pred, np.float64)
    gt = np.asarray(gt, np.float64)
    if invalid_mask is not None:
        keep = ~np.asarray(invalid_mask, bool)
        mse = np.mean((pred[keep] - gt[keep]) ** 2)
    else:
        hc, wc = np.array(pred.shape[:2]) // 10
        mse = np.mean((pred[hc:-hc, wc:-wc] - gt[hc:-hc, wc:-wc]) ** 2)
    return float(-10.0 * np.log(mse) / np.log(10.0))


def _ssim_plane(x, y, data_range, win_size=7, k1=0.01, k2=0.03):
    from scipy.ndimage import uniform_filter
    ft = np.float32 if x.dtype == np.float32 else np.float64
    x, y = x.astype(ft, copy=False), y.astype(ft, copy=False)
    n = win_size ** x.ndim
    cov_norm = n / (n - 1.0)  # sample covariance
    ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
    uxx, uyy, uxy = uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size), uniform_filter(x * y, size=win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win_size - 1) // 2
    return s[pad:-pad, pad:-pad].mean(dtype=np.float64)  # windows that lie inside the image


def ssim(pred, gt, data_range=None):
    """Mean SSIM over the channels of [H,W,C] images (misc/metrics.py:43-45)."""
    pred, gt = np.asarray(pred), np.asarray(gt)
    if pred.shape != gt.shape or pred.ndim != 3:
        raise ValueError(f"ssim: expected two [H,W,C] images of one shape, got {pred.shape} and {gt.shape}")
    if min(pred.shape[:2]) < 7:
        raise ValueError("ssim: the 7x7 window does not fit the image")
    if data_range is None:
        if np.issubdtype(pred.dtype, np.floating):
            data_range = 2.0
        else:
            info = np.iinfo(pred.dtype)
            data_range = float(info.max - info.min)
    return float(np.mean([_ssim_plane(pred[..., c], gt[..., c], data_range) for c in range(pred.shape[-1])]))


class EvalTools:
    """misc/metrics.py:10-69: set_inputs() applies the DTU mask (masked pixels zeroed in both images) or the 80 % centre crop,
    get_metrics() evaluates PSNR / SSIM (/ LPIPS when a callable `lpips_fn(pred, gt) -> float` on [H,W,3] arrays is given)."""

    def __init__(self, device=None, lpips_fn=None):
        self.device, self.lpips_fn = device, lpips_fn
        self.support_metrics = ["PSNR", "SSIM"] + (["LPIPS"] if lpips_fn is not None else [])

    def set_inputs(self, pred_img, gt_img, img_mask=None):
        self.full_pred, self.full_gt, self.img_mask = pred_img, gt_img, img_mask
        if img_mask is not None:
            self.proc_pred, self.proc_gt = pred_img.copy(), gt_img.copy()
            self.proc_pred[img_mask] = 0.0
            self.proc_gt[img_mask] = 0.0
        else:
            hc, wc = np.array(pred_img.shape[:2]) // 10
            self.proc_pred, self.proc_gt = pred_img[hc:-hc, wc:-wc], gt_img[hc:-hc, wc:-wc]

    def _eval(self, metric, pred, gt, use_mask):
        if metric == "PSNR":
            if use_mask:
                return float(-10.0 * np.log(np.mean((pred[~self.img_mask] - gt[~self.img_mask]) ** 2)) / np.log(10.0))
            return float(-10.0 * np.log(np.mean((pred - gt) ** 2)) / np.log(10.0))
        if metric == "SSIM":
            return ssim(pred, gt)
        return float(self.lpips_fn(pred, gt))

    def get_metrics(self, metrics=None, return_full=False):
        out = OrderedDict()
        for metric in metrics or self.support_metrics:
            if metric not in self.support_metrics:
                raise ValueError(f"only support metrics: [{','.join(self.support_metrics)}]")
            out[metric] = self._eval(metric, self.proc_pred, self.proc_gt, self.img_mask is not None)
            if return_full:
                out[f"{metric}_Full"] = self._eval(metric, self.full_pred, self.full_gt, False)
        return out


# ----------------------------------------------------------------------------- LPIPS (VGG-16 variant of lpips v0.1)

LPIPS_VGG_SLICES = ((0, 4), (4, 9), (9, 16), (16, 23), (23, 30))      # torchvision vgg16.features up to relu1_2 .. relu5_3
LPIPS_VGG_CONVS = {0: (3, 64), 2: (64, 64), 5: (64, 128), 7: (128, 128), 10: (128, 256), 12: (256, 256), 14: (256, 256),
                   17: (256, 512), 19: (512, 512), 21: (512, 512), 24: (512, 512), 26: (512, 512), 28: (512, 512)}
LPIPS_VGG_POOLS = (4, 9, 16, 23)
LPIPS_CHANNELS = (64, 128, 256, 512, 512)


def _lpips_module():
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    class LPIPSVGG(nn.Module):
        """d(x, y) = sum_l mean_hw( w_l . (unit(f_l(x)) - unit(f_l(y)))^2 ) over the five VGG-16 stages, inputs RGB in [-1, 1]
        [N,3,H,W] shifted / scaled by the package's ScalingLayer constants; unit() divides by the channel norm + 1e-10; w_l are
        the learned non-negative 1x1 heads (`lin{l}.model.1.weight`, [1,C_l,1,1]).  Parameters carry torchvision's and lpips's own
        key names so that their files load with strict=True: `features.<i>.{weight,bias}` and `lin<l>.model.1.weight`."""

        def __init__(self):
            super().__init__()
            self.features = nn.ModuleDict({str(i): nn.Conv2d(ci, co, 3, padding=1) for i, (ci, co) in LPIPS_VGG_CONVS.items()})
            for l, c in enumerate(LPIPS_CHANNELS):
                head = nn.Module()
                head.model = nn.ModuleDict({"1": nn.Conv2d(c, 1, 1, bias=False)})
                setattr(self, f"lin{l}", head)
            self.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188]).view(1, 3, 1, 1), persistent=False)
            self.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450]).view(1, 3, 1, 1), persistent=False)

        def stages(self, x):
            out = []
            for lo, hi in LPIPS_VGG_SLICES:
                for i in range(lo, hi):
                    if i in LPIPS_VGG_CONVS:
                        x = F.relu(self.features[str(i)](x))
                    elif i in LPIPS_VGG_POOLS:
                        x = F.max_pool2d(x, 2, 2)
                out.append(x)
            return out

        def forward(self, x, y):
            fx, fy = self.stages((x - self.shift) / self.scale), self.stages((y - self.shift) / self.scale)
            total = 0.0
            for l, (a, b) in enumerate(zip(fx, fy)):
                a = a / (a.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
                b = b / (b.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
                total = total + getattr(self, f"lin{l}").model["1"]((a - b) ** 2).mean((2, 3), keepdim=True)
            return total

    return LPIPSVGG


def load_lpips(vgg16_path=None, lin_path=None, device="cpu"):
    """-> callable `lpips_fn(pred, gt) -> float` on [H,W,3] float arrays in [0,1] (what `EvalTools(lpips_fn=...)` takes), built
    from the two files the `lpips` package would have downloaded: torchvision's `vgg16-397923af.pth` (or any state dict with
    `features.*` keys) and lpips v0.1's `vgg.pth`.  Paths default to $MNERF_LPIPS_VGG16 / $MNERF_LPIPS_LIN and torch hub's cache."""
    import os
    import torch
    hub = os.path.join(os.path.expanduser(os.environ.get("TORCH_HOME", "~/.cache/torch")), "hub", "checkpoints")
    vgg16_path = vgg16_path or os.environ.get("MNERF_LPIPS_VGG16") or os.path.join(hub, "vgg16-397923af.pth")
    lin_path = lin_path or os.environ.get("MNERF_LPIPS_LIN") or os.path.join(hub, "lpips_vgg.pth")
    for what, path in (("torchvision VGG-16 weights", vgg16_path), ("lpips v0.1 linear heads (vgg.pth)", lin_path)):
        if not os.path.isfile(path):
            raise FileNotFoundError(f"LPIPS: {what} not found at {path}; there is no network here — copy the file and point "
                                    "MNERF_LPIPS_VGG16 / MNERF_LPIPS_LIN at it (the reference downloads both through the lpips package)")
    net = _lpips_module()()
    sd = {k: v for k, v in torch.load(vgg16_path, map_location="cpu", weights_only=True).items() if k.startswith("features.")}
    sd.update(torch.load(lin_path, map_location="cpu", weights_only=True))
    net.load_state_dict(sd, strict=True)
    net = net.to(device).eval()

    @torch.no_grad()
    def lpips_fn(pred, gt):
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].permute(0, 3, 1, 2).float().to(device) * 2 - 1.0
        return float(net(to_t(pred), to_t(gt)).item())

    return lpips_fn
