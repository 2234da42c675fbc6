"""Image metrics as the reference's evaluation computes them (misc/metrics.py): PSNR over the pixels kept by a mask (DTU:
ground-truth depth != 0) or, without a mask, over the 80 % centre crop; SSIM as scikit-image 0.19.2's
`structural_similarity(pred, gt, channel_axis=-1)` (the version requirements.txt pins) — restated here because scikit-image is
not part of this image: 7x7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03 and, for FLOAT images with no
`data_range` given, data_range = 2 (that version takes the dtype's nominal range [-1, 1]; the reference passes none).
LPIPS needs the `lpips` package and its pretrained VGG weights, neither available offline: `EvalTools` takes it as an optional
callable.  PARITY: PSNR is checked against the reference's formula; the SSIM restatement against a direct per-window evaluation of the
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
