"""PSNR as the reference's evaluation computes it (misc/metrics.py:19-41): over the pixels
kept by a mask (DTU: ground-truth depth != 0) or, without a mask, over the 80 % centre crop.
SSIM / LPIPS need scikit-image / lpips, which are not part of this image."""
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
