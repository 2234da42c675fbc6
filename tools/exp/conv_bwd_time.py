"""Backward of the backbone's convolutions at the DTU training shape (3 views of 512x640): conv_backward.hip against torch (MIOpen)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from matchnerf_amd import hip  # noqa: E402

SHAPES = [(3, 64, 64, 256, 320, 3, 1, 4), (3, 64, 96, 256, 320, 3, 2, 1), (3, 96, 96, 128, 160, 3, 1, 3), (3, 64, 96, 256, 320, 1, 2, 1),
          (3, 96, 128, 128, 160, 3, 2, 1), (3, 128, 128, 64, 80, 3, 1, 3), (3, 96, 128, 128, 160, 1, 2, 1), (3, 128, 128, 64, 80, 1, 1, 1)]


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = [0.0, 0.0, 0.0, 0.0]
for n, ci, co, h, w, k, s, count in SHAPES:
    x = torch.randn(n, ci, h, w, device="cuda")
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    dy = torch.randn(n, co, ho, wo, device="cuda") * 1e-3
    t_dx = timed(lambda: hip.conv2d_backward_data(dy, wt, h, w, s))
    t_dw = timed(lambda: hip.conv2d_backward_weight(x, dy, k, s))
    r_dx = timed(lambda: torch.nn.grad.conv2d_input(x.shape, wt, dy, s, k // 2))
    r_dw = timed(lambda: torch.nn.grad.conv2d_weight(x, wt.shape, dy, s, k // 2))
    gf = 2.0 * n * ho * wo * ci * co * k * k / 1e9
    print(f"{ci:3d}->{co:3d} k{k} s{s} {h}x{w} x{count}: dgrad {t_dx:7.1f} us ({gf / t_dx * 1e3:5.1f} TF/s) torch {r_dx:7.1f} | wgrad {t_dw:7.1f} us ({gf / t_dw * 1e3:5.1f} TF/s) torch {r_dw:7.1f}")
    for i, t in enumerate((t_dx, t_dw, r_dx, r_dw)):
        tot[i] += t * count
print(f"backbone total (counts applied): dgrad {tot[0] / 1e3:.2f} ms, wgrad {tot[1] / 1e3:.2f} ms | torch dgrad {tot[2] / 1e3:.2f} ms, wgrad {tot[3] / 1e3:.2f} ms")
