"""Kernel time of the split-fp16 convolution at the encoder's main shapes ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from matchnerf_amd import gmflow as G, hip  # noqa: E402

shapes = [(3, 64, 64, 3, 1, 256, 320), (3, 96, 96, 3, 1, 128, 160), (3, 128, 128, 3, 1, 64, 80), (6, 128, 128, 3, 1, 128, 160)]
out = []
for (n, ci, co, k, s, h, w) in shapes:
    x = torch.randn(n, ci, h, w, device="cuda")
    ws, ew = G.pack_conv(torch.randn(co, ci, k, k) * 0.05)
    ws = torch.from_numpy(ws).cuda()
    reg = hip.absmax_regions(1, "cuda")
    hip.absmax(x, reg[0])
    y = None
    for _ in range(3):
        y = hip.conv2d(x, ws, None, ci, co, k, s, ew, reg[0], out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.conv2d(x, ws, None, ci, co, k, s, ew, reg[0], out=y)
    e1.record()
    torch.cuda.synchronize()
    out.append("%dx%d@%dx%d %.1f us" % (ci, co, h, w, e0.elapsed_time(e1) / 20 * 1e3))
print(" | ".join(out))
