"""Debug experiment: cycles per loop phase of the pre-split window-attention kernel
(needs tools/exp/build_timeline.sh -> matchnerf_amd/libmnerf_hip_tl.so; run with MNERF_LIB pointing at it)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

tl = torch.zeros(64 * 4 * 6, dtype=torch.int64, device="cuda")
os.environ["MNERF_WA_TIMELINE_PTR"] = str(tl.data_ptr())
from matchnerf_amd import hip  # noqa: E402

g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(6, 64 * 80, 128, generator=g).cuda() for _ in range(3))
for shifted in (False, True):
    for _ in range(3):
        out = hip.window_attention(q, k, v, 64, 80, 2, shifted, math=hip.WA_PRESPLIT_F16)
    torch.cuda.synchronize()
    t = tl.cpu().numpy().reshape(64 * 4, 6).astype(float)
    names = ["issue DMA", "scores", "softmax", "output", "own DMA wait", "barrier"]
    tot = t.sum(1).mean()
    print(f"shifted={int(shifted)}: {tot / 40:.0f} cycles per tile and wave ({tot:.0f} per call)")
    for n, c in zip(names, t.mean(0)):
        print(f"  {n:14s} {c / 40:8.0f}  {100 * c / tot:5.1f} %")
