"""Debug experiment: per-phase s_memtime stamps of the fused decoder kernel (needs the
-DMNERF_TIMELINE build: tools/exp/build_timeline.sh -> matchnerf_amd/libmnerf_hip_tl.so)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

tl = torch.zeros(128 * 4 * 4 * 20, dtype=torch.int64, device="cuda")
os.environ["MNERF_TIMELINE_PTR"] = str(tl.data_ptr())
import bench  # noqa: E402

opt, model, _ = bench.build_model(torch.device("cuda:0"))
_, batch = bench.make_batch(torch.device("cuda:0"), 0)
with torch.no_grad():
    model(batch, mode="test")
    tl.zero_()
    model(batch, mode="test")
torch.cuda.synchronize()
t = tl.cpu().numpy().reshape(128, 4, 4, 20)  # [wg, tile, wave, point]
names = ["prologue+seg0", "film", "L0", "L1-4", "L5", "feature", "views", "rgb", "alpha+qkv+kv-setup", "-", "attention",
         "fc/LN/sigma", "composite", "-"]
d = np.diff(t[..., :15].astype(np.float64), axis=-1)  # [wg,tile,wave,14]
valid = t[..., 14] > 0
print("phase cycles (mean over waves / WGs / tiles 1..3), total per tile:")
dm = d[:, 1:, :, :][valid[:, 1:, :]].reshape(-1, 14)
for n, v in zip(names, dm.mean(0)):
    print(f"  {n:16s} {v:10.0f}  ({100 * v / dm.sum(1).mean():5.1f} %)")
print("  total            %10.0f cycles per tile" % dm.sum(1).mean())
w = t[:, 1:, :, 15:17].astype(np.float64)[valid[:, 1:, :]].reshape(-1, 2)
print("  inside the weight-segment hand-offs: wait for own DMA %.0f, wait at the barrier %.0f cycles per tile (%.1f %% / %.1f %%)"
      % (w[:, 0].mean(), w[:, 1].mean(), 100 * w[:, 0].mean() / dm.sum(1).mean(), 100 * w[:, 1].mean() / dm.sum(1).mean()))
sub = t[:, 1:, :, :].astype(np.float64)[valid[:, 1:, :]]
print("  alpha stage (incl. tail DMA + barrier) %.0f | qkv projections %.0f | K/V/Q to LDS + barrier %.0f"
      % ((sub[:, 17] - sub[:, 8]).mean(), (sub[:, 18] - sub[:, 17]).mean(), (sub[:, 9] - sub[:, 18]).mean()))
print("  per wave barrier wait:", [int(v) for v in t[0, 1, :, 16]], " DMA wait:", [int(v) for v in t[0, 1, :, 15]])

def where(h):
    h = int(h)
    return (h & 0xF, (h >> 4) & 3, (h >> 8) & 0xF, (h >> 12) & 1, (h >> 13) & 7)  # slot, simd, cu, sh, se

print("pairs (block b, b+256): HW (slot,simd,cu,sh,se) and tile start times relative to b's tile 0")
for b in (0, 8, 1, 9, 2):
    a, c = t[b], t[b + 64]
    base = a[0, 0, 0]
    print(f" b={b:3d} {where(a[0,0,19])}  tiles start {[int(a[k,0,0]-base) for k in range(4)]}  attention start {[int(a[k,0,10]-base) for k in range(4)]}")
    print(f" b={b+256:3d} {where(c[0,0,19])}  tiles start {[int(c[k,0,0]-base) for k in range(4)]}  attention start {[int(c[k,0,10]-base) for k in range(4)]}")
