"""Phase timeline of cost_volume_tile_kernel (library built with -DCVT_STATS, selected with MNERF_LIB):
usage: MNERF_LIB=.../libmnerf_hip_cvst.so cvt_stats.py [c2|c5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

dev = torch.device("cuda:0")
dbg = torch.zeros(16, dtype=torch.int64, device=dev)
os.environ["MNERF_CVDBG_PTR"] = str(dbg.data_ptr())
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
if cfg == "c5":
    opt, model, _ = bench.build_model(dev, 10, 64)
    _, batch = bench.make_batch(dev, 0, 512, 640, 10, seed=32)
else:
    opt, model, _ = bench.build_model(dev)
    _, batch = bench.make_batch(dev, 0)
with torch.no_grad():
    model(batch, mode="test")
    torch.cuda.synchronize()
    dbg.zero_()
    model(batch, mode="test")
    torch.cuda.synchronize()
d = dbg.cpu().tolist()
names = ["pass1", "B claim", "wait B", "C copy", "wait C", "D walk", "write", "loop"]
waves = d[10] * 4
tot = sum(d[:8])
print(f"{cfg}: {d[10]} workgroups, {tot / waves / 100e6 * 1e3:.2f} ms per wave at 100 MHz memtime")
for n, v in zip(names, d[:8]):
    print(f"  {n:8s} {v / waves:12.0f} ticks/wave  {100.0 * v / tot:5.1f} %")
print(f"  claims {d[8]}  fallbacks {d[9]}  ({100.0 * d[9] / max(d[8], 1):.2f} %)   wave-walks with a fallback: {d[11]} of {d[8] // 256} "
      f"({100.0 * d[11] / max(d[8] // 256, 1):.1f} %)")
