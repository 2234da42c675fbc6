"""Phase timeline of cost_volume_mm_kernel (library built with -DCVM_STATS, selected with MNERF_LIB):
usage: MNERF_LIB=.../libmnerf_hip_cvms.so cvmm_stats.py [c2|c5]"""
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
names = ["pass1", "pass1b", "side setup", "side a", "side b", "cosines", "write", "loop head"]
waves = d[8]
tot = sum(d[:8])
print(f"{cfg}: {waves} waves, {tot / waves:.0f} ticks per wave")
for n, v in zip(names, d[:8]):
    print(f"  {n:10s} {v / waves:12.0f} ticks/wave  {100.0 * v / tot:5.1f} %")
