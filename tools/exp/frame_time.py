"""Frame / kernel times of one BASELINE configuration with whatever library and knobs the environment selects.
usage: frame_time.py [c2|c3|c5|s<N>|demo] [frames]     s<N>: config[1]'s frame at N samples per ray; demo: demo_own.yaml's shape"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from matchnerf_amd import hip

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
if cfg == "c3":
    opt, model, _ = bench.build_model(dev, 3, 128)
    model.nerf_setbg_opaque = True
    _, batch = bench.make_batch(dev, 0, 800, 800, 3, seed=31, wide=True, focal_scale=1.389, near_far=(2.0, 6.0))
elif cfg == "c5":
    opt, model, _ = bench.build_model(dev, 10, 64)
    _, batch = bench.make_batch(dev, 0, 512, 640, 10, seed=32)
elif cfg.startswith("s"):
    opt, model, _ = bench.build_model(dev, 3, int(cfg[1:]))
    _, batch = bench.make_batch(dev, 0)
elif cfg == "demo":  # configs/demo_own.yaml: 256x160, S = 128, IBRNet-style decoder switches
    opt, model, _ = bench.build_model(dev, 3, 128)
    _, batch = bench.make_batch(dev, 0, 160, 256, 3, seed=33)
else:
    opt, model, _ = bench.build_model(dev)
    _, batch = bench.make_batch(dev, 0)
with torch.no_grad():
    out = model(batch, mode="test")
    timer = hip.KernelTimer()
    model.kernel_timer = timer
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        out = model(batch, mode="test")
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / frames * 1e3
k = timer.summary()
knobs = {n: os.environ[n] for n in os.environ if n.startswith("MNERF_")}
bits = int(out.rgb.contiguous().view(torch.int32).to(torch.int64).sum()) & 0xffffffffffff  # equal bits <=> (almost surely) equal frames
print(f"{cfg} {knobs}: frame {ms:.2f} ms, decoder {k['decoder']['total_ms'] / frames:.2f}, cost volume {k['cost_volume']['total_ms'] / frames:.2f}, "
      f"rgb mean {float(out.rgb.mean()):.6f} bits {bits:012x}")
