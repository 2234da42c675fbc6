"""Frame / kernel times of a 3-view, 64-sample frame of any size.  usage: size_time.py H W [frames]  (MNERF_MAX_RAYS_PER_LAUNCH etc. from the environment)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from matchnerf_amd import hip

h, w = int(sys.argv[1]), int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
opt, model, _ = bench.build_model(dev)
_, batch = bench.make_batch(dev, 0, h, w, seed=7)
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
bits = int(out.rgb.contiguous().view(torch.int32).to(torch.int64).sum()) & 0xffffffffffff
print(f"{h}x{w} {knobs}: frame {ms:.2f} ms ({h * w / ms / 1e3:.2f} M rays/s), decoder {k['decoder']['total_ms'] / frames:.2f}, "
      f"cost volume {k['cost_volume']['total_ms'] / frames:.2f} in {k['cost_volume']['launches'] // frames} launches, bits {bits:012x}")
