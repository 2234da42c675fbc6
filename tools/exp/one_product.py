"""Diagnostic: the bench frame through whatever library MNERF_LIB names; prints decoder / frame time and
saves the frame, or compares it with a saved one.  usage: one_product.py save|compare FILE"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from matchnerf_amd import hip

dev = torch.device("cuda:0")
opt, model, _ = bench.build_model(dev)
_, batch = bench.make_batch(dev, 0)
with torch.no_grad():
    out = model(batch, mode="test")
    timer = hip.KernelTimer()
    model.kernel_timer = timer
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = model(batch, mode="test")
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
k = timer.summary()
print(f"lib {os.path.basename(hip.lib_path())}: frame {ms:.2f} ms, decoder {k['decoder']['total_ms'] / 5:.2f} ms, cost volume "
      f"{k['cost_volume']['total_ms'] / 5:.2f} ms per frame")
rgb = out.rgb[0].cpu()
if sys.argv[1] == "save":
    torch.save(rgb, sys.argv[2])
else:
    ref = torch.load(sys.argv[2])
    print(f"RGB L-inf vs the saved frame: {float((rgb - ref).abs().max()):.3e} (mean {float((rgb - ref).abs().mean()):.3e})")
