"""Small driver for rocprofv3 counter passes: one encoder pass + one full-frame render at
BASELINE config[1] (5 launches of each render kernel), no CPU baseline, no warm-up loop."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt, model, _ = bench.build_model(torch.device("cuda:0"))
_, batch = bench.make_batch(torch.device("cuda:0"), 0)
with torch.no_grad():
    for _ in range(frames):
        out = model(batch, mode="test")
torch.cuda.synchronize()
print("rendered", tuple(out.rgb.shape), float(out.rgb.mean()))
