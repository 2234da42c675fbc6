"""Small driver for rocprofv3 passes: `frames` full-frame renders (encoder + ray chunks) of one BASELINE configuration,
no CPU baseline, no warm-up loop.   prof_render.py [frames] [c2 | c3 | c5 | s<N>]
  c2 (default) BASELINE config[1]: 512x640, 3 views, 64 samples;  c3 config[2]: Blender-like 800x800, 128 samples, white
  background;  c5 config[4]: 10 source views at 512x640;  s<N>: config[1]'s frame at N samples per ray (s256: the sample count
  of configs/test_video_own.yaml, decoder_pp_kernel<256>)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
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
else:
    opt, model, _ = bench.build_model(dev)
    _, batch = bench.make_batch(dev, 0)
with torch.no_grad():
    for _ in range(frames):
        out = model(batch, mode="test")
torch.cuda.synchronize()
print("rendered", cfg, tuple(out.rgb.shape), float(out.rgb.mean()))
