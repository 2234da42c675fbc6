"""Video of a small frame (the reference's render_video loop, models/matchnerf.py:42-71): poses per launch through the pose
table (mnerf_rays.pose_table) against one pose per launch.  usage: video_time.py [H W] [n_poses] [reps] [sample_intvs]     (VT_MAX_RAYS: rays per pose-table launch)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from matchnerf_amd import hip

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 160)
n_poses = int(sys.argv[3]) if len(sys.argv) > 3 else 30
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
if os.environ.get("VT_MAX_RAYS"):  # experiment: launch size (e.g. two 512 x 640 poses in one launch)
    import matchnerf_amd.matchnerf as mn
    mn.MAX_RAYS_PER_POSE_LAUNCH = int(os.environ["VT_MAX_RAYS"])
opt, model, _ = bench.build_model(dev, 3, int(sys.argv[5]) if len(sys.argv) > 5 else 64)
opt.nerf.video_n_frames = n_poses
_, batch = bench.make_batch(dev, 0, h, w, seed=41)
res = {}
for batching in (False, True, False, True):
    model.pose_batching = batching
    with torch.no_grad():
        out = model(batch, mode="test", render_video=True)
        timer = hip.KernelTimer()
        model.kernel_timer = timer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = model(batch, mode="test", render_video=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        model.kernel_timer = None
    k = timer.summary()
    bits = int(out.rgb.contiguous().view(torch.int32).to(torch.int64).sum()) & 0xffffffffffff
    print(f"{h}x{w} x {n_poses} poses, {'pose table' if batching else 'pose by pose'}: video {ms:.2f} ms = {ms / n_poses:.3f} ms per frame "
          f"({n_poses * h * w / ms / 1e3:.2f} M rays/s); decoder {k['decoder']['total_ms'] / reps:.2f} ms in {k['decoder']['launches'] // reps} launches, "
          f"cost volume {k['cost_volume']['total_ms'] / reps:.2f} ms; bits {bits:012x}")
