"""Steady-state per-kernel breakdown of the encoder with torch.profiler (after MIOpen find)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

torch.backends.cudnn.benchmark = True
opt, model, _ = bench.build_model(torch.device("cuda:0"))
_, batch = bench.make_batch(torch.device("cuda:0"), 0)
with torch.no_grad():
    for _ in range(3):
        model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(5):
            model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
