"""rocprofv3 driver: N steady-state encoder passes at BASELINE config[1] (3 views, 512x640)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.backends.cudnn.benchmark = True
opt, model, _ = bench.build_model(torch.device("cuda:0"))
_, batch = bench.make_batch(torch.device("cuda:0"), 0)
with torch.no_grad():
    for _ in range(2):
        model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for _ in range(n):
        model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
    torch.cuda.synchronize()
    print("encoder ms", (time.perf_counter() - t) / n * 1e3)
