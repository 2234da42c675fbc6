"""One train_iteration-shaped step under the profiler: tools/exp/train_step_prof.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = False  # (MIOpen would otherwise time every solver, naive ones included, inside the trace)
opt, model, _ = bench.build_model(dev)
model.train()
opt.nerf.rand_rays_train = 1024
_, batch = bench.make_batch(dev, 0)
params = [p for p in model.parameters() if p.requires_grad]
torch.manual_seed(0)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for p in params:
        p.grad = None
    out = model(batch, mode="train")
    gt = batch.images[:, -1].reshape(1, 3, -1).permute(0, 2, 1)[:, out.ray_idx]
    ((out.rgb - gt) ** 2).mean().backward()
torch.cuda.synchronize()
print("done")
