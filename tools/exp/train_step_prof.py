"""Full train_iterations (zero_grad, forward, loss, backward, clipping, AdamW step: coach.py:215-243) under the profiler:
tools/exp/train_step_prof.py [iterations] [sample_intvs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = False  # (MIOpen would otherwise time every solver, naive ones included, inside the trace)
opt, model, _ = bench.build_model(dev, 3, int(sys.argv[2]) if len(sys.argv) > 2 else 64)
model.train()
opt.nerf.rand_rays_train = 1024
_, batch = bench.make_batch(dev, 0)
optim = torch.optim.AdamW(model.parameters(), lr=1e-7, weight_decay=1e-4)
opt.nerf.sample_stratified = True
torch.manual_seed(0)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    optim.zero_grad(set_to_none=True)
    out = model(batch, mode="train")
    gt = batch.images[:, -1].reshape(1, 3, -1).permute(0, 2, 1)[:, out.ray_idx]
    ((out.rgb - gt) ** 2).mean().backward()
    torch.nn.utils.clip_grad_norm_(model.feat_enc.parameters(), 1.0)
    optim.step()
torch.cuda.synchronize()
print("done")
