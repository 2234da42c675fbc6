"""train_iteration time (zero_grad, forward, loss, backward, clipping, AdamW step) with the CNN's training path on this library's
kernels (default) or on torch's op chain: MNERF_TRAIN_CNN=torch.  usage: train_time.py [sample_intvs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
opt, model, _ = bench.build_model(dev, 3, S)
model.train()
opt.nerf.rand_rays_train = 1024
_, batch = bench.make_batch(dev, 0)
optim = torch.optim.AdamW(model.parameters(), lr=1e-7, weight_decay=1e-4)
opt.nerf.sample_stratified = True
torch.manual_seed(0)


def step():
    optim.zero_grad(set_to_none=True)
    out = model(batch, mode="train")
    gt = batch.images[:, -1].reshape(1, 3, -1).permute(0, 2, 1)[:, out.ray_idx]
    ((out.rgb - gt) ** 2).mean().backward()
    torch.nn.utils.clip_grad_norm_(model.feat_enc.parameters(), 1.0)
    optim.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"train iteration S={S} MNERF_TRAIN_CNN={os.environ.get('MNERF_TRAIN_CNN', 'hip')}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
