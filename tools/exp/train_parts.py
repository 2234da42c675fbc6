"""What a training iteration pays outside forward/backward: re-packing the weight streams after an optimizer step (transformer
on the device, decoder), gradient clipping + AdamW.  usage: train_parts.py [sample_intvs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
opt, model, _ = bench.build_model(dev, 3, S)
model.train()
opt.nerf.rand_rays_train = 1024
opt.nerf.sample_stratified = True
_, batch = bench.make_batch(dev, 0)
optim = torch.optim.AdamW(model.parameters(), lr=1e-7, weight_decay=1e-4)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def bump():
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.0)


def fwd_bwd():
    optim.zero_grad(set_to_none=True)
    out = model(batch, mode="train")
    gt = batch.images[:, -1].reshape(1, 3, -1).permute(0, 2, 1)[:, out.ray_idx]
    ((out.rgb - gt) ** 2).mean().backward()


ft = model.feat_enc.transformer
print("bump only                 %.2f ms" % timed(bump))
print("bump + transformer repack %.2f ms" % timed(lambda: (bump(), ft.refresh_packs(dev))))
print("bump + decoder repack     %.2f ms" % timed(lambda: (bump(), model.nerf_dec.packed(S, dev))))
print("forward + backward        %.2f ms" % timed(fwd_bwd))
fwd_bwd()
print("clip + AdamW step         %.2f ms" % timed(lambda: (torch.nn.utils.clip_grad_norm_(model.feat_enc.parameters(), 1.0), optim.step())))
print("full iteration            %.2f ms" % timed(lambda: (fwd_bwd(), torch.nn.utils.clip_grad_norm_(model.feat_enc.parameters(), 1.0), optim.step())))
