"""cost-volume backward alone at the training shape (1 024 random rays x 64 samples, 3 views at 512 x 640): ms per call"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from matchnerf_amd import camera, hip

dev = torch.device("cuda:0")
opt, model, _ = bench.build_model(dev)
_, batch = bench.make_batch(dev, 0)
with torch.no_grad():
    ref_images = batch.images[:, :3]
    feats = model.get_img_feat(ref_images, cur_n_src_views=3)
    tgt, ref_poses = model.extract_poses(batch)
    ref_host, images_cl = model._frame_ctx(ref_poses, ref_images)
    sc = model._scene(0, ref_host, feats, images_cl)
    dec = model._decoder(64, dev)
ex, it, nf = model._tgt_host(tgt)
kinv, c2w = camera.target_ray_consts(ex[0], it[0], True)
torch.manual_seed(0)
idx = torch.randperm(512 * 640, device=dev)[:1024].int()
rays = hip.make_rays(1024, 64, 512, 640, kinv, c2w, nf[0, 0], nf[0, 1], ray_idx_ptr=idx.data_ptr())
g_cond = torch.randn(1024 * 64, dec.cond_stride, device=dev)
grads = [torch.zeros_like(f[0]) for f in feats]
fn = lambda: hip.cost_volume_backward(sc, rays, dec.cond_stride, g_cond, grads)
fn()
torch.cuda.synchronize()
for g in grads:
    g.zero_()
fn()
torch.cuda.synchronize()
chk = [float(g.double().abs().sum()) for g in grads]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fn()
e1.record()
torch.cuda.synchronize()
print(f"MNERF_CV_BWD_WALK={os.environ.get('MNERF_CV_BWD_WALK', '1')}: {e0.elapsed_time(e1) / 5:.3f} ms per call, |grad| sums {chk[0]:.6e} {chk[1]:.6e}")
