"""Decoder backward at training size (1024 rays x 64 samples): mnerf_decoder_backward against float32 torch autograd through the
plain statement of CondNeRF.forward that the tests use (tests/gpu_helpers.py::decoder_torch)."""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch

import bench
from gpu_helpers import decoder_torch
from matchnerf_amd import hip

dev = torch.device("cuda:0")
opt, model, _ = bench.build_model(dev)
dec = model.nerf_dec
v, r, s = 3, 1024, 64
dc = sum(opt.encoder.cos_n_group) + 4 * v
stride = 24
n = r * s
g = torch.Generator().manual_seed(0)
x = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev)
dirs = torch.nn.functional.normalize(torch.randn(r, 3, generator=g), dim=-1).to(dev)
cond = torch.zeros(n, stride)
cond[:, :dc - v] = torch.randn(n, dc - v, generator=g) * 0.5
cond[:, dc - v:dc] = (torch.rand(n, v, generator=g) > 0.3).float()
cond = cond.to(dev)
g_rgb, g_sig = torch.randn(n, 3, generator=g).to(dev), torch.randn(r, s, generator=g).to(dev)
params = {k: p.detach() for k, p in dec.named_parameters()}


def hip_pass():
    return hip.decoder_backward(opt, params, v, x, dirs, cond, stride, g_rgb, g_sig)


def torch_pass():
    c = cond[:, :dc].reshape(r, s, dc).detach().clone().requires_grad_(True)
    with torch.enable_grad():
        rgb_s, sigma = decoder_torch(opt, dec, x.reshape(r, s, 3), dirs, c, v)
    return torch.autograd.grad([rgb_s, sigma], [c] + list(dec.parameters()), [g_rgb.reshape(r, s, 3), g_sig])


for name, fn in (("mnerf_decoder_backward", hip_pass), ("torch autograd re-evaluation", torch_pass)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per pass ({r} rays x {s} samples)")
