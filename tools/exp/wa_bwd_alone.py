"""The window-attention backward alone at the DTU shape (6 sequences, 64 x 80 tokens, 2 x 2 windows): ms per call, per kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from matchnerf_amd import hip

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
mk = lambda s: (torch.randn(6, 64 * 80, 128, generator=g) * s).to(dev)
q, k, v, go = mk(0.6), mk(0.8), mk(1.0), mk(1.0)
out = hip.window_attention(q, k, v, 64, 80, 2, True)
fn = lambda: hip.window_attention_backward(q, k, v, out, go, 64, 80, 2, True)
r = fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record()
torch.cuda.synchronize()
chk = sum(float(t.double().abs().sum()) for t in r)
print(f"{os.environ.get('MNERF_LIB', 'shipped')[-24:]} {os.environ.get('MNERF_WA_BWD_MATH', 'f16x3')}: {e0.elapsed_time(e1) / 10:.3f} ms per call, checksum {chk:.6e}", flush=True)
