"""train_iteration-shaped step with the HIP window-attention backward vs the torch re-evaluation (MNERF_WA_BACKWARD)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = False
for enc, wa in (("hip", "hip"), ("torch", "hip"), ("torch", "torch"), ("hip", "hip")):
    os.environ["MNERF_ENC_BACKWARD"], os.environ["MNERF_WA_BACKWARD"] = enc, wa
    r = bench.train_step_workload(dev)
    print(f"transformer layers {enc}, attention backward {wa}:", r["ms_per_iteration"], "ms per iteration, decoder backward",
          r["decoder_backward_ms"], "loss", r["loss"], flush=True)
# the attention backward alone at the DTU shape: 6 sequences, 64 x 80 tokens, 2 x 2 windows
from matchnerf_amd import autograd as ag, hip
g = torch.Generator().manual_seed(0)
mk = lambda s: (torch.randn(6, 64 * 80, 128, generator=g) * s).to(dev)
q, k, v, go = mk(0.6), mk(0.8), mk(1.0), mk(1.0)
out = hip.window_attention(q, k, v, 64, 80, 2, True)
for name, fn in (("hip", lambda: hip.window_attention_backward(q, k, v, out, go, 64, 80, 2, True)),
                 ("torch", lambda: torch.autograd.grad(ag._window_attention_torch(*(t.detach().requires_grad_(True) for t in (q, k, v)), 64, 80, 2, True), None, go) if False else None)):
    if name == "torch":
        def fn():
            qq, kk, vv = (t.detach().requires_grad_(True) for t in (q, k, v))
            o = ag._window_attention_torch(qq, kk, vv, 64, 80, 2, True)
            return torch.autograd.grad(o, (qq, kk, vv), go)
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"attention backward alone ({name}): {e0.elapsed_time(e1) / 5:.3f} ms per call", flush=True)
