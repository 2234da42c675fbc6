"""Debug experiment: per-phase s_memtime stamps of the ping-pong decoder (decoder_pp_kernel; needs the -DMNERF_TIMELINE
build: tools/exp/build_timeline.sh -> matchnerf_amd/libmnerf_hip_tl.so, selected with MNERF_LIB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

NP = 28
tl = torch.zeros(32 * 4 * 8 * NP * 2, dtype=torch.int64, device="cuda")
os.environ["MNERF_TIMELINE_PTR"] = str(tl.data_ptr())
import bench  # noqa: E402

opt, model, _ = bench.build_model(torch.device("cuda:0"))
_, batch = bench.make_batch(torch.device("cuda:0"), 0)
with torch.no_grad():
    model(batch, mode="test")
    tl.zero_()
    model(batch, mode="test")
torch.cuda.synchronize()
t = tl.cpu().numpy().reshape(32, 4, 8, NP, 2).astype(np.float64)  # [wg, tile, wave, phase, (work end | barrier passed)]
names = ["V0 inputs", "M0 FiLM", "V1 posenc", "M1 L0", "V2", "M2 L1", "V3", "M3 L2", "V4", "M4 L3", "V5", "M5 L4", "V6 posenc",
         "M6 L5e", "V7 split", "M7 L5h", "V8", "M8 alpha", "V9 -", "M9 feature", "V10", "M10 views", "V11", "M11 rgb", "T1 qkv",
         "T2 attention", "T3 fc/LN/sigma", "T4 composite"]
# phase p of a wave starts when it passed the barrier of phase p-1 (tile 1..2: steady state)
tiles = [1, 2]
start = np.concatenate([t[:, :, :, -1:, 1][:, [k - 1 for k in tiles]], t[:, tiles][:, :, :, :-1, 1]], axis=3)  # [wg, tile, wave, phase]
work = t[:, tiles][..., 0] - start       # cycles of work
wait = t[:, tiles][..., 1] - t[:, tiles][..., 0]  # cycles parked at the barrier
ok = (t[:, tiles][:, :, :4, :, 0] > 0).all(axis=(2, 3))  # (team A's stamps: the B-idle experiment has no others)
print("per phase: cycles of work / cycles at the barrier, team A (waves 0-3) | team B (waves 4-7); mean over workgroups")
tot = 0.0
for p in range(NP):
    wa, wb = work[ok][:, :4, p].mean(), work[ok][:, 4:, p].mean()
    ba, bb = wait[ok][:, :4, p].mean(), wait[ok][:, 4:, p].mean()
    step = (work[ok][:, :4, p] + wait[ok][:, :4, p]).mean()
    tot += step
    print(f"  {p:2d} {names[p]:16s} A {wa:7.0f} + {ba:6.0f}   B {wa if False else wb:7.0f} + {bb:6.0f}   (step {step:7.0f})")
print(f"  one tile (256 samples): {tot:.0f} cycles per team")
