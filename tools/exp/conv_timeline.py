"""Per-phase timeline of conv_kernel (a -DCONV_TIMELINE build of conv.hip: tools/exp/build_variant.sh tl "-DCONV_TIMELINE" conv.hip,
MNERF_LIB=.../libmnerf_hip_tl.so): s_memtime stamps of wave 0 of four workgroups at the phase boundaries of every iteration.
usage: conv_timeline.py [shape index of conv_time.py's list]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from matchnerf_amd import gmflow as G, hip  # noqa: E402

shapes = [(3, 64, 64, 3, 1, 256, 320), (3, 96, 96, 3, 1, 128, 160), (3, 128, 128, 3, 1, 64, 80), (6, 128, 128, 3, 1, 128, 160)]
n, ci, co, k, s, h, w = shapes[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
x = torch.randn(n, ci, h, w, device="cuda")
ws, ew = G.pack_conv(torch.randn(co, ci, k, k) * 0.05)
ws = torch.from_numpy(ws).cuda()
reg = hip.absmax_regions(1, "cuda")
hip.absmax(x, reg[0])
tl = torch.zeros(4 * 64 * 6, dtype=torch.int64, device="cuda")
y = None
for _ in range(3):
    y = hip.conv2d(x, ws, None, ci, co, k, s, ew, reg[0], out=y)
torch.cuda.synchronize()
os.environ["MNERF_CONV_TL"] = hex(tl.data_ptr())
hip.conv2d(x, ws, None, ci, co, k, s, ew, reg[0], out=y)
torch.cuda.synchronize()
t = tl.cpu().reshape(4, 64, 6)
n_iter = 9 * ci // 32 if k == 3 else ci // 32
names = ["operand wait + split", "request + load issue", "matrix instructions", "to next top", "segment wait", "barrier"]
for slot in range(4):
    a = t[slot, :n_iter]
    if int(a[0, 0]) == 0:
        continue
    t0 = int(a[0, 0])
    tot = int(a[n_iter - 1, 3]) - t0
    d = {"operand wait + split": 0, "request + load issue": 0, "matrix instructions": 0, "segment wait": 0, "barrier": 0, "between": 0}
    rows = []
    for i in range(n_iter):
        p = [int(v) for v in a[i]]
        d["operand wait + split"] += p[1] - p[0]
        d["request + load issue"] += p[2] - p[1]
        d["matrix instructions"] += p[3] - p[2]
        end = p[3]
        if p[4]:
            d["segment wait"] += p[4] - p[3]
            d["barrier"] += p[5] - p[4]
            end = p[5]
        if i + 1 < n_iter:
            d["between"] += int(a[i + 1, 0]) - end
        rows.append((p[1] - p[0], p[2] - p[1], p[3] - p[2], (p[4] - p[3]) if p[4] else 0, (p[5] - p[4]) if p[4] else 0))
    print(f"workgroup slot {slot}: {n_iter} iterations, {tot} ticks from first top to last matrix run; " + ", ".join(f"{kk} {vv} ({100 * vv / tot:.0f} %)" for kk, vv in d.items()))
    print("   first 9 iterations (wait+split, issue, matrix, segment wait, barrier):", rows[:9])
