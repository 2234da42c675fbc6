"""Window-attention kernel time at the bench shape (6 x 64x80 tokens, 2x2 windows), per math path.
usage: python tools/exp/wa_time.py [math ...]   (env MNERF_WA_XCD / MNERF_WA_MIN4 are read at library load)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from matchnerf_amd import hip  # noqa: E402

maths = sys.argv[1:] or ["f16pre", "bf16x6"]
table = {"f16pre": hip.WA_PRESPLIT_F16, "f16x3": hip.WA_SPLIT_F16, "bf16x6": hip.WA_SPLIT_BF16, "f32": hip.WA_EXACT_F32}
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(6, 64 * 80, 128, generator=g).cuda() for _ in range(3))
out = torch.empty_like(q)
for m in maths:
    for shifted in (False, True):
        for _ in range(3):
            hip.window_attention(q, k, v, 64, 80, 2, shifted, out=out, math=table[m])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.window_attention(q, k, v, 64, 80, 2, shifted, out=out, math=table[m])
        e1.record()
        torch.cuda.synchronize()
        print(f"{m:8s} shifted={int(shifted)} xcd={os.environ.get('MNERF_WA_XCD', '1')}  {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per call")
