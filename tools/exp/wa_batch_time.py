"""Window attention (pre-split fp16) at 64x80 / 2x2 windows for several batch sizes: does a second workgroup per CU pay?
usage: python tools/exp/wa_batch_time.py   (MNERF_WA_SPLIT is read at library load)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from matchnerf_amd import hip  # noqa: E402

g = torch.Generator().manual_seed(0)
for b in (3, 6, 9, 12, 24, 48):
    q, k, v = (torch.randn(b, 64 * 80, 128, generator=g).cuda() for _ in range(3))
    out = torch.empty_like(q)
    for _ in range(3):
        hip.window_attention(q, k, v, 64, 80, 2, False, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.window_attention(q, k, v, 64, 80, 2, False, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"batch {b:3d} ({b * 40:5d} query blocks) split={os.environ.get('MNERF_WA_SPLIT', '1')}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per call incl. pre-pass")
