"""Which memory format do MIOpen's fp32 convolutions of the encoder prefer on this box?  Times every convolution shape
of the backbone (3 x 512x640 input) and of the up-sampler (6 x 128 x 64x80) in NCHW and in channels_last."""
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
shapes = [  # (n, cin, cout, k, stride, h, w, count)
    (3, 3, 64, 7, 2, 512, 640, 1),
    (3, 64, 64, 3, 1, 256, 320, 4),
    (3, 64, 96, 3, 2, 256, 320, 1), (3, 64, 96, 1, 2, 256, 320, 1),
    (3, 96, 96, 3, 1, 128, 160, 3),
    (3, 96, 128, 3, 2, 128, 160, 1), (3, 96, 128, 1, 2, 128, 160, 1),
    (3, 128, 128, 3, 1, 64, 80, 3),
    (3, 128, 128, 1, 1, 64, 80, 1),
    (6, 128, 128, 3, 1, 64, 80, 1),
    (6, 128, 128, 3, 1, 128, 160, 2),
]
tot = {"nchw": 0.0, "cl": 0.0}
for (n, ci, co, k, s, h, w, cnt) in shapes:
    x = torch.randn(n, ci, h, w, device="cuda")
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    res = {}
    for name, fmt in (("nchw", torch.contiguous_format), ("cl", torch.channels_last)):
        xx, ww = x.contiguous(memory_format=fmt), wt.contiguous(memory_format=fmt)
        for _ in range(3):
            y = F.conv2d(xx, ww, None, s, k // 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = F.conv2d(xx, ww, None, s, k // 2)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 10 * 1e3
        tot[name] += res[name] * cnt
    print(f"n{n} {ci:3d}->{co:3d} k{k} s{s} {h}x{w} x{cnt}: nchw {res['nchw']:7.1f} us   channels_last {res['cl']:7.1f} us   (out cl: {y.is_contiguous(memory_format=torch.channels_last)})")
print("encoder total: nchw %.0f us, channels_last %.0f us" % (tot["nchw"], tot["cl"]))
