"""TFLOP/s of the strided GEMM under the backward kernels (mnerf_debug_gemm) on the transformer layers' shapes: split-bf16 vs exact-f32
matrix instruction vs torch (rocBLAS fp32)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from matchnerf_amd import hip

N = 6 * 64 * 80
cases = [("Linear fwd  [N,128] x [128,128]^T", (N, 128, 128), "nt"), ("Linear fwd  [N,256] x [1024,256]^T", (N, 1024, 256), "nt"),
         ("data grad   [N,1024] x [1024,256]", (N, 256, 1024), "nn"), ("data grad   [N,128] x [128,128]", (N, 128, 128), "nn"),
         ("weight grad [N,1024]^T x [N,256]", (1024, 256, N), "tn"), ("weight grad [N,128]^T x [N,128]", (128, 128, N), "tn"),
         ("decoder     [65536,128] x [128,128]^T", (65536, 128, 128), "nt")]
for name, (I, J, K), layout in cases:
    a = torch.randn(I, K, device="cuda")
    b = torch.randn(K, J, device="cuda")
    if layout == "nt":
        b = b.t().contiguous().t()
    if layout == "tn":
        a = a.t().contiguous().t()
    out = torch.zeros(I, J, device="cuda")
    res = []
    for math in ("bf16x6", "f32", "torch"):
        mode = 2 if layout == "tn" else 0
        f = (lambda: torch.matmul(a, b, out=out)) if math == "torch" else (lambda: hip.debug_gemm(a, b, out=out, mode=mode, math=math))
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.append(f"{math} {ms * 1e3:7.1f} us {2 * I * J * K / ms / 1e9:6.1f} TF/s")
    print(f"{name:40s} " + " | ".join(res))
