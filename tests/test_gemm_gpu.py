"""The strided GEMM under the HIP backward kernels (matchnerf_amd/csrc/gemm_f32.hpp) on its own: the split-bf16 form that products
with I, J >= 128 take must be fp32-grade — judged against float64, next to the exact-f32 matrix instruction and torch's fp32 matmul
on the same operands — for every operand layout the backward passes use (Linear forward / data gradient / weight gradient), ragged
sizes, the three accumulation modes and operands of very different magnitude (the reason it is bf16 x 3 and not fp16 x 2: no
scaling, fp32's exponent range)."""
import pytest
import torch

from matchnerf_amd import hip

pytestmark = pytest.mark.gpu


def rel_err(c, ref):
    return float((c.double() - ref).abs().max() / ref.abs().max())


def operands(I, J, K, layout, seed, spread=0.0):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(I, K, generator=g)
    b = torch.randn(K, J, generator=g)
    if spread:  # rows of A / columns of B of very different magnitude
        a = a * torch.exp2(torch.randint(-int(spread), int(spread) + 1, (I, 1), generator=g).float())
        b = b * torch.exp2(torch.randint(-int(spread), int(spread) + 1, (1, J), generator=g).float())
    a, b = a.cuda(), b.cuda()
    if layout == "nt":    # Linear forward: x [N,K] row-major, w [M,K] row-major -> b is a transposed view
        return a, b.t().contiguous().t()
    if layout == "nn":    # data gradient: dy [N,M] @ w [M,K]
        return a, b
    if layout == "tn":    # weight gradient: dy^T @ x, the reduction runs over the rows of both
        return a.t().contiguous().t(), b
    if layout == "tt":
        return a.t().contiguous().t(), b.t().contiguous().t()
    raise ValueError(layout)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt"])
@pytest.mark.parametrize("shape", [(256, 128, 512), (300, 130, 77), (128, 640, 1281), (1280, 128, 128), (4096, 2048, 72), (4000, 2100, 40)])  # the last two: 128 x 128 tiles
def test_split_bf16_gemm_is_fp32_grade(layout, shape):
    I, J, K = shape
    a, b = operands(I, J, K, layout, seed=I + K)
    ref = a.double() @ b.double()
    c6 = hip.debug_gemm(a, b, math="bf16x6")
    c32 = hip.debug_gemm(a, b, math="f32")
    e6, e32, et = rel_err(c6, ref), rel_err(c32, ref), rel_err(a @ b, ref)
    assert e32 < 2e-6 and et < 2e-6
    assert e6 < 2e-6 and e6 < 4 * max(e32, et) + 1e-7, (e6, e32, et)


def test_magnitudes_spread_over_forty_binades():
    """per-row / per-column scales 2^-20 .. 2^20: every output element is judged against ITS OWN magnitude scale (row scale x
    column scale x sqrt K) — a global-scale split (fp16 with one gain) fails this, three bf16 terms do not"""
    I, J, K = 256, 256, 384
    a, b = operands(I, J, K, "nt", seed=5, spread=20)
    ref = a.double() @ b.double()
    scale = a.double().abs().amax(1, keepdim=True) * b.double().abs().amax(0, keepdim=True) * K ** 0.5
    c6 = hip.debug_gemm(a, b, math="bf16x6")
    c32 = hip.debug_gemm(a, b, math="f32")
    e6 = float(((c6.double() - ref).abs() / scale).max())
    e32 = float(((c32.double() - ref).abs() / scale).max())
    assert e32 < 1e-6 and e6 < 1e-6 and e6 < 4 * e32 + 1e-7, (e6, e32)


def test_bias_accumulate_and_split_k_modes():
    I, J, K = 256, 192, 4096
    a, b = operands(I, J, K, "tn", seed=9)
    bias = torch.randn(J, device="cuda")
    ref = a.double() @ b.double()
    e32 = rel_err(hip.debug_gemm(a, b, math="f32"), ref)  # fp32 accumulation over K = 4 096: the yardstick
    assert e32 < 6e-6
    gate = 2 * e32 + 1e-7
    c = hip.debug_gemm(a, b, bias=bias, math="bf16x6")
    assert rel_err(c, ref + bias.double()) < gate
    base = torch.randn(I, J, device="cuda")
    c = hip.debug_gemm(a, b, out=base.clone(), mode=1)
    assert rel_err(c, ref + base.double()) < gate
    c = hip.debug_gemm(a, b, out=base.clone(), mode=2)  # split-K with atomics: the order of the partial sums is not fixed
    assert rel_err(c, ref + base.double()) < gate
    padded = torch.zeros(I, J + 8, device="cuda")       # C with a row stride of its own
    hip.debug_gemm(a, b, out=padded[:, :J])
    assert rel_err(padded[:, :J], ref) < gate and float(padded[:, J:].abs().max()) == 0.0


def test_small_products_keep_the_exact_f32_kernel():
    """below 128 x 128 the 64-tile exact-f32 kernel runs whatever the math argument says: identical bits"""
    a, b = operands(100, 64, 200, "nt", seed=2)
    assert torch.equal(hip.debug_gemm(a, b, math="bf16x6"), hip.debug_gemm(a, b, math="f32"))


@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt"])
@pytest.mark.parametrize("shape", [(256, 128, 512), (300, 130, 77), (128, 640, 1281), (1280, 128, 128), (4096, 2048, 72), (4000, 2100, 40)])
def test_split_fp16_gemm_with_row_gains_is_fp32_grade(layout, shape):
    """round 6: three fp16 products per MAC, one power-of-two gain per A row / B column of the tile (gemm_h3_kernel)"""
    I, J, K = shape
    a, b = operands(I, J, K, layout, seed=I + K + 1)
    ref = a.double() @ b.double()
    c3 = hip.debug_gemm(a, b, math="f16x3")
    e3, e32 = rel_err(c3, ref), rel_err(hip.debug_gemm(a, b, math="f32"), ref)
    assert e3 < 4e-6 and e3 < 8 * e32 + 2e-7, (e3, e32)


def test_split_fp16_gemm_row_and_column_scales_over_forty_binades():
    """the case a one-gain-per-tensor fp16 split fails (test_magnitudes_spread_over_forty_binades): per-row / per-column gains
    make it exact in the scales; every output element against ITS OWN magnitude scale"""
    I, J, K = 256, 256, 384
    a, b = operands(I, J, K, "nt", seed=6, spread=20)
    ref = a.double() @ b.double()
    scale = a.double().abs().amax(1, keepdim=True) * b.double().abs().amax(0, keepdim=True) * K ** 0.5
    e3 = float(((hip.debug_gemm(a, b, math="f16x3").double() - ref).abs() / scale).max())
    e32 = float(((hip.debug_gemm(a, b, math="f32").double() - ref).abs() / scale).max())
    assert e3 < 1e-6 and e3 < 8 * e32 + 1e-8, (e3, e32)


def test_split_fp16_gemm_modes_and_zero_rows():
    I, J, K = 384, 256, 4096
    a, b = operands(I, J, K, "nn", seed=12)
    a[5] = 0.0  # an all-zero row: gain of a zero maximum
    b[:, 7] = 0.0
    ref = a.double() @ b.double()
    bias = torch.randn(J, device="cuda")
    assert rel_err(hip.debug_gemm(a, b, bias=bias, math="f16x3"), ref + bias.double()) < 4e-6
    base = torch.randn(I, J, device="cuda")
    assert rel_err(hip.debug_gemm(a, b, out=base.clone(), mode=1, math="f16x3"), ref + base.double()) < 4e-6
    assert rel_err(hip.debug_gemm(a, b, out=base.clone(), mode=2, math="f16x3"), ref + base.double()) < 4e-6
    c = hip.debug_gemm(a, b, math="f16x3")
    assert float(c[5].abs().max()) == 0.0 and float(c[:, 7].abs().max()) == 0.0
