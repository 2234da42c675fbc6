"""Backward of the backbone / up-sampler convolutions (csrc/conv_backward.hip) against torch's own conv gradients in float64 on the
CPU: every (channels, filter, stride) combination the GMFlow CNN has, odd sizes, borders, bit-reproducibility of the weight gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, dy, stride):
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = torch.nn.functional.conv2d(x64, w64, None, stride, w.shape[2] // 2)
    assert y.shape == dy.shape, (y.shape, dy.shape)
    y.backward(dy.double())
    return x64.grad, w64.grad


CASES = [  # (n, c_in, c_out, h, w, k, stride)
    (2, 64, 64, 20, 40, 3, 1), (1, 64, 96, 21, 37, 3, 2), (2, 96, 96, 12, 24, 3, 1), (1, 96, 128, 16, 24, 3, 2),
    (1, 128, 128, 9, 13, 3, 1), (2, 64, 96, 10, 18, 1, 2), (1, 96, 128, 11, 15, 1, 2), (2, 128, 128, 8, 10, 1, 1),
    (1, 32, 64, 7, 5, 3, 1), (1, 128, 32, 6, 70, 3, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_backward_matches_torch_float64(case):
    from matchnerf_amd import hip
    n, ci, co, h, w, k, s = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    # gradients over many binades, as a real loss produces them
    dy = torch.randn(n, co, ho, wo, generator=g) * 2.0 ** torch.randint(-20, -4, (n, co, 1, 1), generator=g).float()
    dx_ref, dw_ref = _ref(x, wt, dy, s)
    dx = hip.conv2d_backward_data(dy.cuda(), wt.cuda(), h, w, s).cpu().double()
    dw = hip.conv2d_backward_weight(x.cuda(), dy.cuda(), k, s).cpu().double()
    assert (dx - dx_ref).abs().max() <= 2e-6 * dx_ref.abs().max(), case
    assert (dw - dw_ref).abs().max() <= 2e-6 * dw_ref.abs().max(), case
    again = hip.conv2d_backward_weight(x.cuda(), dy.cuda(), k, s).cpu().double()
    assert torch.equal(again, dw)


def test_conv_backward_at_the_backbone_shape():
    """layer1's convolution at the DTU shape (3 x 64 x 256 x 320): the weight gradient's chunked reduction over 768 rows"""
    from matchnerf_amd import hip
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 64, 256, 320, generator=g).cuda()
    wt = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).cuda()
    dy = (torch.randn(3, 64, 256, 320, generator=g) * 1e-4).cuda()
    dx = hip.conv2d_backward_data(dy, wt, 256, 320, 1)
    dw = hip.conv2d_backward_weight(x, dy, 3, 1)
    ref_dx = torch.nn.grad.conv2d_input(x.shape, wt.double(), dy.double(), 1, 1)
    ref_dw = torch.nn.grad.conv2d_weight(x.double(), wt.shape, dy.double(), 1, 1)
    assert (dx.double() - ref_dx).abs().max() <= 2e-6 * ref_dx.abs().max()
    assert (dw.double() - ref_dw).abs().max() <= 5e-6 * ref_dw.abs().max()


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape", [(2, 5, 16, 24), (1, 3, 7, 9), (1, 2, 256, 320)])
def test_instance_norm_backward_matches_autograd_float64(shape, relu):
    from matchnerf_amd import hip
    g = torch.Generator().manual_seed(sum(shape) + int(relu))
    x = torch.randn(*shape, generator=g) * 3.0 + 0.5
    dy = torch.randn(*shape, generator=g) * 1e-3
    x64 = x.double().requires_grad_(True)
    y = torch.nn.functional.instance_norm(x64)
    if relu:
        y = torch.relu(y)
    y.backward(dy.double())
    dx = hip.instance_norm_backward(x.cuda(), dy.cuda(), relu).cpu().double()
    assert (dx - x64.grad).abs().max() <= 1e-5 * x64.grad.abs().max(), (shape, relu)


@pytest.mark.parametrize("case", CASES + [(2, 3, 64, 30, 44, 7, 2), (1, 3, 64, 17, 23, 7, 2)])
def test_conv_forward_f32_matches_torch_float64(case):
    from matchnerf_amd import hip
    n, ci, co, h, w, k, s = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    bias = torch.randn(co, generator=g) if k == 1 else None
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None if bias is None else bias.double(), s, k // 2)
    y = hip.conv2d_forward_f32(x.cuda(), wt.cuda(), None if bias is None else bias.cuda(), s).cpu().double()
    assert y.shape == ref.shape and (y - ref).abs().max() <= 2e-6 * ref.abs().max(), case


@pytest.mark.parametrize("shape", [(2, 30, 44), (1, 17, 23), (3, 64, 80)])
def test_stem_weight_gradient_matches_torch_float64(shape):
    from matchnerf_amd import hip
    n, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, 3, h, w, generator=g)
    dy = torch.randn(n, 64, (h - 1) // 2 + 1, (w - 1) // 2 + 1, generator=g) * 1e-3
    ref = torch.nn.grad.conv2d_weight(x.double(), (64, 3, 7, 7), dy.double(), 2, 3)
    dw = hip.conv_stem_backward_weight(x.cuda(), dy.cuda()).cpu().double()
    assert (dw - ref).abs().max() <= 3e-6 * ref.abs().max(), shape


@pytest.mark.parametrize("case", CASES)
def test_weight_gradient_on_the_16_bit_pipe_matches_torch_float64(case):
    """split-fp16 operands (one gain per tensor from the absmax regions), three products per MAC: 22-bit operands"""
    from matchnerf_amd import hip
    n, ci, co, h, w, k, s = case
    g = torch.Generator().manual_seed(sum(case) + 2)
    x = (torch.randn(n, ci, h, w, generator=g) * 3.0).cuda()
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    dy = (torch.randn(n, co, ho, wo, generator=g) * 1e-4).cuda()
    regs = hip.absmax_regions(2, x.device)
    hip.absmax(x, regs[0]), hip.absmax(dy, regs[1])
    ref = torch.nn.grad.conv2d_weight(x.cpu().double(), (co, ci, k, k), dy.cpu().double(), s, k // 2)
    dw = hip.conv2d_backward_weight(x, dy, k, s, regs[0], regs[1]).cpu().double()
    assert (dw - ref).abs().max() <= 1e-5 * ref.abs().max(), case
    assert torch.equal(hip.conv2d_backward_weight(x, dy, k, s, regs[0], regs[1]).cpu().double(), dw)


@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (1, 2, 1, 4), (1, 1, 64, 80), (1, 2, 6, 1)])
def test_bilinear_upsampling_and_its_adjoint_match_torch(shape):
    from matchnerf_amd import hip
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    add = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], generator=g)
    x64 = x.double().requires_grad_(True)
    ref = F.interpolate(x64, scale_factor=2, mode="bilinear", align_corners=False) + add.double()
    out = hip.upsample_bilinear2x(x.cuda(), add.cuda()).cpu().double()
    assert (out - ref.detach()).abs().max() < 1e-6
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout.double())
    din = hip.upsample_bilinear2x_backward(gout.cuda()).cpu().double()
    assert (din - x64.grad).abs().max() < 1e-6
