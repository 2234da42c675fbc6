"""GPU parity of the fused InstanceNorm kernel (mnerf_instance_norm) with torch's op chain and with a float64
evaluation, over the register-cached instantiations and the streaming fallback."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from matchnerf_amd import hip as H
    H.load()
    return H


def _chain(x, res, inner, outer):
    v = F.instance_norm(x)
    if inner:
        v = F.relu(v)
    if res is not None:
        v = v + res
    return F.relu(v) if outer else v


@pytest.mark.parametrize("shape", [(3, 64, 256, 320), (3, 96, 128, 160), (2, 128, 64, 80), (1, 5, 7, 9), (1, 3, 400, 400),
                                   (2, 4, 50, 50)])
def test_instance_norm_matches_torch_and_float64(hip, shape):
    gen = torch.Generator().manual_seed(sum(shape))
    # planes with very different offsets and spreads (the centred variance must not cancel)
    x = torch.randn(shape, generator=gen) * (0.1 + 3 * torch.rand(shape[0], shape[1], 1, 1, generator=gen)) \
        + 20 * torch.randn(shape[0], shape[1], 1, 1, generator=gen)
    res = torch.randn(shape, generator=gen)
    xg, rg = x.cuda(), res.cuda()
    for inner, outer, use_res in ((True, False, False), (False, False, False), (True, True, True)):
        want64 = _chain(x.double(), res.double() if use_res else None, inner, outer)
        got = hip.instance_norm(xg, rg if use_res else None, relu_inner=inner, relu_outer=outer)
        torch_gpu = _chain(xg, rg if use_res else None, inner, outer)
        err = float((got.cpu().double() - want64).abs().max())
        err_torch = float((torch_gpu.cpu().double() - want64).abs().max())
        assert err < 2e-5 and err <= 4 * err_torch + 2e-6, (shape, inner, outer, use_res, err, err_torch)
    # in place; the largest magnitude is handed over exactly
    y = xg.clone()
    reg = hip.absmax_regions(1, "cuda")
    hip.instance_norm(y, relu_inner=True, out=y, out_absmax=reg[0])
    assert torch.equal(y, hip.instance_norm(xg, relu_inner=True))
    assert float(hip.absmax_value(reg[0])) == float(y.max())


def test_instance_norm_argument_checks(hip):
    x = torch.randn(2, 3, 8, 8, device="cuda")
    with pytest.raises(hip.MnerfError):
        hip.instance_norm(x.reshape(6, 64))
    with pytest.raises(hip.MnerfError):
        hip.instance_norm(x, residual=torch.randn(2, 3, 8, 4, device="cuda"))
    assert hip.instance_norm(torch.empty(0, 3, 8, 8, device="cuda")).shape == (0, 3, 8, 8)


def test_backbone_fused_norm_matches_op_chain(hip):
    """the CNN backbone with the fused norm kernels (inference) against the same module's torch op chain"""
    from matchnerf_amd.gmflow import CNNEncoder
    torch.manual_seed(0)
    net = CNNEncoder().cuda()
    x = torch.rand(2, 3, 64, 96, device="cuda")
    with torch.no_grad():
        fused = net(x)
    with torch.enable_grad():
        chain = net(x).detach()
    assert float((fused - chain).abs().max()) < 2e-5 * float(chain.abs().max())
