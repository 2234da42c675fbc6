"""Split-fp16 implicit-GEMM convolution (mnerf_conv2d): packer on the CPU, kernel parity on the GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from matchnerf_amd import gmflow as G


def _unpack(ws, c_out, k_total, ew):
    """inverse of pack_conv: the implicit-GEMM matrix the fragments encode"""
    halfs = ws.view(np.float16).reshape(k_total // 16, c_out // 32, 2, 64, 8).astype(np.float64)
    w = halfs[:, :, 0] + halfs[:, :, 1]                      # [steps, blocks, 64, 8]
    mat = np.zeros((c_out, k_total))
    lane = np.arange(64)
    for s in range(k_total // 16):
        for m in range(c_out // 32):
            rows = 32 * m + (lane & 31)
            for j in range(8):
                mat[rows, 16 * s + 8 * (lane >> 5) + j] = w[s, m, lane, j]
    return np.ldexp(mat, -ew)


@pytest.mark.parametrize("c_in,c_out,k", [(64, 64, 3), (64, 96, 1), (96, 128, 3), (128, 128, 3)])
def test_pack_conv_encodes_the_tap_major_matrix(c_in, c_out, k):
    rng = np.random.default_rng(c_in + c_out + k)
    w = (rng.standard_normal((c_out, c_in, k, k)) * 0.1).astype(np.float32)
    ws, ew = G.pack_conv(w)
    from matchnerf_amd import hip
    assert ws.size == c_out // 32 * (k * k * c_in // 16) * 512
    mat = _unpack(ws, c_out, k * k * c_in, ew)
    assert np.abs(mat - w.transpose(0, 2, 3, 1).reshape(c_out, -1)).max() < 2.0 ** -20 * np.abs(w).max()
    # the matrix applied to an im2col of the input (tap = ky * k + kx, then channel) is the convolution
    x = rng.standard_normal((1, c_in, 6, 7)).astype(np.float32)
    xp = np.pad(x, ((0, 0), (0, 0), (k // 2, k // 2), (k // 2, k // 2)))
    cols = np.stack([xp[0, :, ky:ky + 6, kx:kx + 7] for ky in range(k) for kx in range(k)], 0)  # [taps, c, 6, 7]
    y = (mat @ cols.reshape(k * k * c_in, -1)).reshape(c_out, 6, 7)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=k // 2)[0].numpy()
    assert np.abs(y - ref).max() < 1e-5


@pytest.fixture(scope="module")
def hip():
    from matchnerf_amd import hip as H
    H.load()
    return H


CASES = [  # n, c_in, c_out, k, stride, h, w, channels_last, upsample2x, leaky, bias
    (3, 64, 64, 3, 1, 64, 80, False, False, 1.0, False),
    (2, 64, 96, 3, 2, 64, 80, False, False, 1.0, False),
    (2, 64, 96, 1, 2, 64, 80, False, False, 1.0, True),
    (2, 96, 96, 3, 1, 32, 40, False, False, 1.0, False),
    (2, 96, 128, 3, 2, 33, 41, False, False, 1.0, False),   # odd sizes: ragged last tile, odd stride-2 geometry
    (1, 128, 128, 1, 1, 8, 10, False, False, 1.0, True),
    (2, 128, 128, 3, 1, 16, 20, True, False, 1.0, True),     # channel-last tokens in
    (2, 128, 128, 3, 1, 16, 20, True, True, 0.2, True),      # ... through a nearest 2x up-sampling, LeakyReLU epilogue
    (1, 128, 128, 3, 1, 9, 7, False, True, 0.2, True),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_conv2d_matches_float64(hip, case):
    n, ci, co, k, s, h, w, cl, up, leaky, use_bias = case
    gen = torch.Generator().manual_seed(ci * 7 + co + k + h)
    x = torch.randn(n, ci, h, w, generator=gen) * (0.5 + 4 * torch.rand(1, ci, 1, 1, generator=gen))
    x[0, 0, 0, 0] = 37.0                                            # one spike: sets the tensor's operand scale
    wt = torch.randn(co, ci, k, k, generator=gen) * (1.0 / np.sqrt(ci * k * k))
    bias = torch.randn(co, generator=gen) if use_bias else None
    ws, ew = G.pack_conv(wt)
    xin = x.permute(0, 2, 3, 1).contiguous() if cl else x
    scal = hip.absmax_regions(2, "cuda")
    hip.absmax(xin.cuda(), scal[0])
    assert float(hip.absmax_value(scal[0])) == float(x.abs().max())
    got = hip.conv2d(xin.cuda(), torch.from_numpy(ws).cuda(), bias.cuda() if use_bias else None, ci, co, k, s, ew,
                     scal[0], leaky=leaky, channels_last=cl, upsample2x=up, out_absmax=scal[1])
    xr = x.double()
    if up:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    want = F.conv2d(xr, wt.double(), bias.double() if use_bias else None, stride=s, padding=k // 2)
    if leaky != 1.0:
        want = F.leaky_relu(want, leaky)
    assert got.shape == want.shape
    err = float((got.cpu().double() - want).abs().max())
    ref32 = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x, wt, bias, stride=s, padding=k // 2)
    if leaky != 1.0:
        ref32 = F.leaky_relu(ref32, leaky)
    err32 = float((ref32.double() - want).abs().max())              # what an fp32 evaluation (CPU) achieves
    print(f"\nconv {case}: |err| {err:.2e} (fp32 CPU evaluation {err32:.2e}), max|want| {float(want.abs().max()):.2e}")
    assert err < 4 * err32 + 1e-6 * float(want.abs().max()), (case, err, err32)
    # and an absolute gate that does not move with torch: observed <= 1.0e-6 of the largest output over all cases (MI355X)
    assert err < 3e-6 * float(want.abs().max()), (case, err)
    assert float(hip.absmax_value(scal[1])) == float(got.abs().max())


@pytest.mark.gpu
def test_conv2d_adds_bilinear_upsampling(hip):
    """superres.py:37: right = F.interpolate(right, 2x bilinear) + conv(left), the up-sampling and add as the epilogue"""
    gen = torch.Generator().manual_seed(5)
    left = torch.randn(2, 128, 12, 20, generator=gen)
    right = torch.randn(2, 128, 6, 10, generator=gen) * 3
    wt = torch.randn(128, 128, 3, 3, generator=gen) * 0.03
    bias = torch.randn(128, generator=gen)
    ws, ew = G.pack_conv(wt)
    scal = hip.absmax_regions(1, "cuda")
    hip.absmax(left.cuda(), scal[0])
    got = hip.conv2d(left.cuda(), torch.from_numpy(ws).cuda(), bias.cuda(), 128, 128, 3, 1, ew, scal[0], add_bilinear2x=right.cuda())
    want = F.interpolate(right.double(), scale_factor=2, mode="bilinear", align_corners=False) \
        + F.conv2d(left.double(), wt.double(), bias.double(), padding=1)
    assert float((got.cpu().double() - want).abs().max()) < 3e-6 * float(want.abs().max())


@pytest.mark.gpu
def test_conv2d_channel_last_tokens_with_position_tile(hip):
    """the backbone's last convolution writes the transformer's tokens and adds the window position tile"""
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(3, 128, 10, 12, generator=gen)
    wt = torch.randn(128, 128, 1, 1, generator=gen) * 0.1
    bias = torch.randn(128, generator=gen)
    tile = torch.randn(120, 128, generator=gen)
    ws, ew = G.pack_conv(wt)
    scal = hip.absmax_regions(2, "cuda")
    hip.absmax(x.cuda(), scal[0])
    got = hip.conv2d(x.cuda(), torch.from_numpy(ws).cuda(), bias.cuda(), 128, 128, 1, 1, ew, scal[0],
                     out_layout=hip.CONV_OUT_CHANNEL_LAST, add_channel_last=tile.cuda(), out_absmax=scal[1])
    want = F.conv2d(x.double(), wt.double(), bias.double()).permute(0, 2, 3, 1) + tile.double().reshape(10, 12, 128)
    assert got.shape == (3, 10, 12, 128)
    assert float((got.cpu().double() - want).abs().max()) < 3e-6 * float(want.abs().max())
    assert float(hip.absmax_value(scal[1])) == float(got.abs().max())


@pytest.mark.gpu
def test_upsampler_fused_path_matches_op_chain(hip):
    from matchnerf_amd.gmflow import UpSampler
    torch.manual_seed(1)
    net = UpSampler().cuda()
    tok = torch.randn(2, 16, 24, 128, device="cuda")
    with torch.no_grad():
        fused = net.forward_tokens(tok)
        pm = net.forward_tokens(tok, pair_major=True)
        chain = net(tok.permute(0, 3, 1, 2))
    assert float((fused - chain).abs().max()) < 1e-5 * float(chain.abs().max())
    # the cost volume's layout, written by the last convolution itself: [pairs, side, H, W, C]
    assert torch.equal(pm, torch.stack([fused[:1], fused[1:]], 1).permute(0, 1, 3, 4, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 3, 64, 96), (1, 3, 37, 51), (2, 3, 128, 160)])
def test_conv_stem_matches_float64(hip, shape):
    """the 7x7 stride-2 stem (incl. odd sizes: ragged tiles, taps leaving the image on every side)"""
    gen = torch.Generator().manual_seed(shape[2])
    x = (torch.rand(shape, generator=gen) - 0.45) / 0.225
    wt = torch.randn(64, 3, 7, 7, generator=gen) * 0.08
    ws, ew = G.pack_conv_stem(wt)
    reg = hip.absmax_regions(1, "cuda")
    hip.absmax(x.cuda(), reg[0])
    got = hip.conv_stem(x.cuda(), torch.from_numpy(ws).cuda(), ew, reg[0])
    want = F.conv2d(x.double(), wt.double(), stride=2, padding=3)
    assert got.shape == want.shape
    err = float((got.cpu().double() - want).abs().max())
    err32 = float((F.conv2d(x, wt, stride=2, padding=3).double() - want).abs().max())
    assert err < 4 * err32 + 1e-6 * float(want.abs().max()), (err, err32)


@pytest.mark.gpu
def test_conv2d_argument_checks(hip):
    ws = torch.zeros(512 * 4, device="cuda")
    s = torch.ones(hip.ABSMAX_FLOATS, device="cuda")
    with pytest.raises(hip.MnerfError):  # 3 input channels are not built
        hip.conv2d(torch.zeros(1, 3, 8, 8, device="cuda"), ws, None, 3, 64, 3, 1, 0, s)
    with pytest.raises(hip.MnerfError):  # stream size
        hip.conv2d(torch.zeros(1, 64, 8, 8, device="cuda"), ws, None, 64, 64, 3, 1, 0, s)
    with pytest.raises(hip.MnerfError):  # missing operand scale
        hip.conv2d(torch.zeros(1, 32, 8, 8, device="cuda"), torch.zeros(512 * 4, device="cuda"), None, 32, 64, 1, 1, 0, None)


def _emulated_conv(x, wt, stride, up=False):
    """numpy restatement of csrc/conv.hip's data path on the CPU: packed fragments (pack_conv), one power-of-two
    operand gain for the whole input tensor, operands split into fp16 hi + lo, products hi.lo + lo.hi + hi.hi in wide
    accumulation, scale folded back.  x [N,C,H,W] fp32, wt [O,C,k,k] -> [N,O,Ho,Wo] float64."""
    from matchnerf_amd import cond_nerf as CN
    n, c, h, w = x.shape
    o, _, k, _ = wt.shape
    ws, ew = G.pack_conv(wt)
    frag = ws.view(np.float16).reshape(k * k * c // 16, o // 32, 2, 64, 8).astype(np.float64)   # [step][block][hi|lo][lane][j]
    xs = np.repeat(np.repeat(x, 2, 2), 2, 3) if up else x
    pad = k // 2
    xp = np.pad(xs, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    he, we = xs.shape[2], xs.shape[3]
    ho, wo = (he + 2 * pad - k) // stride + 1, (we + 2 * pad - k) // stride + 1
    cols = np.stack([xp[:, :, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride]
                     for ky in range(k) for kx in range(k)], 1)                                  # [N, taps, C, ho, wo]
    ops = cols.reshape(n, k * k * c, ho * wo).astype(np.float32)                                 # K index = tap * C + c
    eg = (CN.F16_TARGET_EXP + 1) - np.frexp(np.float32(np.abs(x).max()))[1]
    scaled = (ops * np.ldexp(np.float32(1), eg)).astype(np.float32)
    hi = scaled.astype(np.float16)
    lo = (scaled - hi.astype(np.float32)).astype(np.float16)
    hi, lo = hi.astype(np.float64), lo.astype(np.float64)
    lane = np.arange(64)
    w_hi = np.zeros((o, k * k * c))
    w_lo = np.zeros_like(w_hi)
    for s in range(frag.shape[0]):                                                               # fragments -> matrices
        for m in range(o // 32):
            for j in range(8):
                w_hi[32 * m + (lane & 31), 16 * s + 8 * (lane >> 5) + j] = frag[s, m, 0, lane, j]
                w_lo[32 * m + (lane & 31), 16 * s + 8 * (lane >> 5) + j] = frag[s, m, 1, lane, j]
    y = w_hi @ lo + w_lo @ hi + w_hi @ hi                                                        # [N, O, ho*wo] (broadcast matmul)
    return (y * np.ldexp(1.0, -(ew + int(eg)))).reshape(n, o, ho, wo)


@pytest.mark.parametrize("c_in,c_out,k,stride,up", [(64, 64, 3, 1, False), (64, 96, 1, 2, False), (96, 128, 3, 2, False),
                                                    (128, 128, 3, 1, True)])
def test_emulated_conv_data_path_matches_float64(c_in, c_out, k, stride, up):
    """CPU: packer + operand scaling + split-fp16 product rule of the convolution kernel, without a GPU"""
    rng = np.random.default_rng(c_in + c_out + k + stride)
    x = (rng.standard_normal((2, c_in, 7, 9)) * (0.2 + 3 * rng.random((1, c_in, 1, 1)))).astype(np.float32)
    x[0, 0, 0, 0] = 41.0                                                                         # sets the tensor's gain
    wt = (rng.standard_normal((c_out, c_in, k, k)) / np.sqrt(c_in * k * k)).astype(np.float32)
    got = _emulated_conv(x, wt, stride, up)
    xr = torch.from_numpy(x).double()
    if up:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    want = F.conv2d(xr, torch.from_numpy(wt).double(), stride=stride, padding=k // 2).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-6 * np.abs(want).max()
