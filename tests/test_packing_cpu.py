"""Device-side weight packing (matchnerf_amd/packing.py) is the numpy packers' bit-for-bit twin — checked with torch CPU tensors
(the same torch ops run on the GPU), plus the one-copy refresh of a whole transformer."""
import numpy as np
import pytest
import torch

from matchnerf_amd import cond_nerf as CN, gmflow, packing


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("scale", [1.0, 3.7e-3, 260.0])
def test_fragments_equal_the_numpy_builder(scale):
    w = rnd((100, 77), 1, scale)
    cols = np.arange(80).reshape(5, 2, 8)
    cols = np.where(cols < 77, cols, -1)
    ew = CN.f16_weight_exponent(w.numpy())
    want = CN._fragments_h(w.numpy(), cols, 4, ew)
    got = packing.fragments_h(w, cols, 4, ew)
    assert got.dtype == torch.float16 and tuple(got.shape) == want.shape
    assert np.array_equal(got.numpy().view(np.uint16), want.view(np.uint16))
    assert packing.weight_exponents([w]) == [ew]


def test_zero_and_nonfinite_exponents():
    assert packing.weight_exponents([torch.zeros(4, 4), torch.full((2, 2), float("inf"))]) == [0, 0]
    assert packing.weight_exponents([]) == []


def test_qkv_stream_equals_numpy_packer():
    wq, wk, wv = rnd((128, 128), 2, 0.09), rnd((128, 128), 3, 0.4), rnd((128, 128), 4, 2.0)
    want, ews = gmflow.pack_qkv(wq, wk, wv)
    got, ews_t = packing.pack_qkv(wq, wk, wv)
    assert ews_t == ews and np.array_equal(got.numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("ffn", [False, True])
def test_encoder_block_stream_equals_numpy_packer(ffn):
    m = rnd((128, 128), 5, 0.1)
    w0, w2 = (rnd((1024, 256), 6, 0.05), rnd((128, 1024), 7, 0.03)) if ffn else (None, None)
    want, ews = gmflow.pack_encoder_block(m, w0, w2)
    got, ews_t = packing.pack_encoder_block(m, w0, w2)
    assert ews_t == ews and np.array_equal(got.numpy().view(np.uint32), want.view(np.uint32))


def test_whole_transformer_in_one_gather_equals_the_per_layer_packers():
    torch.manual_seed(11)
    ft = gmflow.FeatureTransformer(num_layers=3)
    with torch.no_grad():
        for i, p in enumerate(ft.parameters()):
            p.mul_(0.3 + 0.17 * i)  # different scale exponents per tensor
    assert ft.refresh_packs("cpu") == 12
    assert ft.refresh_packs("cpu") == 0
    for blk in ft.layers:
        for layer in (blk.self_attn, blk.cross_attn_ffn):
            ws, ews = layer._qkv[1:]
            want, ews_n = gmflow.pack_qkv(layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight)
            assert ews == ews_n and np.array_equal(ws.numpy().view(np.uint32), want.view(np.uint32))
            ws, ln, ews = layer._blk[1:]
            want, ews_n = gmflow.pack_encoder_block(layer.merge.weight, None if layer.no_ffn else layer.mlp[0].weight,
                                                    None if layer.no_ffn else layer.mlp[2].weight)
            assert ews == ews_n and np.array_equal(ws.numpy().view(np.uint32), want.view(np.uint32))
            assert layer.stale_packs(torch.device("cpu")) == []
    layer = ft.layers[1].cross_attn_ffn
    old = layer._qkv[2]
    with torch.no_grad():
        layer.q_proj.weight.mul_(2.0)   # what an optimizer step does to the version counter
    assert layer.stale_packs(torch.device("cpu")) == ["qkv"]
    assert ft.refresh_packs("cpu") == 12
    assert layer._qkv[2][0] == old[0] - 1 and layer._qkv[2][1:] == old[1:]


@pytest.mark.parametrize("legacy,n_views,posenc", [(True, 3, False), (False, 3, True), (True, 5, False)])
def test_decoder_stream_assembled_with_torch_ops_equals_the_numpy_packer(legacy, n_views, posenc):
    """packing.DecoderPacker (what a GPU-resident decoder uses after every optimizer step: no host copy, exponents from frexp
    on the device) against cond_nerf.pack_wstream_h, incl. an all-zero tensor (exponent 0) and the small block."""
    from matchnerf_amd import options
    opt = options.load_options("configs/test.yaml", verbose=False)
    opt.device = "cpu"
    opt.nerf.legacy_coord, opt.n_src_views, opt.decoder.raytrans_posenc = legacy, n_views, posenc
    dec = CN.CondNeRF(opt)
    torch.manual_seed(4)
    with torch.no_grad():
        for i, p in enumerate(dec.parameters()):
            p.copy_(torch.randn_like(p) * (0.02 + 0.3 * (i % 5)))
        dec.feature_linear.bias.zero_()
        dec.pts_linears[2].weight.zero_()
    sd = {"nerf_dec." + k: v for k, v in dec.state_dict().items()}
    want, cond_dim, cond_stride = CN.pack_wstream_h(sd, n_views, list(opt.encoder.cos_n_group), dec.L_3D, legacy)
    ws, small, cs = dec._packed_on_device(64, torch.device("cpu"))
    assert cs == cond_stride and np.array_equal(ws.numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(small.numpy(), CN.pack_small(sd, 64, posenc))


def test_conv_packer_equals_the_numpy_packers():
    """packing.ConvPacker (training: all convolution streams from one gather) against gmflow.pack_conv / pack_conv_stem bit for
    bit: forward streams of every shape the CNN has incl. the stem's padded matrix, and the backward stream = pack_conv of the
    flipped, transposed filter (the data gradient of a stride-1 convolution as a convolution)."""
    import numpy as np
    import torch

    from matchnerf_amd import gmflow, packing
    torch.manual_seed(0)
    convs = [torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False), torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False), torch.nn.Conv2d(64, 96, 3, 2, 1),
             torch.nn.Conv2d(96, 96, 3, 1, 1), torch.nn.Conv2d(96, 128, 1, 2), torch.nn.Conv2d(128, 128, 1, 1),
             torch.nn.Conv2d(128, 128, 3, 1, 1)]
    with torch.no_grad():
        convs[3].weight.mul_(37.0)  # different exponents per tensor
        convs[5].weight.mul_(1e-3)
    out = packing.ConvPacker(convs, "cpu").pack()
    n_bwd = 0
    for c, (f, b, e) in zip(convs, out):
        ws, ew = gmflow.pack_conv_stem(c.weight) if c.in_channels == 3 else gmflow.pack_conv(c.weight)
        assert ew == e and np.array_equal(ws.view(np.uint32), f.numpy().view(np.uint32)), c
        assert (b is not None) == packing.ConvPacker.has_backward_stream(c)
        if b is not None:
            ws2, ew2 = gmflow.pack_conv(c.weight.detach().flip(2, 3).permute(1, 0, 2, 3).contiguous())
            assert ew2 == e and np.array_equal(ws2.view(np.uint32), b.numpy().view(np.uint32)), c
            n_bwd += 1
    assert n_bwd == 4
