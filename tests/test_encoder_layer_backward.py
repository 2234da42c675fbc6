"""Backward of a whole GMFlow transformer layer in HIP (autograd.transformer_layer: mnerf_encoder_layer_backward +
mnerf_window_attention_backward + mnerf_qkv_backward) against float64 autograd through the ORACLE's transformer layer
(oracle/matchnerf_oracle.py: transformer_layer, pinned to the reference by the goldens)."""
import pytest
import torch

from oracle import matchnerf_oracle as O

pytestmark = pytest.mark.gpu


def _layer(no_ffn, seed):
    from matchnerf_amd.gmflow import TransformerLayer
    torch.manual_seed(seed)
    layer = TransformerLayer(128, no_ffn=no_ffn).cuda()
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():  # LayerNorm parameters off their 1 / 0 defaults, so that their gradients' paths are exercised
        for n, p in layer.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn(p.shape, generator=gen).cuda())
    return layer


@pytest.mark.parametrize("no_ffn,b,h,w,splits,shifted,cross", [
    (True, 2, 16, 24, 2, False, False),    # self-attention layer: merge + norm1 only
    (False, 2, 16, 24, 2, True, True),     # cross-attention + FFN, shifted windows
    (False, 4, 8, 12, 1, False, True),     # one window = the whole map
    (False, 6, 32, 40, 2, True, True),     # 6 sequences, 320-token windows
])
@pytest.mark.parametrize("enc_save", ["1", "0"])
def test_transformer_layer_gradients_match_float64_autograd(no_ffn, b, h, w, splits, shifted, cross, enc_save, monkeypatch):
    """enc_save: "1" (default) - an FFN layer's training forward keeps mlp.0's pre-GELU output and mlp.2's output
    (mnerf_encoder_block_save -> mnerf_encoder_layer_backward_saved); "0" - the backward re-evaluates them"""
    from matchnerf_amd import autograd as ag
    monkeypatch.setenv("MNERF_ENC_SAVE", enc_save)
    layer = _layer(no_ffn, seed=h + w + splits)
    gen = torch.Generator().manual_seed(7 * h + w)
    n = h * w
    src = torch.randn(b, n, 128, generator=gen)
    tgt = torch.randn(b, n, 128, generator=gen) if cross else None
    g = torch.randn(b, n, 128, generator=gen)
    s_gpu = src.cuda().requires_grad_(True)
    t_gpu = tgt.cuda().requires_grad_(True) if cross else s_gpu
    out = ag.transformer_layer(layer, s_gpu, t_gpu, h, w, splits, shifted)
    (out * g.cuda()).sum().backward()

    sd64 = {"l." + k: p.detach().double().cpu().requires_grad_(True) for k, p in layer.named_parameters()}
    s64 = src.double().requires_grad_(True)
    t64 = tgt.double().requires_grad_(True) if cross else s64
    o64 = O.transformer_layer(sd64, "l.", s64, t64, h, w, splits, shifted, not no_ffn)
    assert float((out.detach().cpu().double() - o64.detach()).abs().max()) < 5e-5
    (o64 * g.double()).sum().backward()
    worst = {"source": float((s_gpu.grad.cpu().double() - s64.grad).abs().max() / s64.grad.abs().max())}
    if cross:
        worst["target"] = float((t_gpu.grad.cpu().double() - t64.grad).abs().max() / t64.grad.abs().max())
    for k, p in layer.named_parameters():
        ref = sd64["l." + k].grad
        assert p.grad is not None and ref is not None, k
        worst[k] = float((p.grad.cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-30))
    print({k: f"{v:.1e}" for k, v in worst.items()})
    assert all(v < 5e-5 for v in worst.values()), worst


def test_encoder_block_save_is_the_inference_kernel_plus_two_tensors():
    """mnerf_encoder_block_save: `out` bit-identical to mnerf_encoder_block's; z1 = mlp.0(cat[source, norm1(merge(attn))]) and
    m2 = mlp.2(GELU(z1)) against float64 (ragged token count: the last workgroup's dead lanes write nothing)."""
    from matchnerf_amd import hip
    layer = _layer(False, seed=3)
    gen = torch.Generator().manual_seed(11)
    n = 1000
    attn, source = torch.randn(n, 128, generator=gen).cuda(), torch.randn(n, 128, generator=gen).cuda()
    bws, ln, bews = layer._packed_block(attn.device)
    out0 = hip.encoder_block(attn, source, bws, ln, True, bews)
    out1, m1, z1, m2 = hip.encoder_block(attn, source, bws, ln, True, bews, save=True)
    assert torch.equal(out0, out1)
    a64, s64 = attn.double().cpu(), source.double().cpu()
    P = {k: v.detach().double().cpu() for k, v in layer.named_parameters()}
    assert float((m1.cpu().double() - a64 @ P["merge.weight"].t()).abs().max()) < 5e-6 * float((a64 @ P["merge.weight"].t()).abs().max())
    msg = torch.nn.functional.layer_norm(a64 @ P["merge.weight"].t(), (128,), P["norm1.weight"], P["norm1.bias"])
    z_ref = torch.cat([s64, msg], -1) @ P["mlp.0.weight"].t()
    m_ref = torch.nn.functional.gelu(z_ref) @ P["mlp.2.weight"].t()
    assert float((z1.cpu().double() - z_ref).abs().max() / z_ref.abs().max()) < 5e-6
    assert float((m2.cpu().double() - m_ref).abs().max() / m_ref.abs().max()) < 5e-6
    # a layer without an FFN keeps merge's output only
    layer2 = _layer(True, seed=4)
    bws2, ln2, bews2 = layer2._packed_block(attn.device)
    o0 = hip.encoder_block(attn, source, bws2, ln2, False, bews2)
    o1, m1b, z1b, m2b = hip.encoder_block(attn, source, bws2, ln2, False, bews2, save=True)
    assert torch.equal(o0, o1) and z1b is None and m2b is None
    ref = a64 @ layer2.merge.weight.detach().double().cpu().t()
    assert float((m1b.cpu().double() - ref).abs().max() / ref.abs().max()) < 5e-6


def test_hip_and_torch_backward_of_the_encoder_agree(monkeypatch):
    """the whole FeatureTransformer under autograd: MNERF_ENC_BACKWARD=hip (one autograd node per layer) against =torch (the
    reference's op chain on rocBLAS); same loss, gradients of every transformer parameter within 2e-4 of each other"""
    from matchnerf_amd.gmflow import FeatureTransformer
    torch.manual_seed(3)
    tr = FeatureTransformer(num_layers=2).cuda()
    gen = torch.Generator().manual_seed(11)
    p_n, h, w = 3, 16, 24
    src = torch.randn(2 * p_n, h * w, 128, generator=gen).cuda()
    g = torch.randn(2 * p_n, h * w, 128, generator=gen).cuda()
    res = {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("MNERF_ENC_BACKWARD", mode)
        for p in tr.parameters():
            p.grad = None
        x = src.clone().requires_grad_(True)
        out = tr(x, p_n, h, w, 2, False)
        (out * g).sum().backward()
        res[mode] = (out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in tr.named_parameters()})
    assert float((res["hip"][0] - res["torch"][0]).abs().max()) < 2e-4
    assert float((res["hip"][1] - res["torch"][1]).abs().max() / res["torch"][1].abs().max()) < 2e-4
    for k in res["hip"][2]:
        a, bb = res["hip"][2][k], res["torch"][2][k]
        assert float((a - bb).abs().max() / (bb.abs().max() + 1e-30)) < 2e-4, k
