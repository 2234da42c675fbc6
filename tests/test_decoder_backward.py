"""K3+K4 backward (mnerf_decoder_backward) against float64 autograd through the ORACLE's decoder (oracle/matchnerf_oracle.py:
decoder + ray_attention, the restatement of cond_nerf.py:52-100 / ray_transformer.py:29-79 that the goldens pin to the
reference; it is dtype-generic and differentiable)."""
import pytest
import torch

from helpers import golden_case
from oracle import matchnerf_oracle as O
from test_model_gpu import build_model

pytestmark = pytest.mark.gpu


def _case(name, n_rays, n_samples, seed, **decoder_opts):
    from matchnerf_amd import hip
    g, cfg, sd, _ = golden_case(name)
    opt, model = build_model(g["meta"])
    for k, v in decoder_opts.items():
        opt.decoder[k] = v
    dec = model.nerf_dec
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():  # the goldens' zero biases and unit LayerNorm would hide their own gradients' paths
        for n, p in dec.named_parameters():
            if n.endswith("bias") or "layer_norm" in n:
                p.add_(0.1 * torch.randn(p.shape, generator=gen).to(p.device))
    v = cfg.n_src_views
    dc = sum(opt.encoder.cos_n_group) + 4 * v
    stride = (dc + 1 + 7) // 8 * 8
    n = n_rays * n_samples
    x = torch.rand(n, 3, generator=gen) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1)
    cond = torch.zeros(n, stride)
    cond[:, :dc - v] = torch.randn(n, dc - v, generator=gen) * 0.5
    cond[:, dc - v:dc] = (torch.rand(n, v, generator=gen) > 0.35).float()   # rows with 0, 1 and more visible views
    cond[:, dc] = 1.0
    g_rgb = torch.randn(n, 3, generator=gen)
    g_sig = torch.randn(n_rays, n_samples, generator=gen)
    table = None
    if opt.decoder.raytrans_posenc:
        from matchnerf_amd.cond_nerf import raytrans_table
        table = torch.from_numpy(raytrans_table(n_samples))
    params = {k: p.detach() for k, p in dec.named_parameters()}
    g_cond, grads = hip.decoder_backward(opt, params, v, x.cuda(), dirs.cuda(), cond.cuda(), stride, g_rgb.cuda(), g_sig.cuda(),
                                         raytrans_table=table)
    torch.cuda.synchronize()

    for k, val in decoder_opts.items():  # the oracle's switches follow the module's options
        assert hasattr(cfg, k), k
        setattr(cfg, k, val)
    sd64 = {"nerf_dec." + k: p.detach().double().cpu().clone().requires_grad_(True) for k, p in dec.named_parameters()}
    cond_ref = cond[:, :dc].double().reshape(n_rays, n_samples, dc).clone().requires_grad_(True)
    rgb_s, sigma = O.decoder(cfg, sd64, x.double().reshape(n_rays, n_samples, 3), dirs.double(), cond_ref, cond_ref[..., -v:])
    (rgb_s * g_rgb.double().reshape(n_rays, n_samples, 3)).sum().add((sigma * g_sig.double()).sum()).backward()
    worst = {}
    for k in params:
        p = sd64["nerf_dec." + k]
        assert k in grads and p.grad is not None, k
        scale = float(p.grad.abs().max()) + 1e-30
        worst[k] = float((grads[k].cpu().double() - p.grad).abs().max()) / scale
    scale = float(cond_ref.grad.abs().max())
    worst["cond"] = float((g_cond.cpu().double()[:, :dc].reshape(n_rays, n_samples, dc) - cond_ref.grad).abs().max()) / scale
    assert float(g_cond[:, dc:].abs().max()) == 0.0
    return worst


@pytest.mark.parametrize("name,n_rays,n_samples,opts", [
    ("c1_default", 24, 64, {}),
    ("c1_default", 7, 33, {"raytrans_act": "ELU", "raytrans_posenc": True, "density_maskfill": True}),
    ("nonlegacy", 10, 128, {}),
    ("v4", 5, 200, {"density_maskfill": True}),
])
def test_decoder_backward_matches_float64_autograd(name, n_rays, n_samples, opts):
    worst = _case(name, n_rays, n_samples, seed=n_samples, **opts)
    bad = {k: v for k, v in worst.items() if not v < 2e-4}
    print({k: f"{v:.1e}" for k, v in worst.items()})
    assert len(worst) == 33 and not bad, bad


def test_decoder_backward_accumulates_and_skips():
    """g[k] is accumulated into, a tensor mapped to None gets no gradient, an empty chunk is a no-op."""
    from matchnerf_amd import hip
    g, cfg, sd, _ = golden_case("c1_default")
    opt, model = build_model(g["meta"])
    dec = model.nerf_dec
    v = cfg.n_src_views
    dc = sum(opt.encoder.cos_n_group) + 4 * v
    stride = 24
    gen = torch.Generator().manual_seed(5)
    n_rays, s = 6, 64
    n = n_rays * s
    x = (torch.rand(n, 3, generator=gen) * 2 - 1).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1).cuda()
    cond = torch.zeros(n, stride)
    cond[:, :dc] = torch.rand(n, dc, generator=gen)
    cond = cond.cuda()
    g_rgb, g_sig = torch.randn(n, 3, generator=gen).cuda(), torch.randn(n_rays, s, generator=gen).cuda()
    params = {k: p.detach() for k, p in dec.named_parameters()}
    _, g1 = hip.decoder_backward(opt, params, v, x, dirs, cond, stride, g_rgb, g_sig)
    carry = {"rgb_linear.weight": torch.ones_like(params["rgb_linear.weight"]), "pts_bias.weight": None}
    g_cond, g2 = hip.decoder_backward(opt, params, v, x, dirs, cond, stride, g_rgb, g_sig, want_g_cond=False, grads=carry)
    assert g_cond is None and "pts_bias.weight" not in g2
    assert torch.allclose(g2["rgb_linear.weight"], g1["rgb_linear.weight"] + 1.0, rtol=1e-4, atol=1e-5)
    assert torch.allclose(g2["pts_linears.3.weight"], g1["pts_linears.3.weight"], rtol=1e-3, atol=1e-5)
    _, g0 = hip.decoder_backward(opt, params, v, x[:0], dirs[:0], cond[:0], stride, g_rgb[:0], g_sig[:0])
    assert all(float(t.abs().max()) == 0.0 for t in g0.values())
