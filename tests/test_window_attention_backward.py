"""K6 backward (mnerf_window_attention_backward) against float64 autograd through the ORACLE's window attention
(oracle/matchnerf_oracle.py: window_attention, the per-token restatement of gmflow/transformer.py:46-105 that the goldens pin to
the reference)."""
import pytest
import torch

from oracle import matchnerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,h,w,splits,shifted", [
    (2, 16, 24, 2, False),   # 96-token windows: one and a half 64-row tiles
    (2, 16, 24, 2, True),    # wrap-region mask
    (1, 12, 20, 1, False),   # one window = the whole map (global attention), 240 tokens
    (3, 24, 24, 2, True),    # 144-token windows, odd tile remainder
    (2, 32, 40, 4, True),    # attn_splits 4 (rect_wide / IBRNet-style)
    (6, 64, 80, 2, True),    # the DTU shape: 3 pairs x 2 directions, 1280-token windows
])
@pytest.mark.parametrize("forward_stats", [False, True])
@pytest.mark.parametrize("math", ["f16x3", "bf16x6"])
def test_window_attention_backward_matches_float64_autograd(b, h, w, splits, shifted, forward_stats, math, monkeypatch):
    """forward_stats: the training pair (mnerf_window_attention_presplit_stats -> mnerf_window_attention_backward_stats): the
    forward publishes the softmax's row statistics, the backward has no statistics pass.  math: the two split 16-bit forms
    (MNERF_WA_BWD_MATH, read by the library at every call; f16x3 is the default)"""
    from matchnerf_amd import hip
    monkeypatch.setenv("MNERF_WA_BWD_MATH", math)
    gen = torch.Generator().manual_seed(h * 1000 + w * 10 + splits + int(shifted))
    n, c = h * w, 128
    q = torch.randn(b, n, c, generator=gen) * 0.6       # scores ~ N(0, 0.36 * 0.8^2 * 128 / 11.3^2): a peaked but not one-hot softmax
    k = torch.randn(b, n, c, generator=gen) * 0.8
    v = torch.randn(b, n, c, generator=gen)
    g = torch.randn(b, n, c, generator=gen)
    qg, kg, vg, gg = q.cuda(), k.cuda(), v.cuda(), g.cuda()
    stats = torch.full((2, b * n), float("nan"), device="cuda") if forward_stats else None
    out = hip.window_attention(qg, kg, vg, h, w, splits, shifted, row_stats=stats)
    if forward_stats:
        assert torch.isfinite(stats).all() and float(stats[1].min()) >= 1.0   # every token's row was written; sum of exponentials >= its own maximum term
        assert torch.equal(out, hip.window_attention(qg, kg, vg, h, w, splits, shifted))  # the statistics instance computes the same output bits
    gq, gk, gv = hip.window_attention_backward(qg, kg, vg, out, gg, h, w, splits, shifted, row_stats=stats)
    again = hip.window_attention_backward(qg, kg, vg, out, gg, h, w, splits, shifted, row_stats=stats)
    for a, bb in zip((gq, gk, gv), again):
        assert torch.equal(a, bb)                       # no atomics: bit-reproducible
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    o64 = O.window_attention(q64, k64, v64, h, w, splits, shifted)
    assert float((out.cpu().double() - o64.detach()).abs().max()) < 2e-5
    (o64 * g.double()).sum().backward()
    worst = {}
    for name, got, ref in (("q", gq, q64.grad), ("k", gk, k64.grad), ("v", gv, v64.grad)):
        assert torch.isfinite(got).all()
        worst[name] = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
    print({kk: f"{vv:.1e}" for kk, vv in worst.items()})
    assert all(vv < 2e-5 for vv in worst.values()), worst


@pytest.mark.parametrize("math", ["f16x3", "bf16x6", "f32"])
def test_backward_with_gradient_rows_over_24_binades(math, monkeypatch):
    """The f16x3 form's range management on data it was not tuned for: gradient rows scaled by 2^-k, k in [0, 24] (rays hit few
    tokens hard and most barely), q / k / v at different scales.  dQ carries a gain per query, so EVERY row of it is judged
    against its own magnitude; dK / dV sum over the queries and are judged against the tensor's maximum."""
    from matchnerf_amd import hip
    monkeypatch.setenv("MNERF_WA_BWD_MATH", math)
    gen = torch.Generator().manual_seed(77)
    b, h, w, splits, shifted, c = 2, 16, 24, 2, True, 128
    n = h * w
    q = torch.randn(b, n, c, generator=gen) * 3.0
    k = torch.randn(b, n, c, generator=gen) * 0.05
    v = torch.randn(b, n, c, generator=gen) * 40.0
    g = torch.randn(b, n, c, generator=gen) * torch.exp2(-torch.randint(0, 25, (b, n, 1), generator=gen).float()) * 1e-3
    qg, kg, vg, gg = q.cuda(), k.cuda(), v.cuda(), g.cuda()
    out = hip.window_attention(qg, kg, vg, h, w, splits, shifted)
    gq, gk, gv = hip.window_attention_backward(qg, kg, vg, out, gg, h, w, splits, shifted)
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    (O.window_attention(q64, k64, v64, h, w, splits, shifted) * g.double()).sum().backward()
    worst = {}
    for name, got, ref in (("q", gq, q64.grad), ("k", gk, k64.grad), ("v", gv, v64.grad)):
        assert torch.isfinite(got).all()
        worst[name] = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
    err_rows = (gq.cpu().double() - q64.grad).abs().amax(-1) / q64.grad.abs().amax(-1)
    worst["q_rows"] = float(err_rows.max())
    print(math, {kk: f"{vv:.1e}" for kk, vv in worst.items()})
    assert all(worst[kk] < 2e-5 for kk in "qkv"), worst
    assert worst["q_rows"] < 5e-6, worst


@pytest.mark.parametrize("math", ["f16x3", "bf16x6"])
def test_backward_degenerate_inputs(math, monkeypatch):
    """Edge cases of the range management: a zero gradient (absmax 0: gains of an all-zero tensor), a gradient with ONE non-zero row
    (every other query's row norm is 0), a one-hot softmax (scores scaled until P is 0 / 1) and operands at 1e-20 / 1e+18."""
    from matchnerf_amd import hip
    monkeypatch.setenv("MNERF_WA_BWD_MATH", math)
    gen = torch.Generator().manual_seed(5)
    b, h, w, splits, shifted, c = 1, 16, 24, 2, True, 128
    n = h * w
    mk = lambda s: torch.randn(b, n, c, generator=gen) * s

    def run(q, k, v, g):
        qg, kg, vg, gg = q.cuda(), k.cuda(), v.cuda(), g.cuda()
        out = hip.window_attention(qg, kg, vg, h, w, splits, shifted)
        got = hip.window_attention_backward(qg, kg, vg, out, gg, h, w, splits, shifted)
        q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
        (O.window_attention(q64, k64, v64, h, w, splits, shifted) * g.double()).sum().backward()
        return got, (q64.grad, k64.grad, v64.grad)

    # zero gradient: exact zeros
    got, _ = run(mk(0.6), mk(0.8), mk(1.0), torch.zeros(b, n, c))
    assert all(torch.equal(t.cpu(), torch.zeros(b, n, c)) for t in got)
    # one non-zero row
    g = torch.zeros(b, n, c)
    g[0, 77] = torch.randn(c, generator=gen)
    got, ref = run(mk(0.6), mk(0.8), mk(1.0), g)
    for a, r in zip(got, ref):
        assert torch.isfinite(a).all() and float((a.cpu().double() - r).abs().max() / r.abs().max()) < 2e-5
    # one-hot softmax
    got, ref = run(mk(6.0), mk(8.0), mk(1.0), mk(1.0))
    for a, r in zip(got, ref):
        assert torch.isfinite(a).all() and float((a.cpu().double() - r).abs().max() / r.abs().max()) < 2e-5
    # tiny and huge operands (fp32 products of the maxima stay finite)
    got, ref = run(mk(0.6), mk(0.8), mk(1e-20), mk(1e18))
    for a, r in zip(got, ref):
        assert torch.isfinite(a).all() and float((a.cpu().double() - r).abs().max() / r.abs().max()) < 2e-5


def test_autograd_bridge_uses_the_hip_backward(monkeypatch):
    """autograd.window_attention: HIP forward + HIP backward; the torch re-evaluation (MNERF_WA_BACKWARD=torch) agrees."""
    from matchnerf_amd import autograd as ag
    gen = torch.Generator().manual_seed(9)
    b, h, w = 2, 16, 24
    mk = lambda s: (torch.randn(b, h * w, 128, generator=gen) * s).cuda()
    grads = {}
    g = mk(1.0)
    base = [mk(0.6), mk(0.8), mk(1.0)]
    for mode in ("hip", "torch"):
        monkeypatch.setenv("MNERF_WA_BACKWARD", mode)
        q, k, v = (t.clone().requires_grad_(True) for t in base)
        out = ag.window_attention(q, k, v, h, w, 2, True)
        (out * g).sum().backward()
        grads[mode] = (q.grad, k.grad, v.grad)
    for a, bb in zip(grads["hip"], grads["torch"]):
        assert float((a - bb).abs().max() / bb.abs().max()) < 2e-5
