"""packing.py on the device: the streams a GPU-resident transformer packs for itself equal the host packers' bit for bit, and an
optimizer step costs one device->host copy for the whole transformer, not one per weight."""
import numpy as np
import pytest
import torch

from matchnerf_amd import gmflow, packing

pytestmark = pytest.mark.gpu


def test_refresh_packs_on_the_device_equals_the_host_packers(monkeypatch):
    torch.manual_seed(3)
    ft = gmflow.FeatureTransformer(num_layers=6).cuda()
    calls = []
    inner = packing.exponent_of_absmax
    monkeypatch.setattr(packing, "exponent_of_absmax", lambda m: calls.append(m) or inner(m))
    assert ft.refresh_packs(torch.device("cuda:0")) == 24           # 12 layers x (qkv, block)
    assert len(calls) == 6 * (3 + 1) + 6 * (3 + 3)                  # 60 exponents out of ONE device->host copy
    assert ft.refresh_packs(torch.device("cuda:0")) == 0
    for blk in ft.layers:
        for layer in (blk.self_attn, blk.cross_attn_ffn):
            ws, ews = layer._qkv[1:]
            want, ews_n = gmflow.pack_qkv(layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight)
            assert ews == ews_n and np.array_equal(ws.cpu().numpy().view(np.uint32), want.view(np.uint32))
            ws, ln, ews = layer._blk[1:]
            want, ews_n = gmflow.pack_encoder_block(layer.merge.weight, None if layer.no_ffn else layer.mlp[0].weight,
                                                    None if layer.no_ffn else layer.mlp[2].weight)
            assert ews == ews_n and np.array_equal(ws.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # an optimizer step makes every stream stale again: one more batched fetch
    opt = torch.optim.SGD(ft.parameters(), lr=1e-3)
    for p in ft.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    calls.clear()
    assert ft.refresh_packs(torch.device("cuda:0")) == 24 and len(calls) == 60
