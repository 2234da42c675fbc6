"""Host logic of the decoder weight stream (no GPU): a numpy emulation of the kernel's MFMA
dataflow (csrc/decoder.hip) consumes the packed stream exactly as the wavefront does — same
segment schedule, same (lower | upper) half-wave operand rules — and must reproduce the
oracle decoder.  This pins pack_wstream / decoder_schedule / pack_small without a GPU."""
import numpy as np
import pytest
import torch

from helpers import golden_case, linf
from matchnerf_amd import cond_nerf as CN
from oracle import matchnerf_oracle as O


def emulate_stage(ws, segs, name, b_lo, b_hi):
    """b_lo/b_hi [T, N] operands of the lower/upper half-wave -> Y [nmb*32, N]."""
    parts = [s for s in segs if s[0] == name]
    nmb = parts[0][3]
    y = np.zeros((nmb * 32, b_lo.shape[1]), np.float64)
    for _, first, steps, m, off, _ in parts:
        a = ws[off:off + steps * 64 * m].reshape(steps, 64, m).astype(np.float64)
        for mb in range(m):
            y[mb * 32:(mb + 1) * 32] += a[:, :32, mb].T @ b_lo[first:first + steps]
            y[mb * 32:(mb + 1) * 32] += a[:, 32:, mb].T @ b_hi[first:first + steps]
    return y


def emulate_tail(ws, segs, sub, b_lo, b_hi):
    off = [s for s in segs if s[0] == "tail"][0][4]
    sub_off, t, nmb = CN.TAIL_LAYOUT[sub]
    a = ws[off + sub_off:off + sub_off + t * 64 * nmb].reshape(t, 64, nmb).astype(np.float64)
    y = np.zeros((nmb * 32, b_lo.shape[1]))
    for mb in range(nmb):
        y[mb * 32:(mb + 1) * 32] = a[:, :32, mb].T @ b_lo + a[:, 32:, mb].T @ b_hi
    return y


def reg_order_operands(h, n_blocks):
    lo, hi = CN._reg_order(n_blocks)
    return h[lo], h[hi]


def enc_operands(x, L, legacy):
    """x [3,N] -> operands of the 3L+2 positional-encoding steps (decoder.hip: enc_operand)."""
    lo, hi = [], []
    fm = 1.0 if legacy else np.pi
    for t in range(3 * L):
        l, c = divmod(t, 3)
        arg = x[c].astype(np.float32) * np.float32(2.0 ** l * fm)
        lo.append(np.sin(arg.astype(np.float64)))
        hi.append(np.cos(arg.astype(np.float64)))
    lo += [x[0], x[2]]
    hi += [x[1], np.ones_like(x[0])]
    return np.stack(lo), np.stack(hi)


@pytest.mark.parametrize("name", ["c1_default", "v4", "nonlegacy", "rect_wide"])
def test_emulated_mfma_chain_matches_oracle(name):
    g, cfg, sd, batch = golden_case(name)
    n_rays = 8
    x_ref = torch.from_numpy(g["x_ref"][:n_rays])
    dir_ref = torch.from_numpy(g["dir_ref"][:n_rays])
    cond = torch.from_numpy(g["cond"][:n_rays])
    v = cfg.n_src_views
    mask = cond[..., -v:]
    with torch.no_grad():
        rgb_o, sigma_o = O.decoder(cfg, sd, x_ref, dir_ref, cond, mask)

    ws, cond_dim, cs = CN.pack_wstream(sd, v, cfg.cos_n_group, cfg.L_3D, cfg.legacy_coord)
    segs, total = CN.decoder_schedule(cs, cfg.L_3D)
    assert ws.size == total and cond_dim == cond.shape[-1]
    n = n_rays * cfg.sample_intvs
    x = x_ref.reshape(n, 3).numpy().T.astype(np.float64)
    cpad = np.zeros((cs, n))
    cpad[:cond_dim] = cond.reshape(n, cond_dim).numpy().T
    cpad[cond_dim] = 1.0
    fs = cs // 2
    film = emulate_stage(ws, segs, "film", cpad[:fs], cpad[fs:])
    e_lo, e_hi = enc_operands(x, cfg.L_3D, cfg.legacy_coord)
    h = np.maximum(emulate_stage(ws, segs, "l0", e_lo, e_hi) * film, 0)
    one, zero = np.ones((1, n)), np.zeros((1, n))
    for i in range(1, 5):
        lo, hi = reg_order_operands(h, 4)
        h = np.maximum(emulate_stage(ws, segs, f"l{i}", np.vstack([lo, one]), np.vstack([hi, zero])) * film, 0)
    lo, hi = reg_order_operands(h, 4)
    h = np.maximum((emulate_stage(ws, segs, "l5e", e_lo, e_hi) + emulate_stage(ws, segs, "l5h", lo, hi)) * film, 0)
    lo, hi = reg_order_operands(h, 4)
    a = emulate_stage(ws, segs, "alpha", np.vstack([lo, one]), np.vstack([hi, zero]))
    assert np.abs(a[16:]).max() == 0.0  # padded output rows stay zero
    feat = emulate_stage(ws, segs, "feature", np.vstack([lo, one]), np.vstack([hi, zero]))
    d = np.repeat(dir_ref.numpy().astype(np.float64), cfg.sample_intvs, 0).T
    lo, hi = reg_order_operands(feat, 4)
    hv = np.maximum(emulate_stage(ws, segs, "views", np.vstack([lo, d[0:1], d[2:3]]),
                                  np.vstack([hi, d[1:2], one])), 0)
    lo, hi = reg_order_operands(hv, 2)
    rgb = 1 / (1 + np.exp(-emulate_stage(ws, segs, "rgb", np.vstack([lo, one]), np.vstack([hi, zero]))[:3]))
    assert linf(rgb.T.reshape(n_rays, -1, 3), rgb_o) < 2e-6

    # density branch: q/k/v, fc and the density head through the packed TAIL segment, attention
    # itself (softmax) in numpy, vs the oracle's sigma
    a_t = torch.from_numpy(a[:16].T.reshape(n_rays, -1, 16)).float()
    a_t = O._act(cfg.raytrans_act, a_t)
    if cfg.raytrans_posenc:
        a_t = a_t + torch.from_numpy(CN.raytrans_table(a_t.shape[1]))[None]
    av = a_t.reshape(n, 16).numpy().T.astype(np.float64)                       # [16, N]
    lo16, hi16 = CN._reg_order(1)
    lo16, hi16 = lo16[:8], hi16[:8]
    qkv = emulate_tail(ws, segs, "qkv", av[lo16], av[hi16])                    # rows q|k|v|pad
    assert np.abs(qkv[48:]).max() == 0.0
    s_n = cfg.sample_intvs
    valid = (mask.sum(-1) > 1).reshape(n).numpy()
    q = (qkv[0:16] * 0.5 * valid[None]).T.reshape(n_rays, s_n, 4, 4).transpose(0, 2, 1, 3)
    k = qkv[16:32].T.reshape(n_rays, s_n, 4, 4).transpose(0, 2, 1, 3)
    v_ = qkv[32:48].T.reshape(n_rays, s_n, 4, 4).transpose(0, 2, 1, 3)
    sc = q @ k.transpose(0, 1, 3, 2)
    p = np.exp(sc - sc.max(-1, keepdims=True))
    o = (p / p.sum(-1, keepdims=True)) @ v_                                     # [R,4,S,4]
    o16 = o.transpose(0, 2, 1, 3).reshape(n, 16).T                              # head-major
    x = emulate_tail(ws, segs, "fco", o16[:8], o16[8:])[:16] + av
    small = CN.pack_small(sd, s_n, cfg.raytrans_posenc).astype(np.float64)
    mu = x.mean(0)
    x = (x - mu) / np.sqrt(((x - mu) ** 2).mean(0) + 1e-6) * small[0:16, None] + small[16:32, None]
    z = emulate_tail(ws, segs, "oa0", np.vstack([x[lo16], one]), np.vstack([x[hi16], zero]))[:16]
    z = np.where(z > 0, z, np.exp(z) - 1) if cfg.raytrans_act == "ELU" else np.maximum(z, 0)
    sg = np.maximum(emulate_tail(ws, segs, "oa2", np.vstack([z[lo16], one]), np.vstack([z[hi16], zero]))[0], 0)
    if cfg.density_maskfill:
        sg = np.where(mask.sum(-1).reshape(n).numpy() < 1, 0.0, sg)
    assert linf(sg.reshape(n_rays, s_n), sigma_o) < 2e-6


# ----------------------------------------------------------------------------- split-bf16 stream


def _split3(x):
    hi, mid, lo = CN.split_bf16x3(x.astype(np.float32))
    return [CN.bf16_to_f32(p).astype(np.float64) for p in (hi, mid, lo)]


def emulate_stage16(ws, segs, name, v, six_terms=True):
    """v [T,2,8,N] operands (step, half-wave, j) -> Y [nmb*32, N]; emulates the kernel's arithmetic: bias
    fragment as the initial accumulator, weights and activations as three bf16 terms, six product terms."""
    parts = [s for s in segs if s[0] == name]
    nmb = parts[0][3]
    n = v.shape[-1]
    y = np.zeros((nmb * 32, n), np.float64)
    w16 = ws.view(np.uint16)
    for _, first, steps, m, off, fl, hdr in parts:
        if hdr:
            bias = ws[off:off + CN.FRAG_FLOATS]
            for h in range(2):
                for mb in range(m):
                    for r in range(16):
                        row = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * h
                        y[row] += bias[(h * 4 + mb) * 16 + r]
            off += CN.FRAG_FLOATS
        a = w16[2 * off:2 * off + steps * m * 3 * 512].reshape(steps, m, 3, 64, 8)
        ah, am, al = [CN.bf16_to_f32(a[:, :, p]).astype(np.float64) for p in range(3)]   # [steps,m,64,8]
        vv = v[first:first + steps].astype(np.float32)
        bh, bm, bl = _split3(vv)                                                         # [steps,2,8,N]
        terms = [(ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm)] if six_terms else [(ah + am + al, bh + bm + bl)]
        for wa, vb in terms:
            for mb in range(m):
                for h in range(2):
                    # rows = lanes 32h..32h+31 of block mb; sum over (step, j)
                    y[mb * 32:(mb + 1) * 32] += np.einsum("tlj,tjn->ln", wa[:, mb, 32 * h:32 * h + 32, :], vb[:, h])
    return y


def reg_operands16(h, n_blocks):
    return h[CN._reg_cols16(n_blocks)]                                                   # [T,2,8,N]


def enc_operands16(x, L, legacy):
    lo, hi = enc_operands(x, L, legacy)
    te = (lo.shape[0] + 7) // 8
    v = np.zeros((te, 2, 8, x.shape[1]))
    for a in range(lo.shape[0]):
        v[a // 8, 0, a % 8] = lo[a]
        v[a // 8, 1, a % 8] = hi[a]
    return v


@pytest.mark.parametrize("name", ["c1_default", "v4", "nonlegacy", "rect_wide"])
def test_emulated_split_bf16_chain_matches_oracle(name):
    g, cfg, sd, batch = golden_case(name)
    n_rays = 8
    x_ref = torch.from_numpy(g["x_ref"][:n_rays])
    dir_ref = torch.from_numpy(g["dir_ref"][:n_rays])
    cond = torch.from_numpy(g["cond"][:n_rays])
    v = cfg.n_src_views
    mask = cond[..., -v:]
    with torch.no_grad():
        rgb_o, sigma_o = O.decoder(cfg, sd, x_ref, dir_ref, cond, mask)
    ws, cond_dim, cs = CN.pack_wstream16(sd, v, cfg.cos_n_group, cfg.L_3D, cfg.legacy_coord)
    segs, total = CN.decoder_schedule16(cond_dim, cfg.L_3D)
    assert ws.size == total
    n = n_rays * cfg.sample_intvs
    x = x_ref.reshape(n, 3).numpy().T.astype(np.float64)
    tf = (cond_dim + 15) // 16
    cpad = np.zeros((16 * tf, n))
    cpad[:cond_dim] = cond.reshape(n, cond_dim).numpy().T
    if cond_dim < 16 * tf:
        cpad[cond_dim] = 1.0   # the cost-volume kernel's constant column: must meet zero weights
    film = emulate_stage16(ws, segs, "film", cpad.reshape(tf, 2, 8, n))
    e = enc_operands16(x, cfg.L_3D, cfg.legacy_coord)
    h = np.maximum(emulate_stage16(ws, segs, "l0", e) * film, 0)
    for i in range(1, 5):
        h = np.maximum(emulate_stage16(ws, segs, f"l{i}", reg_operands16(h, 4)) * film, 0)
    h = np.maximum((emulate_stage16(ws, segs, "l5e", e) + emulate_stage16(ws, segs, "l5h", reg_operands16(h, 4))) * film, 0)
    a = emulate_stage16(ws, segs, "alpha", reg_operands16(h, 4))
    assert np.abs(a[16:]).max() == 0.0
    feat = emulate_stage16(ws, segs, "feature", reg_operands16(h, 4))
    d = np.repeat(dir_ref.numpy().astype(np.float64), cfg.sample_intvs, 0).T
    dv = np.zeros((1, 2, 8, n))
    dv[0, 0, :3] = d
    hv = np.maximum(emulate_stage16(ws, segs, "views", np.concatenate([reg_operands16(feat, 4), dv], 0)), 0)
    rgb = 1 / (1 + np.exp(-emulate_stage16(ws, segs, "rgb", reg_operands16(hv, 2))[:3]))
    assert linf(rgb.T.reshape(n_rays, -1, 3), rgb_o) < 2e-6
    # the alpha-head activations feed the (unchanged, exact-f32) tail: compare them with the f32 stream's
    ws32, _, cs32 = CN.pack_wstream(sd, v, cfg.cos_n_group, cfg.L_3D, cfg.legacy_coord)
    segs32, _ = CN.decoder_schedule(cs32, cfg.L_3D)
    tail_floats = segs[-1][5]
    assert np.array_equal(ws[-tail_floats:], ws32[-tail_floats:])
    # and the alpha activations themselves against the oracle's pre-attention features
    lo, hi = reg_order_operands(h, 4)
    one, zero = np.ones((1, n)), np.zeros((1, n))
    a32 = emulate_stage(ws32, segs32, "alpha", np.vstack([lo, one]), np.vstack([hi, zero]))
    assert np.abs(a - a32).max() < 2e-6 * max(1.0, np.abs(a32).max())


def test_schedule16_invariants():
    for cd, L in ((22, 10), (26, 10), (50, 10), (74, 10), (22, 6), (22, 0)):
        segs, total = CN.decoder_schedule16(cd, L)
        assert total % 256 == 0 and len(segs) <= 64
        off = 0
        for name, first, steps, m, o, fl, hdr in segs:
            assert o == off and fl % 256 == 0 and fl <= CN.SEG_CAP_FLOATS
            off += fl
        assert off == total
        assert [s[0] for s in segs][-4:] == ["views", "views", "rgb", "tail"]
        names = [s[0] for s in segs]
        assert names.index("alpha") == max(i for i, n in enumerate(names) if n == "l5h") + 1
        for name in ("l1", "l2", "l3", "l4", "l5h", "feature"):
            assert [s[2] for s in segs if s[0] == name] == [2, 2, 2, 2]


def test_split_bf16x3_is_exact():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(20000) * 10.0 ** rng.integers(-8, 4, 20000)).astype(np.float32)
    hi, mid, lo = CN.split_bf16x3(x)
    rec = CN.bf16_to_f32(hi).astype(np.float64) + CN.bf16_to_f32(mid) + CN.bf16_to_f32(lo)
    assert np.max(np.abs(rec - x) / np.abs(x)) < 2.0 ** -23


def test_schedule_invariants():
    for cs, L in ((24, 10), (32, 10), (56, 10), (64, 10), (24, 6), (24, 0)):
        segs, total = CN.decoder_schedule(cs, L)
        assert total % 256 == 0
        off = 0
        for name, first, steps, m, o, fl in segs:
            assert o == off and fl % 256 == 0 and fl <= CN.SEG_CAP_FLOATS and steps * 64 * m <= fl
            off += fl
        for name, t, m in CN.decoder_stages(cs, L):
            assert sum(s[2] for s in segs if s[0] == name) == t
        assert [s[0] for s in segs if s[0] in ("feature", "views", "rgb", "alpha", "tail")][-4:] == ["views", "rgb", "alpha", "tail"]
        # stages fed from accumulator registers must split 32 | 32(+1) (static unrolling in the kernel)
        for name in ("l1", "l2", "l3", "l4", "feature"):
            assert [s[2] for s in segs if s[0] == name] == [32, 33]
        assert [s[2] for s in segs if s[0] == "l5h"] == [32, 32]
        for name in ("alpha", "views", "rgb", "film", "tail"):
            assert len([s for s in segs if s[0] == name]) == 1
        assert segs[-1][0] == "tail" and segs[-1][5] >= CN.TAIL_FLOATS


def test_small_block_layout():
    g, cfg, sd, _ = golden_case("c1_default")
    s = CN.pack_small(sd, 64, True)
    assert s.size == CN.SMALL_FIXED + 64 * 16
    assert np.array_equal(s[0:16], sd["nerf_dec.ray_attention.layer_norm.weight"].numpy())
    assert np.array_equal(s[16:32], sd["nerf_dec.ray_attention.layer_norm.bias"].numpy())
    assert linf(s[CN.SMALL_FIXED:].reshape(64, 16), O.raytrans_table(64)) == 0.0


# ----------------------------------------------------------------------------- split-fp16 stream


def gain_for(m):
    """the kernel's operand gain: a power of two that puts the sample's largest |operand| m in [2^14, 2^15)"""
    e = np.frexp(np.asarray(m, np.float32))[1]                      # m = f * 2^e, f in [0.5, 1)
    return np.ldexp(np.float32(1.0), (CN.F16_TARGET_EXP + 1) - np.clip(e, -100, 100)).astype(np.float32)


def emulate_stage_h(ws, segs, names, v, gain):
    """v [T,2,8,N] TRUE operand values, gain [N] per-sample power-of-two gain -> Y [nmb*32, N] true outputs.
    Emulates the kernel: operands scaled in fp32 and split into two fp16 terms, weights as (hi, lo) fp16 of
    2^ew W, products hi.hi + hi.lo + lo.hi, bias header times (2^ew gain) as the initial accumulator, and the
    exact inverse scale on the way out.  ``names``: stages that share one accumulator (e.g. l5e + l5h)."""
    names = [names] if isinstance(names, str) else list(names)
    parts = [s for s in segs if s[0] in names]
    nmb = parts[0][3]
    n = v.shape[-1]
    y = np.zeros((nmb * 32, n), np.float64)
    w16 = ws.view(np.float16)
    winv = None
    step0 = 0
    for name in names:
        mine = [s for s in parts if s[0] == name]
        for _, first, steps, m, off, fl, hdr in mine:
            if hdr:
                winv = float(ws[off + 128])
                assert winv == 2.0 ** -int(ws[off + 129]) and float(ws[off + 129]) == int(ws[off + 129])  # what the kernel reads
                bias = ws[off:off + 128].astype(np.float64)
                for h in range(2):
                    for mb in range(m):
                        for r in range(16):
                            row = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * h
                            y[row] += bias[(h * 4 + mb) * 16 + r] * (gain.astype(np.float64) / winv)
                off += CN.FRAG_FLOATS
            a = w16[2 * off:2 * off + steps * m * 2 * 512].reshape(steps, m, 2, 64, 8).astype(np.float64)
            ah, al = a[:, :, 0], a[:, :, 1]
            xs = (v[step0 + first:step0 + first + steps].astype(np.float32) * gain.astype(np.float32)).astype(np.float32)
            assert np.abs(xs).max() < 65504
            bh = xs.astype(np.float16)
            bl = (xs - bh.astype(np.float32)).astype(np.float16)
            bh, bl = bh.astype(np.float64), bl.astype(np.float64)
            for wa, vb in ((ah, bl), (al, bh), (ah, bh)):
                for mb in range(m):
                    for h in range(2):
                        y[mb * 32:(mb + 1) * 32] += np.einsum("tlj,tjn->ln", wa[:, mb, 32 * h:32 * h + 32, :], vb[:, h])
        step0 += sum(s[2] for s in mine)
    return y * (winv / gain.astype(np.float64))


def _rowmax(*arrs):
    return np.max(np.stack([np.abs(a).reshape(-1, a.shape[-1]).max(0) for a in arrs]), 0)


@pytest.mark.parametrize("name", ["c1_default", "v4", "nonlegacy", "rect_wide"])
def test_emulated_split_fp16_chain_matches_oracle(name):
    g, cfg, sd, batch = golden_case(name)
    n_rays = 8
    x_ref = torch.from_numpy(g["x_ref"][:n_rays])
    dir_ref = torch.from_numpy(g["dir_ref"][:n_rays])
    cond = torch.from_numpy(g["cond"][:n_rays])
    v = cfg.n_src_views
    mask = cond[..., -v:]
    with torch.no_grad():
        rgb_o, sigma_o = O.decoder(cfg, sd, x_ref, dir_ref, cond, mask)
    ws, cond_dim, cs = CN.pack_wstream_h(sd, v, cfg.cos_n_group, cfg.L_3D, cfg.legacy_coord)
    segs, total = CN.decoder_schedule_h(cond_dim, cfg.L_3D)
    assert ws.size == total
    n = n_rays * cfg.sample_intvs
    x = x_ref.reshape(n, 3).numpy().T.astype(np.float64)
    tf = (cond_dim + 15) // 16
    cpad = np.zeros((16 * tf, n))
    cpad[:cond_dim] = cond.reshape(n, cond_dim).numpy().T
    if cond_dim < 16 * tf:
        cpad[cond_dim] = 1.0
    fixed = np.full(n, 2.0 ** CN.F16_TARGET_EXP, np.float32)            # cosines, colours, masks: |.| <= 1
    film = emulate_stage_h(ws, segs, "film", cpad.reshape(tf, 2, 8, n), fixed)
    e = enc_operands16(x, cfg.L_3D, cfg.legacy_coord)
    enc_max = np.maximum(1.0, np.abs(x).max(0))
    h = np.maximum(emulate_stage_h(ws, segs, "l0", e, gain_for(enc_max)) * film, 0)
    for i in range(1, 5):
        h = np.maximum(emulate_stage_h(ws, segs, f"l{i}", reg_operands16(h, 4), gain_for(_rowmax(h))) * film, 0)
    g5 = gain_for(np.maximum(_rowmax(h), enc_max))
    h = np.maximum(emulate_stage_h(ws, segs, ["l5e", "l5h"], np.concatenate([e, reg_operands16(h, 4)], 0), g5) * film, 0)
    gh = gain_for(_rowmax(h))
    a = emulate_stage_h(ws, segs, "alpha", reg_operands16(h, 4), gh)
    assert np.abs(a[16:]).max() == 0.0
    feat = emulate_stage_h(ws, segs, "feature", reg_operands16(h, 4), gh)
    d = np.repeat(dir_ref.numpy().astype(np.float64), cfg.sample_intvs, 0).T
    dv = np.zeros((1, 2, 8, n))
    dv[0, 0, :3] = d
    gv = gain_for(np.maximum(_rowmax(feat), 1.0))
    hv = np.maximum(emulate_stage_h(ws, segs, "views", np.concatenate([reg_operands16(feat, 4), dv], 0), gv), 0)
    rgb = 1 / (1 + np.exp(-emulate_stage_h(ws, segs, "rgb", reg_operands16(hv, 2), gain_for(_rowmax(hv)))[:3]))
    err = linf(rgb.T.reshape(n_rays, -1, 3), rgb_o)
    print(f"f16x3 emulation {name}: per-sample rgb err {err:.2e}")
    assert err < 5e-6
    ws32, _, cs32 = CN.pack_wstream(sd, v, cfg.cos_n_group, cfg.L_3D, cfg.legacy_coord)
    segs32, _ = CN.decoder_schedule(cs32, cfg.L_3D)
    tail_floats = segs[-1][5]
    assert np.array_equal(ws[-tail_floats:], ws32[-tail_floats:])
    lo, hi = reg_order_operands(h, 4)
    one, zero = np.ones((1, n)), np.zeros((1, n))
    a32 = emulate_stage(ws32, segs32, "alpha", np.vstack([lo, one]), np.vstack([hi, zero]))
    assert np.abs(a - a32).max() < 4e-6 * max(1.0, np.abs(a32).max())


def test_schedule_h_invariants():
    for cd, L in ((22, 10), (26, 10), (50, 10), (74, 10), (22, 6), (22, 0)):
        segs, total = CN.decoder_schedule_h(cd, L)
        assert total % 256 == 0 and len(segs) <= 64
        off = 0
        for name, first, steps, m, o, fl, hdr in segs:
            assert o == off and fl % 256 == 0 and fl <= CN.SEG_CAP_FLOATS
            off += fl
        assert off == total
        names = [s[0] for s in segs]
        assert names[-4:] == ["views", "views", "rgb", "tail"]
        assert names.index("alpha") == max(i for i, n in enumerate(names) if n == "l5h") + 1
        for name in ("l1", "l2", "l3", "l4", "l5h", "feature"):
            assert [s[2] for s in segs if s[0] == name] == [4, 4]


def test_split_f16x2_is_22_bit():
    rng = np.random.default_rng(0)
    x = (rng.uniform(0.25, 32768.0, 20000) * rng.choice([-1, 1], 20000)).astype(np.float32)
    hi, lo = CN.split_f16x2(x)
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    assert np.max(np.abs(rec - x) / np.abs(x)) < 2.0 ** -21.9
    w = rng.standard_normal((128, 128)).astype(np.float32) * 0.09
    ew = CN.f16_weight_exponent(w)
    assert 2.0 ** 13 <= np.abs(w).max() * 2.0 ** ew < 2.0 ** 14
