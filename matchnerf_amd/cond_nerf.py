"""CondNeRF — the conditional radiance-field decoder of MatchNeRF, MI355X-native.

Host-side mirror of /root/reference/models/rfdecoder/cond_nerf.py:8-50 (+ nerf.py,
ray_transformer.py): the module owns the same parameters under the same ``state_dict`` keys
(binding for checkpoint drop-in), but it does not evaluate them with
torch ops.  ``pack_wstream_h`` / ``pack_wstream16`` / ``pack_wstream`` re-lay the weights out as the MFMA
A-fragment stream consumed by the fused HIP kernel (csrc/decoder.hip) and the evaluation itself happens in
``libmnerf_hip.so`` (``mnerf_decoder_chunk`` / ``mnerf_decoder_samples`` / ``mnerf_render_chunk``).

Weight-stream layouts (also DESIGN.md §5).  Three formats: the exact-f32 one described here, the split-bf16
one (``pack_wstream16``) and the split-fp16 one (default; ``pack_wstream_h``) — the last two run the same chain
with 16-wide K-steps.
------------------------------------------------------------
f32 format: every Linear ``y = W x + b`` is evaluated transposed with ``v_mfma_f32_32x32x2_f32``:
for K-step ``t`` and output block ``m`` the wave needs one float per lane,
``A[t][lane][m] = W[m*32 + (lane & 31)][col(t, lane >> 5)]`` where ``col(t, half)`` is the
input feature the lower / upper half-wave feeds at that step:

* hidden activations arrive in accumulator-register order: step ``t = 16*mb + r`` pairs
  features ``32*mb + (r & 3) + 8*(r >> 2)`` (lower half) and that ``+ 4`` (upper half);
* positional encoding: step ``t = 3*l + c`` pairs ``sin(2^l x_c)`` and ``cos(2^l x_c)``;
  the two trailing steps carry ``(x | y)`` and ``(z | 1)``;
* conditioning vector: step ``t`` pairs ``cond[t]`` and ``cond[stride/2 + t]``;
* a bias is one more column multiplied by the constant 1 operand.

Stages in consumption order (steps x output blocks):
FiLM(stride/2 x4) L0(3L+2 x4) L1..L4(65 x4) L5-enc(3L+2 x4) L5-h(64 x4) feature(65 x4)
views(66 x2) rgb(33 x1) alpha(65 x1) and a "tail" = [qkv(8 x2) | fc(8 x1) | out_alpha.0(9 x1) |
out_alpha.2(9 x1)] — the ray-transformer projections and the density head, which stay resident
in LDS across the attention phase.  Each stage is cut into segments of at most 33 KiB, each
segment padded to a multiple of 256 floats (1 KiB DMA pieces); the tail is one segment.
"""
import math

import numpy as np
import torch
import torch.nn as nn

SEG_CAP_FLOATS = 33 * 256
SMALL_FIXED = 32  # floats of the `small` block before the ray-posenc table: LayerNorm weight | bias
# sub-stages of the tail segment: name -> (float offset in segment, K-steps, output blocks)
TAIL_LAYOUT = {"qkv": (0, 8, 2), "fco": (1024, 8, 1), "oa0": (1536, 9, 1), "oa2": (2112, 9, 1)}
TAIL_FLOATS = 2688


# ----------------------------------------------------------------------------- schedule


def decoder_stages(cond_stride, L_3D):
    fs, es = cond_stride // 2, 3 * L_3D + 2
    names = ["film", "l0", "l1", "l2", "l3", "l4", "l5e", "l5h", "feature", "views", "rgb", "alpha"]
    steps = [fs, es, 65, 65, 65, 65, es, 64, 65, 66, 33, 65]
    nmb = [4, 4, 4, 4, 4, 4, 4, 4, 4, 2, 1, 1]
    return list(zip(names, steps, nmb))


def decoder_schedule(cond_stride, L_3D):
    """Mirror of build_schedule() in csrc/decoder.hip -> (segments, total_floats);
    segment = (stage, first_step, n_steps, nmb, float_offset, padded_floats)."""
    segs, off = [], 0
    for name, t, m in decoder_stages(cond_stride, L_3D):
        cap = SEG_CAP_FLOATS // (64 * m)
        nseg = (t + cap - 1) // cap
        base = t // nseg
        first = 0
        for k in range(nseg):
            steps = t - base * (nseg - 1) if k == nseg - 1 else base
            fl = ((steps * 64 * m + 255) // 256) * 256
            assert fl <= SEG_CAP_FLOATS
            segs.append((name, first, steps, m, off, fl))
            off += fl
            first += steps
    fl = ((TAIL_FLOATS + 255) // 256) * 256
    segs.append(("tail", 0, 0, 0, off, fl))
    off += fl
    return segs, off


# ----------------------------------------------------------------------------- column maps

BIAS, ZERO = -1, -2


def _reg_order(n_blocks):
    """input feature fed by (lower, upper) half-wave at step t when the operand is a previous
    layer's accumulator (C/D layout of v_mfma_f32_32x32x2_f32)."""
    lo, hi = [], []
    for mb in range(n_blocks):
        for r in range(16):
            f = 32 * mb + (r & 3) + 8 * (r >> 2)
            lo.append(f)
            hi.append(f + 4)
    return lo, hi


def _enc_cols(L, legacy, base=0):
    """columns of the positional-encoding part of a weight matrix per step.
    legacy layout  [x(3) | sin(l-major,c) (3L) | cos (3L)]      (cond_nerf.py:108-116)
    regular layout [x(3) | per c: sin(l=0..L-1), cos(l=0..L-1)] (nerf.py:126-133)"""
    lo, hi = [], []
    for t in range(3 * L):
        l, c = divmod(t, 3)
        if legacy:
            lo.append(base + 3 + t)
            hi.append(base + 3 + 3 * L + t)
        else:
            lo.append(base + 3 + c * 2 * L + l)
            hi.append(base + 3 + c * 2 * L + L + l)
    lo += [base + 0, base + 2]
    hi += [base + 1, BIAS]
    return lo, hi


def _fragments(weight, bias, lo, hi, nmb):
    """A[t, lane, m] for one stage (numpy float32 [T, 64, nmb])."""
    n_out, n_in = weight.shape
    rows = nmb * 32
    ext = np.zeros((rows, n_in + 2), np.float32)
    ext[:n_out, :n_in] = weight
    if bias is not None:
        ext[:n_out, n_in] = bias
    lo = np.asarray([n_in if c == BIAS else (n_in + 1 if c == ZERO else c) for c in lo])
    hi = np.asarray([n_in if c == BIAS else (n_in + 1 if c == ZERO else c) for c in hi])
    t_n = len(lo)
    lane = np.arange(64)
    col = np.where((lane >> 5)[None, :] == 0, lo[:, None], hi[:, None])          # [T,64]
    row = (lane & 31)[None, :, None] + 32 * np.arange(nmb)[None, None, :]         # [1,64,nmb]
    return ext[np.broadcast_to(row, (t_n, 64, nmb)), np.broadcast_to(col[:, :, None], (t_n, 64, nmb))]


def pack_wstream(sd, n_views, cos_n_group, L_3D=10, legacy=True, prefix="nerf_dec."):
    """state_dict (numpy or torch, reference keys) -> (wstream float32 [total], cond_dim, cond_stride)."""

    def g(name):
        v = sd[prefix + name]
        return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)

    cond_dim = int(sum(cos_n_group)) + 4 * n_views
    cond_stride = ((cond_dim + 1 + 7) // 8) * 8
    fs = cond_stride // 2
    d_enc = 3 + 6 * L_3D
    h_lo, h_hi = _reg_order(4)
    e_lo, e_hi = _enc_cols(L_3D, legacy)
    film_lo = [t if t < cond_dim else (BIAS if t == cond_dim else ZERO) for t in range(fs)]
    film_hi = [fs + t if fs + t < cond_dim else (BIAS if fs + t == cond_dim else ZERO) for t in range(fs)]
    w5 = g("pts_linears.5.weight")
    assert w5.shape[1] == d_enc + 128, "decoder.skip must be [4] with net_width 128"
    hv_lo, hv_hi = _reg_order(2)
    frags = {
        "film": _fragments(g("pts_bias.weight"), g("pts_bias.bias"), film_lo, film_hi, 4),
        "l0": _fragments(g("pts_linears.0.weight"), g("pts_linears.0.bias"), e_lo, e_hi, 4),
        "l5e": _fragments(w5[:, :d_enc], g("pts_linears.5.bias"), e_lo, e_hi, 4),
        "l5h": _fragments(w5[:, d_enc:], None, h_lo, h_hi, 4),
        "alpha": _fragments(g("alpha_linear.0.weight"), g("alpha_linear.0.bias"), h_lo + [BIAS], h_hi + [ZERO], 1),
        "feature": _fragments(g("feature_linear.weight"), g("feature_linear.bias"), h_lo + [BIAS], h_hi + [ZERO], 4),
        "views": _fragments(g("views_linears.0.weight"), g("views_linears.0.bias"),
                            h_lo + [128, 130], h_hi + [129, BIAS], 2),
        "rgb": _fragments(g("rgb_linear.weight"), g("rgb_linear.bias"), hv_lo + [BIAS], hv_hi + [ZERO], 1),
    }
    for i in range(1, 5):
        frags[f"l{i}"] = _fragments(g(f"pts_linears.{i}.weight"), g(f"pts_linears.{i}.bias"),
                                    h_lo + [BIAS], h_hi + [ZERO], 4)
    segs, total = decoder_schedule(cond_stride, L_3D)
    # tail: ray-transformer projections [w_qs; w_ks; w_vs] (48 rows), fc, out_alpha_linear.{0,2}
    a_lo, a_hi = _reg_order(1)
    a_lo, a_hi = a_lo[:8], a_hi[:8]                       # 16 inputs held in registers 0..7 of one block
    wqkv = np.concatenate([g("ray_attention.w_qs.weight"), g("ray_attention.w_ks.weight"),
                           g("ray_attention.w_vs.weight")], 0)
    tail = {
        "qkv": _fragments(wqkv, None, a_lo, a_hi, 2),
        "fco": _fragments(g("ray_attention.fc.weight"), None, list(range(8)), list(range(8, 16)), 1),
        "oa0": _fragments(g("out_alpha_linear.0.weight"), g("out_alpha_linear.0.bias"), a_lo + [BIAS], a_hi + [ZERO], 1),
        "oa2": _fragments(g("out_alpha_linear.2.weight"), g("out_alpha_linear.2.bias"), a_lo + [BIAS], a_hi + [ZERO], 1),
    }
    out = np.zeros(total, np.float32)
    for name, first, steps, m, off, _ in segs:
        if name == "tail":
            for sub, (sub_off, t, nm) in TAIL_LAYOUT.items():
                a = tail[sub]
                assert a.shape == (t, 64, nm), (sub, a.shape)
                out[off + sub_off:off + sub_off + a.size] = a.reshape(-1)
            continue
        a = frags[name][first:first + steps]
        assert a.shape == (steps, 64, m), (name, a.shape, steps, m)
        out[off:off + a.size] = a.reshape(-1)
    return out, cond_dim, cond_stride


# ----------------------------------------------------------------------------- split-bf16 stream
# Format 1 ("bf16x6"): the same transposed MFMA chain on v_mfma_f32_32x32x16_bf16.  Every fp32
# weight is split into three bf16 terms w = hi + mid + lo (24 significand bits, exact), the kernel
# splits the activations the same way on the fly, and a product is accumulated in fp32 from the six
# terms hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid (dropped terms < 2^-24 relative): fp32-grade
# results (measured: max error BELOW an fp32 FMA chain's) at 16/6 of the f32 matrix rate.
#
# A K-step covers 16 inputs: lane (n, half) supplies k = 8*half + j, j < 8.  With activations in
# accumulator-register order, step t of a 16-register block takes registers 8*(t&1)+j, i.e. feature
# 32*blk + (reg&3) + 8*(reg>>2) + 4*half — again no transposes between layers.
# Stream unit = 1 KiB fragment (64 lanes x 8 bf16).  Per (step, block): [hi | mid | lo] fragments.
# The first segment of a stage starts with a 1 KiB fp32 BIAS fragment, bias[(half*4 + m)*16 + r],
# which the kernel loads as the initial accumulator value (so biases stay exact fp32).

FRAG_FLOATS = 256


def decoder_stages16(cond_dim, L_3D):
    """(name, nmb, [K16-steps per segment], has_bias) in consumption order."""
    tf = (cond_dim + 15) // 16
    te = (3 * L_3D + 2 + 7) // 8

    def pairs(t):
        return [2] * (t // 2) + ([1] if t % 2 else [])

    st = [("film", 4, pairs(tf), True), ("l0", 4, pairs(te), True)]
    st += [(f"l{i}", 4, [2, 2, 2, 2], True) for i in range(1, 5)]
    # alpha right after the trunk: it and feature are the only consumers of the last hidden activations, which
    # are then dead during views / rgb (its 16 outputs wait in 8 registers for the ray transformer)
    st += [("l5e", 4, pairs(te), True), ("l5h", 4, [2, 2, 2, 2], False), ("alpha", 1, [8], True),
           ("feature", 4, [2, 2, 2, 2], True), ("views", 2, [4, 5], True), ("rgb", 1, [4], True)]
    return st


def decoder_schedule16(cond_dim, L_3D):
    """-> (segments, total_floats); segment = (stage, first_step, n_steps, nmb, float_offset, floats, has_bias_header)."""
    segs, off = [], 0
    for name, m, seg_steps, has_bias in decoder_stages16(cond_dim, L_3D):
        first = 0
        for k, steps in enumerate(seg_steps):
            hdr = has_bias and k == 0
            fl = (steps * m * 3 + (1 if hdr else 0)) * FRAG_FLOATS
            assert fl <= SEG_CAP_FLOATS
            segs.append((name, first, steps, m, off, fl, hdr))
            off += fl
            first += steps
    fl = ((TAIL_FLOATS + 255) // 256) * 256
    segs.append(("tail", 0, 0, 0, off, fl, False))
    return segs, off + fl


def bf16_round(x):
    """float32 array -> bf16 bit patterns (uint16), round to nearest even."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def split_bf16x3(w):
    """fp32 -> (hi, mid, lo) bf16 bit patterns with hi + mid + lo == w (to 24 bits)."""
    w = np.ascontiguousarray(w, np.float32)
    hi = bf16_round(w)
    r1 = w - bf16_to_f32(hi)
    mid = bf16_round(r1)
    r2 = r1 - bf16_to_f32(mid)
    return hi, mid, bf16_round(r2)


def _reg_cols16(n_blocks):
    """[T=2*n_blocks, 2, 8] input feature per (step, half, j) for accumulator-order operands."""
    cols = np.zeros((2 * n_blocks, 2, 8), np.int64)
    for t in range(2 * n_blocks):
        for h in range(2):
            for j in range(8):
                reg = 8 * (t & 1) + j
                cols[t, h, j] = 32 * (t >> 1) + (reg & 3) + 8 * (reg >> 2) + 4 * h
    return cols


def _enc_cols16(L, legacy):
    """positional-encoding slots a = 8t + j: (sin | cos) of argument a, then (x | y), (z | -)."""
    lo, hi = _enc_cols(L, legacy)
    te = (len(lo) + 7) // 8
    cols = np.full((te, 2, 8), ZERO, np.int64)
    for a in range(len(lo)):
        cols[a // 8, 0, a % 8] = lo[a]
        cols[a // 8, 1, a % 8] = ZERO if hi[a] == BIAS else hi[a]
    return cols


def _fragments16(weight, cols, nmb):
    """uint16 [T, nmb, 3, 64, 8]: bf16 parts of W[32m + lane%32, cols[t, lane//32, j]]."""
    n_out, n_in = weight.shape
    ext = np.zeros((nmb * 32, n_in + 1), np.float32)
    ext[:n_out, :n_in] = weight
    cols = np.where(cols < 0, n_in, cols)
    t_n = cols.shape[0]
    lane = np.arange(64)
    col = cols[:, lane >> 5, :]                                               # [T,64,8]
    row = (lane & 31)[None, None, :, None] + 32 * np.arange(nmb)[None, :, None, None]  # [1,nmb,64,1]
    w = ext[np.broadcast_to(row, (t_n, nmb, 64, 8)), np.broadcast_to(col[:, None], (t_n, nmb, 64, 8))]
    return np.stack(split_bf16x3(w), 2)                                      # [T,nmb,3,64,8]


def _bias_fragment(bias, nmb):
    out = np.zeros(FRAG_FLOATS, np.float32)
    if bias is None:
        return out
    b = np.zeros(nmb * 32, np.float32)
    b[:bias.shape[0]] = bias
    for h in range(2):
        for m in range(nmb):
            for r in range(16):
                out[(h * 4 + m) * 16 + r] = b[32 * m + (r & 3) + 8 * (r >> 2) + 4 * h]
    return out


def pack_wstream16(sd, n_views, cos_n_group, L_3D=10, legacy=True, prefix="nerf_dec."):
    """state_dict -> (wstream as float32 words [total], cond_dim, cond_stride) in the split-bf16 format."""

    def g(name):
        v = sd[prefix + name]
        return (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)).astype(np.float32)

    cond_dim = int(sum(cos_n_group)) + 4 * n_views
    cond_stride = ((cond_dim + 1 + 7) // 8) * 8
    d_enc = 3 + 6 * L_3D
    tf = (cond_dim + 15) // 16
    film_cols = np.arange(16 * tf).reshape(tf, 2, 8)
    film_cols = np.where(film_cols < cond_dim, film_cols, ZERO)
    e_cols = _enc_cols16(L_3D, legacy)
    h_cols = _reg_cols16(4)
    w5 = g("pts_linears.5.weight")
    assert w5.shape[1] == d_enc + 128, "decoder.skip must be [4] with net_width 128"
    dir_cols = np.full((1, 2, 8), ZERO, np.int64)
    dir_cols[0, 0, :3] = [128, 129, 130]
    stages = {
        "film": (g("pts_bias.weight"), g("pts_bias.bias"), film_cols),
        "l0": (g("pts_linears.0.weight"), g("pts_linears.0.bias"), e_cols),
        "l5e": (w5[:, :d_enc], g("pts_linears.5.bias"), e_cols),
        "l5h": (w5[:, d_enc:], None, h_cols),
        "feature": (g("feature_linear.weight"), g("feature_linear.bias"), h_cols),
        "views": (g("views_linears.0.weight"), g("views_linears.0.bias"), np.concatenate([h_cols, dir_cols], 0)),
        "rgb": (g("rgb_linear.weight"), g("rgb_linear.bias"), _reg_cols16(2)),
        "alpha": (g("alpha_linear.0.weight"), g("alpha_linear.0.bias"), h_cols),
    }
    for i in range(1, 5):
        stages[f"l{i}"] = (g(f"pts_linears.{i}.weight"), g(f"pts_linears.{i}.bias"), h_cols)
    segs, total = decoder_schedule16(cond_dim, L_3D)
    f32_tail = pack_wstream(sd, n_views, cos_n_group, L_3D, legacy, prefix)[0][-((TAIL_FLOATS + 255) // 256) * 256:]
    out = np.zeros(total, np.float32)
    out16 = out.view(np.uint16)
    frag_cache = {}
    for name, first, steps, m, off, fl, hdr in segs:
        if name == "tail":
            out[off:off + fl] = f32_tail
            continue
        w, b, cols = stages[name]
        if name not in frag_cache:
            frag_cache[name] = _fragments16(w, cols, m)
        if hdr:
            out[off:off + FRAG_FLOATS] = _bias_fragment(b, m)
            off += FRAG_FLOATS
        a = frag_cache[name][first:first + steps]
        out16[2 * off:2 * off + a.size] = a.reshape(-1)
    return out, cond_dim, cond_stride


# ----------------------------------------------------------------------------- split-fp16 stream
# Format 2 ("f16x3", default): the same transposed chain on v_mfma_f32_32x32x16_f16 with TWO fp16 terms per
# operand and THREE products per MAC (hi.hi + hi.lo + lo.hi; the dropped lo.lo is < 2^-22 of the product) -
# half the matrix work and ~half the operand-split VALU work of bf16x6.  fp16 has 11 significand bits but only a
# 5-bit exponent, so both operands are range-managed with exact power-of-two scales:
#   * weights: one scale 2^ew per weight TENSOR chosen on the host so that max|W| lands in [2^13, 2^14);
#     hi = f16(W 2^ew), lo = f16(W 2^ew - hi) (round-to-nearest; residual exact in fp32).  2^-ew travels in the
#     stage header;
#   * activations: one scale per SAMPLE and stage chosen in the kernel from the running maximum of the sample's
#     128 features (in-lane max + one cross-half shuffle) so that the largest operand lands in [2^14, 2^15).
# The accumulator then holds 2^ew * g * (W x) and the layer epilogue multiplies the exact inverse back in.
# Error per product <= ~2^-22 relative (fp32 chain: 2^-24 per rounding); elements more than 2^17 below the
# sample's largest feature lose their lo term to fp16 underflow, an absolute error below 2^-25 of that maximum.
# Stream unit = (K16-step, block): [hi | lo] fragments of 1 KiB (64 lanes x 8 fp16), same (lane, j) -> k map as
# format 1.  A stage's first segment starts with a 1 KiB fp32 header: floats [0,128) = bias in accumulator order
# (UNscaled; the kernel multiplies it by the operand gain), float [128] = 2^-ew, float [129] = ew.

F16_TARGET_EXP = 14  # weights: max |W 2^ew| in [2^13, 2^14); activations: max in [2^14, 2^15)
H_SEG_STEPS = 4      # K16-steps per segment of a 4-block stage: 4 x 4 x 2 KiB + 1 KiB header = 33 KiB


def decoder_stages_h(cond_dim, L_3D):
    """(name, nmb, [K16-steps per segment], has_header) in consumption order for the split-fp16 stream."""
    tf = (cond_dim + 15) // 16
    te = (3 * L_3D + 2 + 7) // 8

    def chunks(t, per):
        return [per] * (t // per) + ([t % per] if t % per else [])

    st = [("film", 4, chunks(tf, H_SEG_STEPS), True), ("l0", 4, chunks(te, H_SEG_STEPS), True)]
    st += [(f"l{i}", 4, [4, 4], True) for i in range(1, 5)]
    st += [("l5e", 4, chunks(te, H_SEG_STEPS), True), ("l5h", 4, [4, 4], False), ("alpha", 1, [8], True),
           ("feature", 4, [4, 4], True), ("views", 2, [4, 5], True), ("rgb", 1, [4], True)]
    return st


def decoder_schedule_h(cond_dim, L_3D):
    """-> (segments, total_floats); segment = (stage, first_step, n_steps, nmb, float_offset, floats, has_header)."""
    segs, off = [], 0
    for name, m, seg_steps, has_hdr in decoder_stages_h(cond_dim, L_3D):
        first = 0
        for k, steps in enumerate(seg_steps):
            hdr = has_hdr and k == 0
            fl = (steps * m * 2 + (1 if hdr else 0)) * FRAG_FLOATS
            assert fl <= SEG_CAP_FLOATS, (name, fl)
            segs.append((name, first, steps, m, off, fl, hdr))
            off += fl
            first += steps
    fl = ((TAIL_FLOATS + 255) // 256) * 256
    segs.append(("tail", 0, 0, 0, off, fl, False))
    return segs, off + fl


def f16_weight_exponent(w):
    """ew such that max|w| * 2^ew is in [2^(F16_TARGET_EXP-1), 2^F16_TARGET_EXP)  (0 for an all-zero tensor)."""
    m = float(np.max(np.abs(w))) if w.size else 0.0
    if m == 0.0 or not np.isfinite(m):
        return 0
    return int(F16_TARGET_EXP - np.frexp(np.float32(m))[1])


def split_f16x2(x):
    """fp32 -> (hi, lo) fp16 with hi + lo ~= x to 22 bits (round to nearest; the residual is exact in fp32)."""
    x = np.ascontiguousarray(x, np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def _fragments_h(weight, cols, nmb, ew):
    """float16 [T, nmb, 2, 64, 8]: (hi, lo) parts of 2^ew * W[32m + lane%32, cols[t, lane//32, j]]."""
    n_out, n_in = weight.shape
    ext = np.zeros((nmb * 32, n_in + 1), np.float32)
    ext[:n_out, :n_in] = np.ldexp(weight.astype(np.float32), ew)
    cols = np.where(cols < 0, n_in, cols)
    t_n = cols.shape[0]
    lane = np.arange(64)
    col = cols[:, lane >> 5, :]
    row = (lane & 31)[None, None, :, None] + 32 * np.arange(nmb)[None, :, None, None]
    w = ext[np.broadcast_to(row, (t_n, nmb, 64, 8)), np.broadcast_to(col[:, None], (t_n, nmb, 64, 8))]
    return np.stack(split_f16x2(w), 2)


def pack_wstream_h(sd, n_views, cos_n_group, L_3D=10, legacy=True, prefix="nerf_dec."):
    """state_dict -> (wstream as float32 words [total], cond_dim, cond_stride) in the split-fp16 format."""

    def g(name):
        v = sd[prefix + name]
        return (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)).astype(np.float32)

    cond_dim = int(sum(cos_n_group)) + 4 * n_views
    cond_stride = ((cond_dim + 1 + 7) // 8) * 8
    d_enc = 3 + 6 * L_3D
    tf = (cond_dim + 15) // 16
    film_cols = np.arange(16 * tf).reshape(tf, 2, 8)
    film_cols = np.where(film_cols < cond_dim, film_cols, ZERO)
    e_cols = _enc_cols16(L_3D, legacy)
    h_cols = _reg_cols16(4)
    w5 = g("pts_linears.5.weight")
    assert w5.shape[1] == d_enc + 128, "decoder.skip must be [4] with net_width 128"
    ew5 = f16_weight_exponent(w5)  # one tensor, one accumulator: both parts share the scale
    dir_cols = np.full((1, 2, 8), ZERO, np.int64)
    dir_cols[0, 0, :3] = [128, 129, 130]
    stages = {
        "film": (g("pts_bias.weight"), g("pts_bias.bias"), film_cols, None),
        "l0": (g("pts_linears.0.weight"), g("pts_linears.0.bias"), e_cols, None),
        "l5e": (w5[:, :d_enc], g("pts_linears.5.bias"), e_cols, ew5),
        "l5h": (w5[:, d_enc:], None, h_cols, ew5),
        "feature": (g("feature_linear.weight"), g("feature_linear.bias"), h_cols, None),
        "views": (g("views_linears.0.weight"), g("views_linears.0.bias"), np.concatenate([h_cols, dir_cols], 0), None),
        "rgb": (g("rgb_linear.weight"), g("rgb_linear.bias"), _reg_cols16(2), None),
        "alpha": (g("alpha_linear.0.weight"), g("alpha_linear.0.bias"), h_cols, None),
    }
    for i in range(1, 5):
        stages[f"l{i}"] = (g(f"pts_linears.{i}.weight"), g(f"pts_linears.{i}.bias"), h_cols, None)
    segs, total = decoder_schedule_h(cond_dim, L_3D)
    f32_tail = pack_wstream(sd, n_views, cos_n_group, L_3D, legacy, prefix)[0][-((TAIL_FLOATS + 255) // 256) * 256:]
    out = np.zeros(total, np.float32)
    out16 = out.view(np.float16)
    frag_cache = {}
    for name, first, steps, m, off, fl, hdr in segs:
        if name == "tail":
            out[off:off + fl] = f32_tail
            continue
        w, b, cols, ew = stages[name]
        if ew is None:
            ew = f16_weight_exponent(w)
        if name not in frag_cache:
            frag_cache[name] = _fragments_h(w, cols, m, ew)
        if hdr:
            out[off:off + 128] = _bias_fragment(b, m)[:128]
            out[off + 128] = np.ldexp(np.float32(1.0), -ew)
            out[off + 129] = np.float32(ew)   # the kernel reads the exponent (scales are tracked as integers)
            off += FRAG_FLOATS
        a = frag_cache[name][first:first + steps]
        out16[2 * off:2 * off + a.size] = a.reshape(-1)
    return out, cond_dim, cond_stride


WSTREAM_FORMATS = {"f32": 0, "bf16x6": 1, "f16x3": 2, "f16": 3}


def decoder_math():
    """Matrix arithmetic of the fused decoder, chosen with MNERF_DECODER_MATH:
    'f16x3' (default) fp32 operands as two range-managed fp16 terms, three products per MAC on the fp16 MFMA;
    'bf16x6' fp32 operands as three bf16 terms, six products per MAC (strict: error of an fp32 FMA chain);
    'f32' the exact-f32 MFMA;
    'f16' (opt-in FAST mode, reduced precision: outside the 1e-4 parity gate) the f16x3 stream read as plain fp16 weights, one
    product per MAC on fp16 activations with per-sample gains, fp32 accumulation (ping-pong decoder only)."""
    import os
    m = os.environ.get("MNERF_DECODER_MATH", "f16x3")
    if m not in WSTREAM_FORMATS:
        raise ValueError(f"MNERF_DECODER_MATH={m!r}: expected one of {sorted(WSTREAM_FORMATS)}")
    return m


def pack_for_math(math):
    return {"f32": pack_wstream, "bf16x6": pack_wstream16, "f16x3": pack_wstream_h, "f16": pack_wstream_h}[math]


def raytrans_table(n_samples, d_hid=16):
    """Sinusoid table of the ray transformer, float64 -> float32 as the reference builds it
    (cond_nerf.py:118-127)."""
    pos = np.arange(n_samples, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    tab = pos / np.power(10000, 2 * (j // 2) / d_hid)
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return tab.astype(np.float32)


def pack_small(sd, n_samples, raytrans_posenc, prefix="nerf_dec."):
    """Parameters that are not matrix operands: [0:16) LayerNorm weight, [16:32) LayerNorm bias of
    the ray transformer, then the [S,16] sinusoid table when ``raytrans_posenc``."""

    def g(name):
        v = sd[prefix + name]
        return (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)).astype(np.float32).reshape(-1)

    out = np.zeros(SMALL_FIXED + (n_samples * 16 if raytrans_posenc else 0), np.float32)
    out[0:16] = g("ray_attention.layer_norm.weight")
    out[16:32] = g("ray_attention.layer_norm.bias")
    if raytrans_posenc:
        out[SMALL_FIXED:] = raytrans_table(n_samples).reshape(-1)
    return out


# ----------------------------------------------------------------------------- module


class _RayAttentionParams(nn.Module):
    """Parameter holder with the keys of MultiHeadAttention(4, 16, 4, 4)
    (ray_transformer.py:32-47); evaluated inside csrc/decoder.hip."""

    def __init__(self):
        super().__init__()
        self.w_qs = nn.Linear(16, 16, bias=False)
        self.w_ks = nn.Linear(16, 16, bias=False)
        self.w_vs = nn.Linear(16, 16, bias=False)
        self.fc = nn.Linear(16, 16, bias=False)
        self.layer_norm = nn.LayerNorm(16, eps=1e-6)


class CondNeRF(nn.Module):
    """Same constructor contract and parameter names as the reference's CondNeRF
    (cond_nerf.py:11-50).  ``forward`` keeps the reference signature and runs the fused HIP decoder on
    caller-supplied sample coordinates (``mnerf_decoder_samples``); ``MatchNeRF.render`` uses the ray-chunk form
    that also rebuilds the rays and composites in-kernel.  ``composite`` keeps the reference signature
    (nerf.py:101) and runs the K5 HIP kernel."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        dec, nerf = opt.decoder, opt.nerf
        W, D = dec.net_width, dec.net_depth
        skip = list(dec.skip)
        if not nerf.view_dep:
            raise NotImplementedError("nerf.view_dep=false is unusable in the reference as well "
                                      "(cond_nerf.py:47-50 references views_linears unconditionally)")
        if W != 128 or D != 6 or skip != [4]:
            raise NotImplementedError(
                f"fused decoder kernel is built for net_width=128, net_depth=6, skip=[4] "
                f"(configs/base.yaml:30-32); got {W}, {D}, {skip}")
        L_3D = dec.posenc.L_3D if dec.posenc else 0
        L_view = dec.posenc.L_view if dec.posenc else 0
        if L_view != 0:
            raise NotImplementedError("decoder.posenc.L_view > 0 is not built (base.yaml:35 uses 0)")
        if dec.raytrans_act not in ("ReLU", "ELU"):
            raise NotImplementedError(f"decoder.raytrans_act={dec.raytrans_act}")
        self.L_3D = L_3D
        d3 = 3 + 6 * L_3D
        self.cond_dim = int(sum(opt.encoder.cos_n_group)) + opt.n_src_views * 4
        self.pts_linears = nn.ModuleList(
            [nn.Linear(d3, W)] + [nn.Linear(W + d3, W) if i in skip else nn.Linear(W, W) for i in range(D - 1)])
        self.pts_bias = nn.Linear(self.cond_dim, W)
        act = getattr(nn, dec.raytrans_act)
        self.views_linears = nn.ModuleList([nn.Linear(3 + W, W // 2)])
        self.alpha_linear = nn.Sequential(nn.Linear(W, 16), act())
        self.ray_attention = _RayAttentionParams()
        self.out_alpha_linear = nn.Sequential(nn.Linear(16, 16), act(), nn.Linear(16, 1), nn.ReLU())
        self.feature_linear = nn.Linear(W, W)
        self.rgb_linear = nn.Linear(W // 2, 3)
        for m in [*self.pts_linears, *self.views_linears, self.feature_linear, self.alpha_linear[0], self.rgb_linear]:
            nn.init.kaiming_normal_(m.weight.data)   # cond_nerf.py:102-106
            nn.init.zeros_(m.bias.data)
        self._packed = None  # (key, wstream, small)

    # -- packing cache -----------------------------------------------------------------
    def _pack_key(self, n_samples):
        ver = tuple(int(p._version) for p in self.parameters())
        ptr = tuple(int(p.data_ptr()) for p in self.parameters())
        return (ver, ptr, n_samples, bool(self.opt.decoder.raytrans_posenc), bool(self.opt.nerf.legacy_coord),
                self.math_for(n_samples, self._training_now()))

    def _training_now(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def math_for(self, n_samples, training=False):
        """the matrix path the kernel runs for S samples per ray (MNERF_DECODER_MATH).  f16x3 / bf16x6 / f32 exist for every
        S <= 256.  The one-product fast mode "f16" exists in the ping-pong decoder at S <= 128 only and has no backward of its
        own (mnerf_decoder_backward re-evaluates the forward in fp32: gradients of another function), so a request for it falls
        back to the parity path f16x3 - with one warning - for longer rays and under autograd, instead of failing with
        MNERF_E_UNSUPPORTED deep inside a launch."""
        math = decoder_math()
        if math == "f16" and (n_samples > 128 or training):
            if not getattr(self, "_warned_f16", False):
                import warnings
                warnings.warn(f"MNERF_DECODER_MATH=f16 (one-product fast mode) does not exist for "
                              f"{'training' if training else f'sample_intvs={n_samples} > 128'}: using f16x3")
                self._warned_f16 = True
            return "f16x3"
        return math

    def packed(self, n_samples, device):
        """(wstream, small, cond_stride, wstream_format) for the HIP kernel; re-packed when any
        parameter changed (load_state_dict / optimizer step), S or MNERF_DECODER_MATH changed."""
        key = self._pack_key(n_samples)
        if self._packed is None or self._packed[0] != key or self._packed[1].device != torch.device(device):
            math = self.math_for(n_samples, self._training_now())
            if math in ("f16x3", "f16") and self.pts_bias.weight.is_cuda and self.pts_bias.weight.device == torch.device(device):
                # parameters on the GPU (every training iteration re-packs after the optimizer step): the stream is assembled
                # THERE, without a device->host copy (packing.DecoderPacker, bit-identical to pack_wstream_h)
                self._packed = (key,) + self._packed_on_device(n_samples, torch.device(device)) + (WSTREAM_FORMATS[math],)
                return self._packed[1], self._packed[2], self._packed[3], self._packed[4]
            sd = {"nerf_dec." + k: v for k, v in self.state_dict().items()}
            ws, cond_dim, cond_stride = pack_for_math(math)(sd, self.opt.n_src_views, list(self.opt.encoder.cos_n_group),
                                                            self.L_3D, bool(self.opt.nerf.legacy_coord))
            assert cond_dim == self.cond_dim
            limit = 64 if math == "f32" else 96  # MNERF_COND_STRIDE_MAX_F32 / MNERF_COND_STRIDE_MAX
            if cond_stride > limit:
                raise NotImplementedError(
                    f"cond_dim={cond_dim} (n_src_views={self.opt.n_src_views}) needs {cond_stride} floats per sample; "
                    f"the {math} decoder stream supports {limit}")
            small = pack_small(sd, n_samples, bool(self.opt.decoder.raytrans_posenc))
            self._packed = (key, torch.from_numpy(ws).to(device), torch.from_numpy(small).to(device), cond_stride,
                            WSTREAM_FORMATS[math])
        return self._packed[1], self._packed[2], self._packed[3], self._packed[4]

    def _packed_on_device(self, n_samples, device):
        from . import packing
        legacy = bool(self.opt.nerf.legacy_coord)
        pkey = (self.opt.n_src_views, tuple(self.opt.encoder.cos_n_group), self.L_3D, legacy, str(device))
        if getattr(self, "_packer", None) is None or self._packer[0] != pkey:
            self._packer = (pkey, packing.DecoderPacker(self, self.opt.n_src_views, list(self.opt.encoder.cos_n_group), self.L_3D,
                                                        legacy, device), {})
        _, pk, tables = self._packer
        assert pk.cond_dim == self.cond_dim
        if pk.cond_stride > 96:  # MNERF_COND_STRIDE_MAX
            raise NotImplementedError(f"cond_dim={pk.cond_dim} (n_src_views={self.opt.n_src_views}) needs {pk.cond_stride} floats "
                                      f"per sample; the f16x3 decoder stream supports 96")
        ln = self.ray_attention.layer_norm
        parts = [ln.weight.detach().float().reshape(-1), ln.bias.detach().float().reshape(-1)]
        if bool(self.opt.decoder.raytrans_posenc):
            if n_samples not in tables:
                tables[n_samples] = torch.from_numpy(raytrans_table(n_samples).reshape(-1)).to(device)
            parts.append(tables[n_samples])
        return pk.pack(), torch.cat(parts), pk.cond_stride

    def decoder_struct(self, n_samples, device, setbg_opaque=False):
        """C-ABI ``mnerf_decoder`` for S samples per ray (the packed tensors stay cached on the module)."""
        from . import hip
        ws, small, cond_stride, wfmt = self.packed(n_samples, device)
        d = hip.Decoder()
        d.wstream, d.wstream_floats, d.small_ = ws.data_ptr(), ws.numel(), small.data_ptr()
        d.n_views, d.cond_dim, d.cond_stride = self.opt.n_src_views, self.cond_dim, cond_stride
        d.wstream_format = wfmt
        d.L_3D = self.L_3D
        dec, nerf = self.opt.decoder, self.opt.nerf
        d.raytrans_posenc, d.raytrans_elu = int(bool(dec.raytrans_posenc)), int(dec.raytrans_act == "ELU")
        d.density_maskfill, d.wo_render_interval = int(bool(dec.density_maskfill)), int(bool(nerf.wo_render_interval))
        d.setbg_opaque = int(bool(setbg_opaque))
        return d

    def forward(self, opt, points_3D, ray_unit=None, cond_info=None, mode=None):
        """cond_nerf.py:52-100 with the reference's signature: points_3D [B,R,S,3] (coordinates w.r.t. source view
        0), ray_unit [B,R,S,3] (or [B,R,3]), cond_info = dict(feat_info, color_info, mask_info) [B,R,S,*]
        -> rgb [B,R,S,3], density [B,R,S].  One launch of the fused HIP decoder per batch element
        (mnerf_decoder_samples): no eager op chain, no autograd (training goes through MatchNeRF.render)."""
        from . import hip
        if ray_unit is None or cond_info is None:
            raise ValueError("CondNeRF.forward needs ray_unit and cond_info (nerf.view_dep is always true here)")
        if not points_3D.is_cuda:
            raise RuntimeError("CondNeRF.forward: the HIP decoder needs CUDA tensors (no CPU fallback)")
        b, r, s, _ = points_3D.shape
        dev = points_3D.device
        dec = self.decoder_struct(s, dev)
        if ray_unit.dim() == 3:
            ray_unit = ray_unit[:, :, None, :].expand(b, r, s, 3)
        cond = torch.zeros(b, r * s, dec.cond_stride, device=dev)
        c = torch.cat([cond_info["feat_info"], cond_info["color_info"], cond_info["mask_info"]], -1)
        if c.shape[-1] != self.cond_dim:
            raise ValueError(f"cond_info has {c.shape[-1]} channels, the decoder was built for {self.cond_dim}")
        cond[..., :self.cond_dim] = c.reshape(b, r * s, self.cond_dim).float()
        cond[..., self.cond_dim] = 1.0
        rgb, sigma = [], []
        with torch.no_grad():
            for i in range(b):
                o = hip.decoder_samples(dec, points_3D[i].float().contiguous(), ray_unit[i].float().contiguous(), cond[i],
                                        legacy_coord=bool(opt.nerf.legacy_coord))
                rgb.append(o[0])
                sigma.append(o[1])
        return torch.stack(rgb, 0), torch.stack(sigma, 0)

    def composite(self, opt, ray, rgb_samples, density_samples, depth_samples, setbg_opaque):
        """nerf.py:101-124 with the reference signature; tensors [B,R,S,*] on the GPU -> rgb [B,R,3], depth [B,R,1],
        opacity [B,R,1], prob [B,R,S,1]."""
        from . import hip
        b, r, s = density_samples.shape
        ray_len = None if opt.nerf.wo_render_interval else ray.norm(dim=-1).reshape(b * r).contiguous()
        rgb, depth, opacity, prob = hip.composite(
            rgb_samples.reshape(b * r, s, 3).contiguous(), density_samples.reshape(b * r, s).contiguous(),
            depth_samples.reshape(b * r, s).contiguous(), ray_len,
            wo_render_interval=bool(opt.nerf.wo_render_interval), setbg_opaque=bool(setbg_opaque), want_prob=True)
        return rgb.reshape(b, r, 3), depth.reshape(b, r, 1), opacity.reshape(b, r, 1), prob.reshape(b, r, s, 1)
