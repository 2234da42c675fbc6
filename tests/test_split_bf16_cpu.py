"""The split-bf16 arithmetic of the backward kernels (gemm_b6_kernel, wa_bwd_*_split_kernel<.., 3>, and the f16x3 twin <.., 2>; matchnerf_amd/csrc/gemm_f32.hpp,
window_attention_backward.hip) restated in numpy — no GPU:
* three round-to-nearest-even bf16 terms taken as the kernels take them (term, residual, term, residual, term) represent an fp32
  value EXACTLY for |x| >= 2^-100 (below that the third term drops into fp32's subnormals and loses bits; the value is then still
  within 2^-16 relative: the first two terms);
* the six term products the kernels keep (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi) differ from the exact product by what the
  three dropped ones hold, at most a few 2^-24 of it: a dot product accumulated from them is as accurate as an fp32 FMA chain;
* the row permutation of the attention backward's [channel][row] tiles is the accumulator register map: position 16 s + 8 h + j of a
  line must hold the row that register 8 s + j of lane half h holds."""
import numpy as np


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 (what v_cvt_pk_bf16_f32 does for normal values)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_rne(r2)
    return hi, mid, lo


def test_three_bf16_terms_are_exact():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp2(rng.integers(-90, 100, 200000))).astype(np.float32)
    x = x[np.abs(x) >= 2.0 ** -100]
    x = np.concatenate([x, np.float32([0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.1754944e-38 * 2 ** 30, 0.1, 1 / 3])])
    hi, mid, lo = split3(x)
    total = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64))          # exact: 3 x 8 significand bits cover fp32's 24
    for t in (hi, mid, lo):                                       # every term is a bf16 value
        assert np.all((t.view(np.uint32) & 0xFFFF) == 0)
    nz = x != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(hi[nz]) * 2.0 ** -8 * 1.01) and np.all(np.abs(lo[nz]) <= np.abs(hi[nz]) * 2.0 ** -16 * 1.01)
    tiny = (rng.standard_normal(20000) * np.exp2(rng.integers(-115, -100, 20000))).astype(np.float32)  # the graceful end
    tiny = tiny[np.abs(tiny) >= 2.0 ** -118]    # (normal fp32 values)
    th, tm, tl = split3(tiny)
    err = np.abs(th.astype(np.float64) + tm.astype(np.float64) + tl.astype(np.float64) - tiny.astype(np.float64))
    assert np.all(err <= np.abs(tiny.astype(np.float64)) * 2.0 ** -15)


def test_six_products_are_fp32_grade():
    rng = np.random.default_rng(1)
    n, k = 512, 256
    a = (rng.standard_normal((n, k)) * np.exp2(rng.integers(-8, 8, (n, 1)))).astype(np.float32)
    b = (rng.standard_normal((n, k)) * np.exp2(rng.integers(-8, 8, (n, 1)))).astype(np.float32)
    ah, am, al = (t.astype(np.float64) for t in split3(a))
    bh, bm, bl = (t.astype(np.float64) for t in split3(b))
    kept = al * bh + ah * bl + am * bm + am * bh + ah * bm + ah * bh   # the kernels' six, smallest first
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(kept - exact) / np.maximum(np.abs(exact), 1e-300)
    assert rel.max() < 4 * 2.0 ** -24                                  # dropped: mid.lo, lo.mid, lo.lo <= 3 x 2^-24
    dot6 = kept.sum(1)                                                  # (the matrix pipe accumulates in fp32: + its rounding)
    dot32 = np.zeros(n, np.float32)
    for j in range(k):
        dot32 = (a[:, j] * b[:, j] + dot32).astype(np.float32)          # an fp32 multiply-add chain
    ref = exact.sum(1)
    scale = (np.abs(a.astype(np.float64)) * np.abs(b.astype(np.float64))).sum(1)
    assert (np.abs(dot6 - ref) / scale).max() < 2.0 ** -23
    assert (np.abs(dot6 - ref) / scale).max() <= (np.abs(dot32.astype(np.float64) - ref) / scale).max()


def wb_row(r, half):
    """window_attention_backward.hip: register r of lane (n, half) of a 32 x 32 accumulator block holds row f(r, half)"""
    return (r & 3) + 8 * (r >> 2) + 4 * half


def test_role2_row_permutation_is_the_accumulator_register_map():
    seen = set()
    for s in range(2):            # K16-step
        for h in range(2):        # lane half
            for j in range(8):    # slot of the 8 consecutive k a lane supplies
                row = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)      # wb6_build_r2 stores this row at position 16 s + 8 h + j
                assert row == wb_row(8 * s + j, h)                  # wb6_chain_product pairs it with register 8 s + j
                seen.add((16 * s + 8 * h + j, row))
    assert sorted(p for p, _ in seen) == list(range(32)) and sorted(r for _, r in seen) == list(range(32))  # a permutation


def test_lds_strides_are_conflict_free_for_16_lane_groups():
    """a ds_read_b128 serves 16 lanes per pass: their 4-dword accesses must fall into distinct banks (64 banks x 4 B)"""
    for stride_bytes in (272, 80):                      # R1 row stride, R2 line stride
        banks = set()
        for lane in range(16):
            first = (lane * stride_bytes // 4) % 64
            for d in range(4):
                banks.add((first + d) % 64)
        assert len(banks) == 64, stride_bytes


def mfma6(a_terms, b_terms):
    """one K16-step of wb6_mfma6: A [32,16] x B [16,32] from the six kept term products, smallest first, fp32 accumulation"""
    acc = np.zeros((a_terms[0].shape[0], b_terms[0].shape[1]), np.float32)
    for ta, tb in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
        acc = (acc + (a_terms[ta].astype(np.float64) @ b_terms[tb].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc


def split2h(x, mult):
    """split8h (split_f16.hpp): x * mult (a power of two) as fp16 hi (nearest even) + fp16 lo of the exact fp32 residual"""
    xm = (x.astype(np.float32) * np.float32(mult)).astype(np.float32)
    hi = xm.astype(np.float16)
    lo = (xm - hi.astype(np.float32)).astype(np.float32).astype(np.float16)
    assert np.isfinite(hi.astype(np.float32)).all()           # the gain keeps every operand inside fp16's range
    return hi.astype(np.float32), lo.astype(np.float32)


def mfma3(a_terms, b_terms):
    """one K16-step of the f16x3 form: lo.hi + hi.lo + hi.hi, smallest first, fp32 accumulation"""
    acc = np.zeros((a_terms[0].shape[0], b_terms[0].shape[1]), np.float32)
    for ta, tb in ((1, 0), (0, 1), (0, 0)):
        acc = (acc + (a_terms[ta].astype(np.float64) @ b_terms[tb].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc


def tensor_gain(x):
    """wbs_tensor_gain: max|x| * gain in [2^14, 2^15)"""
    m = float(np.abs(x).max())
    return 2.0 ** (15 - (np.frexp(np.float32(m))[1] if m > 0 else 0))


def _dk_dv_restated(q, k, v, go, form):
    """wa_bwd_dkv_split_kernel<0 | 1, NT> for one workgroup, restated: a 96-token window (no shift) against a 128-row stationary
    key block (32 rows beyond the window), three 32-query streaming tiles; role-1 fragments [row][16 channels of step u], role-2
    lines with the permuted rows, P / dS taken register by register from the accumulator map; the forward's row statistics as
    input.  form: "bf16x6" (three bf16 terms, no gains) or "f16x3" (two fp16 terms, the kernel's gains).  Returns the workgroup's
    dV, dK and the dense float64 gradients."""
    lw, c = q.shape
    scale, log2e = 1.0 / np.sqrt(c), 1.4426950408889634
    s64 = (q.astype(np.float64) @ k.astype(np.float64).T) * scale
    p64 = np.exp(s64 - s64.max(1, keepdims=True))
    p64 /= p64.sum(1, keepdims=True)
    o64 = p64 @ v.astype(np.float64)
    dp64 = go.astype(np.float64) @ v.astype(np.float64).T
    d64 = (go.astype(np.float64) * o64).sum(1, keepdims=True)
    ds64 = p64 * (dp64 - d64)
    dv_ref, dk_ref = p64.T @ go.astype(np.float64), (ds64.T @ q.astype(np.float64)) * scale
    # what the forward publishes (log2 domain) and the row dot products
    s2 = s64 * log2e
    row_m = s2.max(1).astype(np.float32)
    row_l = np.exp2(s2 - row_m[:, None]).sum(1).astype(np.float32)
    row_d = d64[:, 0].astype(np.float32)
    if form == "bf16x6":
        g_q = g_k = g_v = g_do = p_gain = ds_gain = 1.0
        terms, mfma = (lambda x, mult: list(split3(x))), mfma6
    else:
        g_q, g_k, g_v, g_do = (tensor_gain(t) for t in (q, k, v, go))
        p_gain, ds_gain = 2.0 ** 15, 2.0 ** -23                # WBS_P_GAIN, WBS_DS_GAIN
        terms, mfma = (lambda x, mult: list(split2h(x, mult))), mfma3
    s_scale, d_gain = np.float32(scale / (g_q * g_k)), np.float32(g_do * g_v)

    dk, dv = np.zeros((128, c), np.float32), np.zeros((128, c), np.float32)
    for wave in range(4):                                   # a wave owns 32 keys
        key_rows = np.arange(32 * wave, 32 * wave + 32)
        k_ok = key_rows < lw
        kblk = np.where(k_ok[:, None], k[np.minimum(key_rows, lw - 1)], 0.0).astype(np.float32)
        vblk = np.where(k_ok[:, None], v[np.minimum(key_rows, lw - 1)], 0.0).astype(np.float32)
        kf, vf = terms(kblk, g_k), terms(vblk, g_v)         # stationary fragments: [32 keys][128 ch] per term
        res_k, res_v = np.zeros((c, 32), np.float32), np.zeros((c, 32), np.float32)  # dK^T / dV^T [channel][key]
        for qt in range(3):                                 # streaming tiles of 32 queries
            rows = np.arange(32 * qt, 32 * qt + 32)
            qtile, dotile = q[rows], go[rows]
            q_r1, do_r1 = terms(qtile, g_q), terms(dotile, g_do)
            s_blk, dp_blk = np.zeros((32, 32), np.float32), np.zeros((32, 32), np.float32)
            for u in range(8):                              # S = Q K^T and dP = dO V^T over the channels, one K16-step at a time
                ch = slice(16 * u, 16 * u + 16)
                s_blk = (s_blk + mfma([t[:, ch] for t in q_r1], [t[:, ch].T for t in kf])).astype(np.float32)
                dp_blk = (dp_blk + mfma([t[:, ch] for t in do_r1], [t[:, ch].T for t in vf])).astype(np.float32)
            sc = (s_blk * s_scale) * np.float32(log2e)
            p = np.where(k_ok[None, :], np.exp2(sc - row_m[rows][:, None]) / row_l[rows][:, None], 0.0).astype(np.float32)
            ds = (p * (dp_blk - (row_d[rows] * d_gain)[:, None])).astype(np.float32)   # in the units of the dP accumulator
            # role 2: [channel][position], position 16 s + 8 h + j holds row 16 s + 8 (j >> 2) + 4 h + (j & 3)
            pos_row = np.array([16 * s + 8 * (j >> 2) + 4 * h + (j & 3) for s in range(2) for h in range(2) for j in range(8)])
            do_r2 = [t[pos_row].T for t in terms(dotile, g_do)]   # [term][channel][32 positions]
            q_r2 = [t[pos_row].T for t in terms(qtile, g_q)]
            for s in range(2):                              # chain products: the B operand is P / dS register by register
                regs = np.array([wb_row(8 * s + j, h) for h in range(2) for j in range(8)])   # rows in K16 order (h, j)
                pos = slice(16 * s, 16 * s + 16)
                res_v = (res_v + mfma([t[:, pos] for t in do_r2], terms(p[regs], p_gain))).astype(np.float32)
                res_k = (res_k + mfma([t[:, pos] for t in q_r2], terms(ds[regs], ds_gain))).astype(np.float32)
        dv[key_rows] = res_v.T * np.float32(1.0 / (g_do * p_gain))
        dk[key_rows] = res_k.T * np.float32(scale / (g_q * (d_gain * ds_gain if form == "f16x3" else 1.0)))
    return dv, dk, dv_ref, dk_ref


def test_dk_dv_dataflow_of_the_split_bf16_attention_backward():
    """Judge: the dense float64 gradients."""
    rng = np.random.default_rng(7)
    lw, c = 96, 128
    q, k, v, go = (rng.standard_normal((lw, c)).astype(np.float32) * s for s in (0.6, 0.8, 1.0, 1.0))
    dv, dk, dv_ref, dk_ref = _dk_dv_restated(q, k, v, go, "bf16x6")
    for got, ref in ((dv[:lw], dv_ref), (dk[:lw], dk_ref)):
        assert np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max() < 3e-6
    assert np.all(dv[lw:] == 0) and np.all(dk[lw:] == 0)   # rows beyond the window: nothing


def test_dk_dv_dataflow_of_the_split_fp16_attention_backward_and_its_gains():
    """The f16x3 form with the kernel's gains (one per tensor from its maximum, 2^15 for P, 2^-23 for dS): no operand leaves fp16's
    range (split2h asserts it) and the gradients are fp32-grade — also with operand scales far from 1 and gradient rows spread over
    sixteen binades (dK / dV sum over the queries: judged against the tensor's maximum, as the GPU test does)."""
    rng = np.random.default_rng(8)
    lw, c = 96, 128
    for scales, spread in (((0.6, 0.8, 1.0, 1.0), 0), ((3.0, 0.05, 40.0, 1e-3), 16), ((200.0, 1e-3, 1e-4, 3e4), 8)):
        q, k, v, go = (rng.standard_normal((lw, c)).astype(np.float32) * np.float32(s) for s in scales)
        go = (go * np.exp2(-rng.integers(0, spread + 1, (lw, 1)))).astype(np.float32)
        dv, dk, dv_ref, dk_ref = _dk_dv_restated(q, k, v, go, "f16x3")
        for got, ref in ((dv[:lw], dv_ref), (dk[:lw], dk_ref)):
            assert np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max() < 3e-6, scales
        assert np.all(dv[lw:] == 0) and np.all(dk[lw:] == 0)
