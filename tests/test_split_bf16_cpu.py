"""The split-bf16 arithmetic of the backward kernels (gemm_b6_kernel, wa_bwd_*_b6_kernel; matchnerf_amd/csrc/gemm_f32.hpp,
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
