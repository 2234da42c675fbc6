"""CPU restatement of the K6 backward's dataflow (csrc/window_attention_backward.hip) in numpy, checked against float64 autograd
through the oracle's window attention.  It follows the kernels step by step — win_token's roll / window / wrap-region
arithmetic, 64-row tiles with zero rows beyond the window, the online-softmax statistics pass in the log2 domain merged over the
two key halves, P recomputed from (max, sum), dS = P (dP - D), and the accumulator -> operand chain in which K-step r pairs
rows f(r, 0) and f(r, 1) — so that an index or formula error shows up without a GPU."""
import math

import numpy as np
import pytest
import torch

from oracle import matchnerf_oracle as O

T = 64
LOG2E = 1.4426950408889634


def win_token(h, w, wh, ww, sh, sw, wy, wx, li):
    """wa_common.hpp: window-local index -> token id in the un-rolled sequence and the wrap region of its rolled position"""
    ly, lx = divmod(li, ww)
    ry, rx = wy * wh + ly, wx * ww + lx
    oy, ox = (ry + sh) % h, (rx + sw) % w
    regy = int(ry >= h - wh) + int(ry >= h - sh)
    regx = int(rx >= w - ww) + int(rx >= w - sw)
    return oy * w + ox, regy * 3 + regx


def f_row(r, half):
    return (r & 3) + 8 * (r >> 2) + 4 * half


def chain_product(x_rows_by_ch, s_block):
    """out^T[ch][col] += sum over the block's 32 rows, visited in the kernel's K-step order (r, half)"""
    out = np.zeros((x_rows_by_ch.shape[0], s_block.shape[1]), np.float64)
    for r in range(16):
        for half in (0, 1):
            row = f_row(r, half)
            out += np.outer(x_rows_by_ch[:, row], s_block[row])
    return out


def backward_emulated(q, k, v, out, g_out, h, w, splits, shifted):
    b, n, c = q.shape
    wh, ww = h // splits, w // splits
    sh, sw = (wh // 2, ww // 2) if (shifted and splits > 1) else (0, 0)
    do_shift = shifted and splits > 1
    lw = wh * ww
    n_tiles = (lw + T - 1) // T
    scale = 1.0 / math.sqrt(c)
    gq, gk, gv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
    row_d = (g_out * out).sum(-1)                                    # wa_bwd_rowdot_kernel
    row_m, row_l = np.zeros((b, n)), np.zeros((b, n))

    def tile(src, seq, wy, wx, i0):
        rows, toks, regs = np.zeros((T, c)), -np.ones(T, int), np.zeros(T, int)
        for r in range(T):
            if i0 + r < lw:
                toks[r], regs[r] = win_token(h, w, wh, ww, sh, sw, wy, wx, i0 + r)
                rows[r] = src[seq, toks[r]]
        return rows, toks, regs

    def scores(dot, qreg, kreg, kvalid):                             # [64 q, 64 k], log2 domain
        s = (dot * scale + np.where(do_shift & (qreg[:, None] != kreg[None, :]), -100.0, 0.0)) * LOG2E
        return np.where(kvalid[None, :], s, -np.inf)

    for seq in range(b):
        for wy in range(splits):
            for wx in range(splits):
                # ---- wa_bwd_dq_kernel: one workgroup per query tile
                for qt in range(n_tiles):
                    qr, qtok, qreg = tile(q, seq, wy, wx, qt * T)
                    dor, _, _ = tile(g_out, seq, wy, wx, qt * T)
                    run_m = np.full((2, T), -np.inf)                 # per key half (the two wk waves)
                    run_l = np.zeros((2, T))
                    for kt in range(n_tiles):
                        kr, ktok, kreg = tile(k, seq, wy, wx, kt * T)
                        s = scores(qr @ kr.T, qreg, kreg, ktok >= 0)
                        for wk in (0, 1):
                            blk = s[:, 32 * wk:32 * wk + 32]
                            m_new = np.maximum(run_m[wk], blk.max(1))
                            ok = m_new > -np.inf
                            with np.errstate(invalid="ignore"):
                                add = np.where(ok, np.exp2(blk - np.where(ok, m_new, 0.0)[:, None]).sum(1), 0.0)
                                old = np.where(run_m[wk] > -np.inf, run_l[wk] * np.exp2(run_m[wk] - np.where(ok, m_new, 0.0)), 0.0)
                            run_l[wk] = np.where(ok, old + add, run_l[wk])
                            run_m[wk] = np.where(ok, m_new, run_m[wk])
                    m = np.maximum(run_m[0], run_m[1])
                    with np.errstate(invalid="ignore"):
                        l = sum(np.where(run_m[i] > -np.inf, run_l[i] * np.exp2(run_m[i] - m), 0.0) for i in (0, 1))
                    for r in range(T):
                        if qtok[r] >= 0:
                            row_m[seq, qtok[r]], row_l[seq, qtok[r]] = m[r], l[r]
                    d = np.array([row_d[seq, t] if t >= 0 else 0.0 for t in qtok])
                    dq_t = np.zeros((c, T))
                    for kt in range(n_tiles):
                        kr, ktok, kreg = tile(k, seq, wy, wx, kt * T)
                        vr, _, _ = tile(v, seq, wy, wx, kt * T)
                        s = scores(qr @ kr.T, qreg, kreg, ktok >= 0)
                        p = np.where((qtok >= 0)[:, None], np.exp2(s - m[:, None]) / l[:, None], 0.0)
                        ds = p * (dor @ vr.T - d[:, None])           # [q, k]
                        for wk in (0, 1):                            # dQ^T[ch][q] += K^T[ch][key] dS^T[key][q]
                            dq_t += chain_product(kr[32 * wk:32 * wk + 32].T, ds[:, 32 * wk:32 * wk + 32].T)
                    for r in range(T):
                        if qtok[r] >= 0:
                            gq[seq, qtok[r]] = dq_t[:, r] * scale
                # ---- wa_bwd_dkv_kernel: one workgroup per key tile
                for kt in range(n_tiles):
                    kr, ktok, kreg = tile(k, seq, wy, wx, kt * T)
                    vr, _, _ = tile(v, seq, wy, wx, kt * T)
                    dk_t, dv_t = np.zeros((c, T)), np.zeros((c, T))
                    for qt in range(n_tiles):
                        qr, qtok, qreg = tile(q, seq, wy, wx, qt * T)
                        dor, _, _ = tile(g_out, seq, wy, wx, qt * T)
                        m = np.array([row_m[seq, t] if t >= 0 else 0.0 for t in qtok])
                        il = np.array([1.0 / row_l[seq, t] if t >= 0 else 0.0 for t in qtok])
                        d = np.array([row_d[seq, t] if t >= 0 else 0.0 for t in qtok])
                        s = scores(qr @ kr.T, qreg, kreg, ktok >= 0)
                        p = np.where((ktok >= 0)[None, :], np.exp2(s - m[:, None]) * il[:, None], 0.0)
                        ds = p * (dor @ vr.T - d[:, None])
                        for wq in (0, 1):
                            dv_t += chain_product(dor[32 * wq:32 * wq + 32].T, p[32 * wq:32 * wq + 32])
                            dk_t += chain_product(qr[32 * wq:32 * wq + 32].T, ds[32 * wq:32 * wq + 32])
                    for r in range(T):
                        if ktok[r] >= 0:
                            gk[seq, ktok[r]] = dk_t[:, r] * scale
                            gv[seq, ktok[r]] = dv_t[:, r]
    return gq, gk, gv


@pytest.mark.parametrize("b,h,w,splits,shifted", [(1, 8, 12, 2, True), (1, 8, 12, 2, False), (2, 6, 10, 1, False), (1, 16, 20, 2, True)])
def test_emulated_backward_matches_float64_autograd(b, h, w, splits, shifted):
    gen = torch.Generator().manual_seed(h * 100 + w + splits)
    n, c = h * w, 128
    q = (torch.randn(b, n, c, generator=gen) * 0.6).double().requires_grad_(True)
    k = (torch.randn(b, n, c, generator=gen) * 0.8).double().requires_grad_(True)
    v = torch.randn(b, n, c, generator=gen).double().requires_grad_(True)
    g = torch.randn(b, n, c, generator=gen).double()
    out = O.window_attention(q, k, v, h, w, splits, shifted)
    (out * g).sum().backward()
    gq, gk, gv = backward_emulated(q.detach().numpy(), k.detach().numpy(), v.detach().numpy(), out.detach().numpy(), g.numpy(),
                                   h, w, splits, shifted)
    for got, ref in ((gq, q.grad), (gk, k.grad), (gv, v.grad)):
        assert np.abs(got - ref.numpy()).max() <= 1e-10 * float(ref.abs().max())
