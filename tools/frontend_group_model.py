"""Index arithmetic of frontend_group_kernel (passiveradar_amd/csrc/frontend.hip), replayed in NumPy.

The kernel gives every thread `up` consecutive outputs m = up N + q and writes the polyphase sum against the input
offset r from dn N:   y[up N + q] = sum_r hz[s_q - up r] xe[dn N + r],   s_q = (q + n_pre_remove) dn.
The tap table T[row][q] (row = r_hi - r), the LDS offsets roff[row] (one pad sample per dn when dn is even), the window
of a workgroup of 64 groups and the split of the rows over its wavefronts are made here exactly as
prc_frontend_plan_create / the kernel make them; tests/test_host_logic.py checks the result against the resampler the
kernel replaces for several ratios, so a slip in the tables shows on the CPU and not as a wrong number on the GPU."""
import numpy as np

G = 64          # groups per workgroup (FEG_G)
W = 8           # wavefronts per workgroup (FEG_WAVES): they split the rows


SEGS = 10       # FEG_SEGS: segments per wavefront


def split_rows(ntaps, up, dn, n_pre_remove, balance=True):
    """The segments each wavefront works through, as prc_frontend_plan_create deals them (round 5).  A trip is two
    consecutive rows; its column window is the narrowest prefix [0, w) or suffix [nq - w, nq) that holds what its rows reach
    (code = w, or w | 0x100 for a suffix); consecutive trips with the same window form a segment; trips go to the W
    wavefronts in order, cut where the running cost (2 w + 8 per trip) passes the next W-th of the total.
    Returns (segments[W] = [(row0, ntrips, code)], rows_total, r_hi, nq)."""
    J = -(-ntaps // up)
    s0, sl = n_pre_remove * dn, (up - 1 + n_pre_remove) * dn
    r_hi, r_lo = sl // up, s0 // up - (J - 1)
    nrows = r_hi - r_lo + 1
    nq = 4 if up <= 4 else (8 if up <= 8 else (13 if up <= 13 else 16))

    def col_range(row):
        qs = [q for q in range(up) if 0 <= (q + n_pre_remove) * dn - up * (r_hi - row) < ntaps]
        return (qs[0], qs[-1] + 1) if qs else (0, 0)
    ntrip = (nrows + 1) // 2 if balance else W * ((((nrows + W - 1) // W + 1) & ~1) // 2)
    code, cost = [], []
    for t in range(ntrip):
        lo, hi = nq, 0
        for r in range(2 * t, min(2 * t + 2, nrows)):
            l, h = col_range(r)
            if h > l:
                lo, hi = min(lo, l), max(hi, h)
        c = nq
        if balance:
            c = 1 if hi <= lo else (hi if hi <= nq - lo else (nq - lo) | 0x100)
            if (c & 0xff) >= nq:
                c = nq
        code.append(c)
        cost.append(2 * (c & 0xff) + 8)
    total, t, run, segs = sum(cost), 0, 0, []
    for w in range(W):
        until = total * (w + 1) // W
        mine = []
        while t < ntrip:
            if (run + cost[t] // 2 > until and w + 1 < W) if balance else (t >= (w + 1) * (ntrip // W)):
                break
            if mine and mine[-1][2] == code[t]:
                mine[-1][1] += 1
            elif len(mine) < SEGS:
                mine.append([2 * t, 1, code[t]])
            else:
                mine[-1][2] = nq
                mine[-1][1] += 1
            run += cost[t]
            t += 1
        segs.append([tuple(m) for m in mine])
    assert t == ntrip
    return segs, 2 * ntrip, r_hi, nq


def tables(taps, up, dn, n_pre_remove, balance=True):
    """(T [rows_total + 1][16], roff [rows_total], segments, r_first, lane_stride, pad, span): T as the plan builds it, roff
    as the kernel's scalar arithmetic produces it row by row"""
    taps = np.asarray(taps, dtype=np.float32)
    segs, rows_total, r_hi, nq = split_rows(taps.size, up, dn, n_pre_remove, balance)
    pad = 1 if dn % 2 == 0 else 0
    o_max = rows_total - 1
    span = dn * (G - 1) + rows_total
    T = np.zeros((rows_total + 1, 16), dtype=np.float32)
    roff = np.zeros(rows_total, dtype=np.int64)
    for row in range(rows_total):
        r = r_hi - row
        for q in range(up):
            idx = (q + n_pre_remove) * dn - up * r
            if 0 <= idx < taps.size:
                T[row, q] = taps[idx]
        o = o_max - row
        roff[row] = o + pad * (o // dn)
    return T, roff, (segs, nq), r_hi - o_max, dn + pad, pad, span


def window_of(code, nq):
    """(first column, width) of a segment's compile-time column window"""
    w = code & 0xff
    return (nq - w if code & 0x100 else 0), w


def run(xe_of, n_out, taps, up, dn, n_pre_remove, dtype=np.complex64, balance=True):
    """xe_of(i): the (tuned, linearly extended) input at any integer index, vectorised.  Returns y[n_out] computed the
    kernel's way: staged windows with the padded layout, per-wavefront partial sums in float32, combined in order."""
    T, roff, (segs, nq), r_first, lane_stride, pad, span = tables(taps, up, dn, n_pre_remove, balance)
    y = np.zeros(n_out, dtype=dtype)
    nwg = -(-n_out // (G * up))
    for wg in range(nwg):
        N0 = wg * G
        i_w = N0 * dn + r_first
        k = np.arange(span)
        at = k + pad * (k // dn)
        X = np.zeros(at.max() + 1, dtype=dtype)
        X[at] = xe_of(i_w + k).astype(dtype)
        part = np.zeros((W, 16, G), dtype=dtype)
        lanes = np.arange(G) * lane_stride
        for w in range(W):
            acc = np.zeros((16, G), dtype=dtype)
            for row0, ntrips, code in segs[w]:                           # only the columns this segment's rows reach
                qa, wd = window_of(code, nq)
                for row in range(row0, row0 + 2 * ntrips):
                    x = X[lanes + roff[row]]
                    acc[qa:qa + wd] = acc[qa:qa + wd] + T[row, qa:qa + wd, None].astype(dtype) * x[None, :]
            part[w] = acc
        tot = part[0]
        for w in range(1, W):
            tot = tot + part[w]                                          # [q][n], wavefront order
        o = np.arange(G * up)
        m = N0 * up + o
        ok = m < n_out
        y[m[ok]] = tot[o[ok] % up, o[ok] // up]
    return y
