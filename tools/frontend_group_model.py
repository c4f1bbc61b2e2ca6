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


def tables(taps, up, dn, n_pre_remove):
    """(T [W rpw][16], roff [W rpw], rpw, r_first, lane_stride, pad, span): T as the plan builds it, roff as the kernel's
    scalar arithmetic produces it row by row"""
    taps = np.asarray(taps, dtype=np.float32)
    J = -(-taps.size // up)
    s0, sl = n_pre_remove * dn, (up - 1 + n_pre_remove) * dn
    r_hi, r_lo = sl // up, s0 // up - (J - 1)
    nrows = r_hi - r_lo + 1
    rpw = ((nrows + W - 1) // W + 1) & ~1
    pad = 1 if dn % 2 == 0 else 0
    o_max = W * rpw - 1
    span = dn * (G - 1) + W * rpw
    T = np.zeros((W * rpw, 16), dtype=np.float32)
    roff = np.zeros(W * rpw, dtype=np.int64)
    for row in range(W * rpw):
        r = r_hi - row
        for q in range(up):
            idx = (q + n_pre_remove) * dn - up * r
            if 0 <= idx < taps.size:
                T[row, q] = taps[idx]
        o = o_max - row
        roff[row] = o + pad * (o // dn)
    return T, roff, rpw, r_hi - o_max, dn + pad, pad, span


def run(xe_of, n_out, taps, up, dn, n_pre_remove, dtype=np.complex64):
    """xe_of(i): the (tuned, linearly extended) input at any integer index, vectorised.  Returns y[n_out] computed the
    kernel's way: staged windows with the padded layout, per-wavefront partial sums in float32, combined in order."""
    T, roff, rpw, r_first, lane_stride, pad, span = tables(taps, up, dn, n_pre_remove)
    y = np.zeros(n_out, dtype=dtype)
    nwg = -(-n_out // (G * up))
    for wg in range(nwg):
        N0 = wg * G
        i_w = N0 * dn + r_first
        k = np.arange(span)
        at = k + pad * (k // dn)
        X = np.zeros(at.max() + 1, dtype=dtype)
        X[at] = xe_of(i_w + k).astype(dtype)
        part = np.zeros((W, up, G), dtype=dtype)
        lanes = np.arange(G) * lane_stride
        for w in range(W):
            acc = np.zeros((up, G), dtype=dtype)
            for row in range(w * rpw, (w + 1) * rpw):
                x = X[lanes + roff[row]]
                acc = acc + T[row, :up, None].astype(dtype) * x[None, :]
            part[w] = acc
        tot = part[0]
        for w in range(1, W):
            tot = tot + part[w]                                          # [q][n], wavefront order
        o = np.arange(G * up)
        m = N0 * up + o
        ok = m < n_out
        y[m[ok]] = tot[o[ok] % up, o[ok] // up]
    return y
