#!/usr/bin/env python3
"""Model for DESIGN.md section 7 item 3 (not wired into the library yet): replace the dense T_0^{-1} of the LS chain
(Trench recurrence, 1.1 MB per chunk, streamed from L2 for every mat-vec) by its Gohberg-Semencul form,

    T^{-1} = (1/x_0) [ L(x) L(x)^H - L(z) L(z)^H ],   x = T^{-1} e_0 = a / err  (a: forward predictor from
    Levinson-Durbin, a[0] = 1, err: final prediction error),  z = (0, conj(x[T-1]), ..., conj(x[1])),

L(v) = lower-triangular Toeplitz matrix with first column v: a mat-vec is four triangular Toeplitz products on
T-vectors that live in LDS.  Checks: (1) the identity against a dense inverse, (2) that it is what the Trench
recurrence of ls_prepare_kernel builds, (3) that the refinement loop of ls_solve_kernel converges identically with
either preconditioner on a Doppler-shifted system, (4) operation counts."""
import numpy as np


def durbin(c):
    """forward predictor a (a[0]=1) and final error of the Hermitian Toeplitz matrix with first column c"""
    T = c.shape[0]
    a = np.zeros(T, complex); a[0] = 1.0
    err = c[0].real
    for m in range(1, T):
        k = -np.dot(a[:m], c[m:0:-1]) / err
        a[:m + 1] = a[:m + 1] + k * np.conj(a[m::-1])
        err *= 1.0 - abs(k) ** 2
    return a, err


def lower_toeplitz(v):
    T = v.shape[0]
    return np.array([[v[i - j] if i >= j else 0.0 for j in range(T)] for i in range(T)])


def gs_apply(a, err, v):
    """T^{-1} v by four triangular Toeplitz products (what the kernel would do out of LDS)"""
    T = a.shape[0]
    z = np.zeros(T, complex); z[1:] = np.conj(a[:0:-1])
    p = np.array([np.dot(np.conj(a[:T - k]), v[k:]) for k in range(T)])       # L(a)^H v
    q = np.array([np.dot(np.conj(z[:T - k]), v[k:]) for k in range(T)])       # L(z)^H v
    y = np.array([np.dot(a[i::-1], p[:i + 1]) - np.dot(z[i::-1], q[:i + 1]) for i in range(T)])
    return y / err


def main():
    rng = np.random.default_rng(3)
    n, T, peek = 6000, 48, 10
    rho = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)
    rho = np.convolve(rho, [1.0, 0.5, 0.2])[:n]                                # mildly coloured
    c0 = np.array([np.vdot(rho[:n - k], rho[k:]) for k in range(T)])          # c0[k] = sum rho[n] conj(rho[n-k])
    Tm = np.array([[c0[i - j] if i >= j else np.conj(c0[j - i]) for j in range(T)] for i in range(T)])
    a, err = durbin(c0)
    dense = np.linalg.inv(Tm)
    x = a / err
    z = np.zeros(T, complex); z[1:] = np.conj(x[:0:-1])
    gs = (lower_toeplitz(x) @ lower_toeplitz(x).conj().T - lower_toeplitz(z) @ lower_toeplitz(z).conj().T) / x[0]
    print("(1) Gohberg-Semencul vs dense inverse      :", np.abs(gs - dense).max() / np.abs(dense).max())
    # (2) Trench recurrence as coded in ls_prepare_kernel
    ie = 1.0 / err
    tr = np.zeros((T, T), complex)
    for d in range(T):
        v = a[d] * ie
        tr[d, 0] = v; tr[0, d] = np.conj(v)
        for j in range(0, T - 1 - d):
            i = j + d
            v = v + (a[i + 1] * np.conj(a[j + 1]) - np.conj(a[T - 1 - i]) * a[T - 1 - j]) * ie
            tr[i + 1, j + 1] = v; tr[j + 1, i + 1] = np.conj(v)
    print("(2) Trench recurrence vs Gohberg-Semencul  :", np.abs(tr - gs).max() / np.abs(gs).max())
    v = rng.standard_normal(T) + 1j * rng.standard_normal(T)
    print("    four triangular products vs dense      :", np.abs(gs_apply(a, err, v) - dense @ v).max() / np.abs(dense @ v).max())
    # (3) refinement on a perturbed, rotated system: c_f = D (c0 + (gamma-1) S_e)
    theta = 2 * np.pi * 2.0 / 1e4
    gamma = np.exp(-1j * theta * n)
    Se = 0.02 * n / T * (rng.standard_normal(T) + 1j * rng.standard_normal(T)); Se[0] = Se[0].real
    D = np.exp(1j * theta * np.arange(T))
    cf = D * (c0 + (gamma - 1) * Se)
    Tf = np.array([[cf[i - j] if i >= j else np.conj(cf[j - i]) for j in range(T)] for i in range(T)])
    b = rng.standard_normal(T) + 1j * rng.standard_normal(T)
    exact = np.linalg.solve(Tf, b)
    for name, apply in (("dense ", lambda r: dense @ r), ("G-S   ", lambda r: gs_apply(a, err, r))):
        w = D * apply(np.conj(D) * b)
        hist = [np.abs(w - exact).max() / np.abs(exact).max()]
        for _ in range(4):
            r = b - Tf @ w
            w = w + D * apply(np.conj(D) * r)
            hist.append(np.abs(w - exact).max() / np.abs(exact).max())
        print(f"(3) refinement with {name} preconditioner    :", " ".join(f"{h:.1e}" for h in hist))
    print(f"(4) per mat-vec at T=266: dense {266 * 266} complex MACs and {266 * 266 * 16 / 1e6:.2f} MB streamed; "
          f"G-S {2 * 266 * 267} complex MACs on {4 * 266 * 16 / 1e3:.1f} KB held in LDS")


if __name__ == "__main__":
    main()


# ---- thread-level model of ls_solve_gs_kernel's triangular Toeplitz products (csrc/ls.hip) ------------------------
def _tri_model(T, terms, threads=1024):
    """terms: list of (up, c, rev, conj, sign, dmin, vec).  Returns out[o] = sum over terms of
         up  : sum_{d=dmin}^{T-1-o} coef(d) vec[o+d]        down: sum_{d=dmin}^{o} coef(d) vec[o-d]
       computed the way the kernel does: groups of 4 consecutive outputs per thread, the group's own lag range cut into
       `parts` equal slices, sliding 4-element window over a zero-padded, 4-way de-interleaved copy of vec, partial
       sums per slice."""
    G = (T + 3) // 4
    parts = max(1, min(threads // G, 16))
    Q = (T + 8 + 3) // 4

    def plane(vec):
        pv = np.zeros(4 * Q, complex)
        for e in range(T):
            ep = e + 4
            pv[(ep & 3) * Q + (ep >> 2)] = vec[e]
        return pv

    def rd(pv, e):
        ep = e + 4
        assert 0 <= ep < T + 8, e
        return pv[(ep & 3) * Q + (ep >> 2)]

    pacc = np.zeros((parts, 4 * G), complex)
    for c in range(parts):
        for g in range(G):
            o0 = 4 * g
            acc = np.zeros(4, complex)
            for up, cv, rev, cj, sign, dmin, vec in terms:
                pv = plane(vec)
                rmax = (T - 1 - o0) if up else min(o0 + 3, T - 1)
                ln = rmax - dmin + 1
                if ln <= 0:
                    continue
                per = -(-ln // parts)
                dlo = dmin + c * per
                dhi = min(dlo + per - 1, rmax)
                if dlo > dhi:
                    continue
                w = [rd(pv, o0 + j + dlo) if up else rd(pv, o0 + j - dlo) for j in range(4)]
                for d in range(dlo, dhi + 1):
                    co = cv[T - d] if rev else cv[d]
                    co = (np.conj(co) if cj else co) * sign
                    for j in range(4):
                        acc[j] += co * w[j]
                    if up:
                        w = w[1:] + [rd(pv, o0 + 3 + d + 1)]
                    else:
                        w = [rd(pv, o0 - (d + 1))] + w[:3]
            pacc[c, o0:o0 + 4] = acc
    return pacc.sum(axis=0)[:T]


def check_kernel_mapping():
    rng = np.random.default_rng(5)
    for T in (50, 266, 267):
        a = rng.standard_normal(T + 1) + 1j * rng.standard_normal(T + 1)   # a[T] is never read (rev needs d >= 1)
        a = a[:T]
        apad = np.concatenate((a, [0]))
        v = rng.standard_normal(T) + 1j * rng.standard_normal(T)
        p = _tri_model(T, [(True, apad, False, True, 1.0, 0, v)])
        q = _tri_model(T, [(True, apad, True, False, 1.0, 1, v)])
        y = _tri_model(T, [(False, apad, False, False, 1.0, 0, p), (False, apad, True, True, -1.0, 1, q)])
        ref = gs_apply(a, 1.0, v)
        e1 = np.abs(y - ref).max() / np.abs(ref).max()
        cf = rng.standard_normal(T) + 1j * rng.standard_normal(T); cf[0] = cf[0].real
        Tf = np.array([[cf[i - j] if i >= j else np.conj(cf[j - i]) for j in range(T)] for i in range(T)])
        cfp = np.concatenate((cf, [0]))
        r = _tri_model(T, [(False, cfp, False, False, 1.0, 0, v), (True, cfp, False, True, 1.0, 1, v)])
        e2 = np.abs(r - Tf @ v).max() / np.abs(Tf @ v).max()
        print(f"(5) kernel mapping T={T}: G-S mat-vec {e1:.1e}, Hermitian Toeplitz product {e2:.1e}")
        assert e1 < 1e-12 and e2 < 1e-12


if __name__ == "__main__":
    check_kernel_mapping()
