"""NumPy model of the shared-inverse LS solve used by the GPU LS chain (math check only).

For Doppler bin f the reference builds r_f = roll(ref * exp(j phi_f(n)), -peek) and solves
Toeplitz(c_f) w = b_f (clutter_removal.py:139-150).  With rho = roll(ref, -peek) and
theta = 2 pi f / Fs:   r_f[n] = rho[n] e^{j theta (n+peek)} g[n],  g[n] = 1 (n < N-peek), gamma = e^{-j theta N} (wrapped)
  =>  c_f[k] = e^{j theta k} ( c_0[k] + (gamma - 1) S_e[k] ),
      S_e[k] = sum_{n >= max(N-peek, k), n-k < N-peek} rho[n] conj(rho[n-k])          (<= peek terms)
so ONE Levinson-Durbin run on c_0 per block (forward predictor a, error E) gives T_0^{-1}
(Trench recursion, dense), and every bin is  w = D T_0^{-1} D^H b  + one refinement step against
the exact c_f  (D = diag(e^{j theta k})).  This file checks each identity against direct evaluation.
"""
import numpy as np
from scipy.linalg import solve_toeplitz, toeplitz


def durbin(c):
    """forward predictor a (a[0]=1) and error E with Toeplitz(c) a = E e_0"""
    n = c.size
    a = np.zeros(n, complex); a[0] = 1.0
    err = c[0].real
    for m in range(1, n):
        k = -np.dot(a[:m], c[m:0:-1]) / err
        prev = a[:m + 1].copy()
        a[:m + 1] = prev + k * np.conj(prev[::-1])
        err *= (1.0 - abs(k) ** 2)
    return a, err


def trench_inverse(a, err):
    """dense inverse of the Hermitian Toeplitz matrix from its forward predictor (Gohberg-Semencul
    written as a recurrence along diagonals)"""
    n = a.size
    x = a / err                                  # first column of the inverse
    inv = np.zeros((n, n), complex)
    inv[:, 0] = x
    inv[0, :] = np.conj(x)
    xr = np.conj(x[::-1])                        # xr[i] = conj(x[n-1-i])
    for i in range(n - 1):
        for j in range(n - 1):
            inv[i + 1, j + 1] = inv[i, j] + (x[i + 1] * np.conj(x[j + 1]) - xr[i] * np.conj(xr[j])) / x[0]
    return inv


if __name__ == "__main__":
    rng = np.random.default_rng(3)
    N, L, peek, fs = 6000, 30, 10, 12000.0
    T = L + peek
    ref = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / np.sqrt(2)).astype(np.complex64)
    ref = (ref + 0.5 * np.roll(ref, 1)).astype(np.complex64)          # mildly coloured
    srv = (np.roll(ref, 2) + 0.3 * np.roll(ref, 9) + 0.05 * rng.standard_normal(N)).astype(np.complex64)
    rho = np.roll(ref, -peek).astype(complex)

    def corr(x, y, k):      # sum_{n>=k} x[n] conj(y[n-k])
        return np.vdot(y[:N - k], x[k:])

    c0 = np.array([corr(rho, rho, k) for k in range(T)])
    Se = np.array([sum(rho[n] * np.conj(rho[n - k]) for n in range(max(N - peek, k), N) if n - k < N - peek)
                   for k in range(T)])
    a, E = durbin(c0)
    Tinv = trench_inverse(a, E)
    print("trench inverse err", np.abs(Tinv - np.linalg.inv(toeplitz(c0, np.conj(c0)))).max() * abs(c0[0]))
    for f in (1.0, -1.0, 2.0, 0.37):
        theta = 2 * np.pi * f / fs
        nn = np.arange(N, dtype=np.complex64)
        rf = np.roll(ref * np.exp(1j * 2 * np.pi * f * nn / fs), -peek).astype(complex)   # reference form (f32 phase)
        cf = np.array([corr(rf, rf, k) for k in range(T)])
        gamma = np.exp(-1j * theta * N)
        cf_model = np.exp(1j * theta * np.arange(T)) * (c0 + (gamma - 1) * Se)
        bf = np.array([corr(srv.astype(complex), rf, k) for k in range(T)])
        w_ref = solve_toeplitz(cf, bf)
        D = np.exp(1j * theta * np.arange(T))
        Tf = toeplitz(cf_model, np.conj(cf_model))
        x = D * (Tinv @ (np.conj(D) * bf))
        e0 = np.abs(x - w_ref).max() / np.abs(w_ref).max()
        x = x + D * (Tinv @ (np.conj(D) * (bf - Tf @ x)))
        e1 = np.abs(x - w_ref).max() / np.abs(w_ref).max()
        print(f"f={f:+.2f}  c_f model err {np.abs(cf - cf_model).max() / abs(cf[0]):.2e}   "
              f"taps err: no refinement {e0:.2e}, one refinement {e1:.2e}")


def check_unrotated_reference_form():
    """Second identity set (cached-spectrum LS chain): everything is expressed with the UNROTATED
    rho = roll(ref, -peek), the rotation moves to the surveillance stream and to the output:
        s~[n]   = s[n] e^{-j theta (n+peek)}
        b_f[k]  = e^{j theta k} ( sum_{n>=k} s~[n] conj(rho[n-k]) + (conj(gamma)-1) E_b[k] ),
        E_b[k]  = sum_{m=N-peek}^{N-1-k} conj(rho[m]) s~[m+k]                     (k < peek, else 0)
        w~[k]   = w[k] e^{-j theta k}
        out[n]  = s[n] - e^{j theta (n+peek)} ( (rho * w~)[n] + [n >= N-peek] (gamma-1) sum_{k<=n-(N-peek)} w~[k] rho[n-k] )
    """
    rng = np.random.default_rng(5)
    N, L, peek, fs = 5000, 24, 10, 9000.0
    T = L + peek
    ref = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / np.sqrt(2)).astype(np.complex64)
    srv = (np.roll(ref, 3) + 0.2 * np.roll(ref, 7) + 0.05 * rng.standard_normal(N)).astype(np.complex64)
    rho = np.roll(ref, -peek).astype(complex)
    for f in (1.0, -0.37):
        theta = 2 * np.pi * f / fs
        nn = np.arange(N)
        rf = np.roll(ref.astype(complex) * np.exp(1j * theta * nn), -peek)       # reference form (exact phase)
        gamma = np.exp(-1j * theta * N)
        st = srv.astype(complex) * np.exp(-1j * theta * (nn + peek))
        b_ref = np.array([np.vdot(rf[:N - k], srv.astype(complex)[k:]) for k in range(T)])
        Bt = np.array([np.vdot(rho[:N - k], st[k:]) for k in range(T)])
        Eb = np.array([sum(np.conj(rho[m]) * st[m + k] for m in range(N - peek, N - k)) if k < peek else 0.0
                       for k in range(T)])
        b_model = np.exp(1j * theta * np.arange(T)) * (Bt + (np.conj(gamma) - 1) * Eb)
        w = rng.standard_normal(T) + 1j * rng.standard_normal(T)
        out_ref = srv.astype(complex) - np.convolve(rf, w)[:N]
        wt = w * np.exp(-1j * theta * np.arange(T))
        y = np.convolve(rho, wt)[:N]
        for n in range(N - peek, N):
            y[n] += (gamma - 1) * sum(wt[k] * rho[n - k] for k in range(0, n - (N - peek) + 1))
        out_model = srv.astype(complex) - np.exp(1j * theta * (nn + peek)) * y
        print(f"f={f:+.2f}  b_f model err {np.abs(b_ref - b_model).max() / np.abs(b_ref).max():.2e}   "
              f"FIR model err {np.abs(out_ref - out_model).max():.2e}")


if __name__ == "__main__":
    check_unrotated_reference_form()
