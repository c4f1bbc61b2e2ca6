"""NumPy model of the block arithmetic of the cached-spectrum LS chain (passiveradar_amd/csrc/ls_fft.hip,
ls_fft_team_cached.hip, ls_fft_team_corr_cached.hip): pieces of B = P - E samples with the slot origin E >= T - 1
(E = T - 1 on the 1024-point kernels, T - 1 rounded up to 16 samples on the 4096-point ones), the block spectrum
X_p = FFT(rho[n0 - E : n0 + B]) shared by the overlap-save FIR and by the "roles swapped" correlations
    sum_n s[n] conj(rho[n - k]) = sum_p IFFT( FFT(s piece in slots [E, E + cnt)) conj(X_p) )[k],   k <= T - 1,
for rho = roll(ref, -peek) taken with zeros before the block and the <= peek wrapped samples added at its end
(clutter_removal.py:139-155).  Checks the identities the kernels rely on against direct evaluation -- including that
extra history slots (E > T - 1) and the assignment of pieces to teams change nothing.  No GPU."""
import numpy as np
import pytest


def _pieces(n, B):
    return [(p * B, min(B, n - p * B)) for p in range((n + B - 1) // B)]


def _block(rho_lin, n0, E, P):
    """slots 0..P-1 <-> rho_lin[n0 - E + idx], zero outside [0, n)"""
    idx = n0 - E + np.arange(P)
    ok = (idx >= 0) & (idx < rho_lin.size)
    x = np.zeros(P, complex)
    x[ok] = rho_lin[idx[ok]]
    return x


@pytest.mark.parametrize("n,T,peek,P,E", [
    (5000, 40, 10, 1024, 39),        # 1024-point kernels: E = T - 1
    (5000, 266, 10, 1024, 265),
    (24234 + 5, 58, 10, 4096, 64),   # 4096-point kernels: E = T - 1 rounded up to 16; last piece shorter than peek
    (30000, 266, 10, 4096, 272),
    (30000, 266, 0, 4096, 272),      # peek = 0: no wrapped samples
    (9000, 769, 10, 4096, 768),
])
def test_block_identities(n, T, peek, P, E):
    rng = np.random.default_rng(n + T)
    ref = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    s = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    w = (rng.standard_normal(T) + 1j * rng.standard_normal(T)) / T
    rho = np.roll(ref, -peek)                       # clutter_removal.py:139
    B = P - E
    assert E >= T - 1 and B > 0
    X = [np.fft.fft(_block(rho, n0, E, P)) for n0, _ in _pieces(n, B)]

    # overlap-save FIR: out = s - (rho * w)[0:n] (linear convolution, :153-155), outputs taken from slots >= E
    H = np.fft.fft(np.concatenate([w, np.zeros(P - T)]))
    out = np.empty(n, complex)
    for (n0, cnt), Xp in zip(_pieces(n, B), X):
        y = np.fft.ifft(Xp * H)
        out[n0:n0 + cnt] = s[n0:n0 + cnt] - y[E:E + cnt]
    want = s - np.convolve(rho, w)[:n]
    assert np.abs(out - want).max() < 1e-10 * np.abs(want).max()

    # correlations with the roles swapped: the piece in slots [E, E + cnt), zeros elsewhere; lags 0 .. T-1
    def corr(sig, order):
        acc = np.zeros(P, complex)
        for p in order:
            n0, cnt = _pieces(n, B)[p]
            u = np.zeros(P, complex)
            u[E:E + cnt] = sig[n0:n0 + cnt]
            acc += np.fft.fft(u) * np.conj(X[p])
        return np.fft.ifft(acc)[:T]
    npc = len(X)
    direct = np.array([np.sum(s[k:] * np.conj(rho[:n - k])) for k in range(T)])          # xcorr, :145-147
    strided = corr(s, [p for t in range(3) for p in range(t, npc, 3)])                    # pieces team, team + 3, ...
    per = -(-npc // 3)
    runs = corr(s, [p for t in range(3) for p in range(t * per, min((t + 1) * per, npc))])  # a contiguous run per team
    scale = np.abs(direct).max()
    assert np.abs(strided - direct).max() < 1e-10 * scale and np.abs(runs - direct).max() < 1e-10 * scale
    auto = corr(rho, range(npc))
    direct_auto = np.array([np.sum(rho[k:] * np.conj(rho[:n - k])) for k in range(T)])   # :142-144
    assert np.abs(auto - direct_auto).max() < 1e-10 * np.abs(direct_auto).max()


def test_piece_size_of_the_4096_point_chain():
    """ltc_piece (ls_team_cached.h): B = 4096 - E with E = T - 1 rounded up to 16 samples -- every piece of a 128-byte
    aligned stream then starts on a 128-byte line, and E still covers the T - 1 samples of history the FIR needs"""
    for T in (2, 17, 18, 250, 266, 273, 769):
        E = (T - 1 + 15) & ~15
        B = 4096 - E
        assert E >= T - 1 and E - (T - 1) < 16 and (B * 8) % 128 == 0 and B > 0
