#!/usr/bin/env python3
"""Index model of the 8-point-per-thread 4096-point team transform (passiveradar_amd/csrc/fft_team8.h): 512 threads =
eight wavefronts, four radix-8 passes in registers, one cross-wave exchange and two wave-private ones through ONE
36 KB LDS buffer.  Every phase below is written thread by thread exactly as the kernel addresses it (same LDS element
indices), checked against numpy.fft, and every LDS access pattern is checked for bank conflicts under the gfx950 rules
of MI355X_MICROARCH.md (ds_read_b64: two groups of 32 lanes over 64 banks; ds_write_b64: four groups of 16).

n = 512 n1 + 64 n2 + 8 n3 + n4,  k = k1 + 8 k2 + 64 k3 + 512 k4;  thread t = 64 w + l
  time layout      : thread t, register r  <->  sample 512 r + t
  frequency layout : wave k1, lane 8 k2 + k3, register k4  <->  bin k1 + 8 k2 + 64 k3 + 512 k4
"""
import numpy as np

P, T, PITCH = 4096, 512, 72
REGION = 8 * PITCH                     # float2 per wave region (576): the exchange buffer is 8 regions = 36 864 B
W = lambda N, e: np.exp(-2j * np.pi * (np.asarray(e) % N) / N)


def dft8(x, sign=-1):                  # over the register axis (last), natural order
    k = np.arange(8)
    M = np.exp(sign * 2j * np.pi * np.outer(k, k) / 8)
    return x @ M.T


def conflicts(addr, group):
    """worst number of distinct float2 addresses on one bank within a lane group: ds_read_b64 serves 32 lanes per cycle
    over 64 four-byte banks (float2 index mod 32), ds_write_b64 16 lanes over 32 banks (float2 index mod 16)"""
    worst = 1
    mod = 32 if group == 32 else 16
    for g0 in range(0, 64, group):
        a = np.unique(addr[g0:g0 + group])
        worst = max(worst, np.bincount(a % mod, minlength=mod).max())
    return worst


def forward(x, report):
    t = np.arange(T); w, l = t >> 6, t & 63
    regs = np.stack([x[512 * r + t] for r in range(8)], axis=1)          # time layout
    lds = np.zeros(8 * REGION, complex)
    # S1 over n1, T1 = W_4096^(t k1) = W_4096^(l k1) [table k1][l]  *  W_64^(w k1) [applied after the exchange, wave-uniform]
    regs = dft8(regs)
    regs = regs * W(4096, np.outer(l, np.arange(8)))
    # X1: write E[k1][n2 = w][l]; barrier; read thread (w = k1, l) register n2
    for k1 in range(8):
        a = k1 * REGION + w * 64 + l
        lds[a] = regs[:, k1]
        report("X1 write", max(conflicts(a[64 * ww:64 * ww + 64], 16) for ww in range(8)))
    new = np.empty_like(regs)
    for n2 in range(8):
        a = w * REGION + n2 * 64 + l
        new[:, n2] = lds[a]
        report("X1 read", max(conflicts(a[64 * ww:64 * ww + 64], 32) for ww in range(8)))
    regs = new * W(64, np.outer(w, np.arange(8)))                         # T1b: W_64^(n2 k1), k1 = this wave: SGPR constants
    # S2 over n2, T2 = W_512^(l k2)
    regs = dft8(regs) * W(512, np.outer(l, np.arange(8)))
    # X2 (wave-private): write tile[k2][l] pitch 72; read lane (k2', n4 = l & 7) register n3 <- tile[k2'][8 n3 + n4]
    for k2 in range(8):
        a = w * REGION + k2 * PITCH + l
        lds[a] = regs[:, k2]
        report("X2 write", conflicts(a[:64], 16))
    k2p, n4 = l >> 3, l & 7
    for n3 in range(8):
        a = w * REGION + k2p * PITCH + 8 * n3 + n4
        new[:, n3] = lds[a]
        report("X2 read", conflicts(a[:64], 32))
    # S3 over n3, T3 = W_64^(n4 k3)
    regs = dft8(new) * W(64, np.outer(n4, np.arange(8)))
    # X3 (wave-private): lane (k2, n4) register k3 -> tile[72 k2 + 9 k3 + n4]; read lane (k2', k3' = l & 7) register n4
    for k3 in range(8):
        a = w * REGION + k2p * PITCH + 9 * k3 + n4
        lds[a] = regs[:, k3]
        report("X3 write", conflicts(a[:64], 16))
    k3p = l & 7
    for m in range(8):
        a = w * REGION + k2p * PITCH + 9 * k3p + m
        new[:, m] = lds[a]
        report("X3 read", conflicts(a[:64], 32))
    return dft8(new)                                                      # S4 over n4: register k4


def inverse(X, report):
    """frequency layout -> time layout, unnormalised (x 4096): the forward phases backwards, conjugated"""
    t = np.arange(T); w, l = t >> 6, t & 63
    k2p, low = l >> 3, l & 7
    lds = np.zeros(8 * REGION, complex)
    regs = dft8(X, +1)                                                    # over k4 -> n4; lane (k2, k3)
    for m in range(8):
        a = w * REGION + k2p * PITCH + 9 * low + m
        lds[a] = regs[:, m]
        report("X3' write", conflicts(a[:64], 16))
    new = np.empty_like(regs)
    for k3 in range(8):                                                   # lane (k2, n4) register k3
        a = w * REGION + k2p * PITCH + 9 * k3 + low
        new[:, k3] = lds[a]
        report("X3' read", conflicts(a[:64], 32))
    regs = dft8(new * np.conj(W(64, np.outer(low, np.arange(8)))), +1)    # over k3 -> n3
    for n3 in range(8):
        a = w * REGION + k2p * PITCH + 8 * n3 + low
        lds[a] = regs[:, n3]
        report("X2' write", conflicts(a[:64], 16))
    for k2 in range(8):                                                   # lane l = 8 n3 + n4, register k2
        a = w * REGION + k2 * PITCH + l
        new[:, k2] = lds[a]
        report("X2' read", conflicts(a[:64], 32))
    regs = dft8(new * np.conj(W(512, np.outer(l, np.arange(8)))), +1)     # over k2 -> n2
    regs = regs * np.conj(W(64, np.outer(w, np.arange(8))))               # T1b conj: k1 = this wave, register n2
    for n2 in range(8):
        a = w * REGION + n2 * 64 + l
        lds[a] = regs[:, n2]
    for k1 in range(8):                                                   # barrier; thread (w = n2, l) register k1
        a = k1 * REGION + w * 64 + l
        new[:, k1] = lds[a]
    regs = dft8(new * np.conj(W(4096, np.outer(l, np.arange(8)))), +1)    # T1a conj, over k1 -> n1
    return regs


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    x = rng.standard_normal(P) + 1j * rng.standard_normal(P)
    seen = {}

    def report(name, ways):
        seen[name] = max(seen.get(name, 1), ways)
    X = forward(x, report)
    t = np.arange(T); w, l = t >> 6, t & 63
    want = np.fft.fft(x)
    k = w[:, None] + 8 * (l >> 3)[:, None] + 64 * (l & 7)[:, None] + 512 * np.arange(8)[None, :]
    err_f = np.abs(X - want[k]).max() / np.abs(want).max()
    back = inverse(X, report)
    n = 512 * np.arange(8)[None, :] + t[:, None]
    err_i = np.abs(back / P - x[n]).max()
    print(f"forward vs numpy.fft: {err_f:.2e}; inverse(forward) vs input: {err_i:.2e}")
    print("LDS bank conflicts (ways; 1 = conflict-free):", seen)
    assert err_f < 1e-12 and err_i < 1e-12 and max(seen.values()) == 1
    print("OK")
