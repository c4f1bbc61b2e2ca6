#!/usr/bin/env python3
"""Index-for-index model of the 256-thread, 4096-point "team" FFT of passiveradar_amd/csrc/fft_team.h.

Every register, every LDS address and every twiddle below is what the HIP code uses; the model checks
  * forward and inverse transforms against numpy.fft (and the permuted frequency layout),
  * the bank behaviour of every LDS access phase under the gfx950 ds_read/write_b64 rules (lane groups
    {0-31}, {32-63}; bank of a float2 address a = (2a) mod 64, each access covers 2 banks): all conflict
    free except the cross-wave read (pattern B), which has one 2-way conflict per lane group -- no odd
    pitch avoids it and an even pitch would make every write conflict,
  * that a wave's exchange traffic outside the one barrier stays inside its own quarter of a buffer.

Layout (t = thread 0..255, hi = t >> 4, lo = t & 15, r = register 0..15):
  time      : x[256 r + t]                                   (coalesced global loads)
  frequency : X[hi + 16 lo + 256 r]                          (products are pointwise: never un-permuted)
4096 = 16 x 16 x 16 with n = 256 n1 + 16 n2 + n3, k = k1 + 16 k2 + 256 k3.  Both exchange buffers are
"thread major": element (thread tau, register rho) lives at 17 tau + rho, so every access is a per-thread
base plus a compile-time offset, and a wave only ever WRITES its own quarter (threads 64w .. 64w+63):
  S1 DFT16 over n1 (registers)           thread (n2, n3) = (hi, lo)
  T1 twiddle W_256^(n2 k1)               LDS table [k1][n2], broadcast reads
  X1 exchange across the 4 waves         write (t, k1) | __syncthreads | read (16 m + lo, hi)      [pattern B]
  S2 DFT16 over n2 (registers)           thread (k1, n3) = (hi, lo)
  T2 twiddle W_4096^(n3 (k1 + 16 k2))    16 per-thread constants
  X2 exchange inside each 16-lane row    OTHER buffer: write (t, k2) | read (16 hi + j, lo)        [pattern A]
  S3 DFT16 over n3 (registers)           thread (k1, k2) = (hi, lo), register k3
The inverse runs the stages backwards with conjugated twiddles: X2 write (t, j) / read (16 hi + m, lo), then X1
write (t, m) | __syncthreads | read (16 k1 + lo, hi), both in the target buffer.
"""
import numpy as np

P = 4096
PITCH = 17             # float2 elements per thread row
BUF = 256 * PITCH      # one exchange buffer


def dft16(v, sign):
    """v: (..., 16) natural order in and out, sum_n v[n] W16^(sign n k)"""
    n = np.arange(16)
    W = np.exp(sign * -2j * np.pi * np.outer(n, n) / 16)
    return v @ W


def bank_conflicts(addrs, what, allow=1):
    """addrs: float2 addresses of the 256 threads for ONE register (one wave-instruction per wave)."""
    worst = 1
    for w in range(4):
        for half in range(2):
            lanes = addrs[64 * w + 32 * half:64 * w + 32 * half + 32]
            banks = {}
            for a in lanes:
                for d in (0, 1):
                    b = (2 * int(a) + d) % 64
                    banks.setdefault(b, set()).add(int(a))
            worst = max(worst, max(len(s) for s in banks.values()))
    assert worst <= allow, f"{what}: {worst}-way bank conflict"


def tables():
    k1 = np.arange(16)[:, None]
    n2 = np.arange(16)[None, :]
    tw1 = np.exp(-2j * np.pi * (k1 * n2) / 256.0)                      # [k1][n2]
    t = np.arange(256)
    kk1, n3 = t >> 4, t & 15
    tw2 = np.exp(-2j * np.pi * (n3[:, None] * (kk1[:, None] + 16 * np.arange(16)[None, :])) / 4096.0)   # [t][k2]
    return tw1, tw2


def own_quarter(addr, t):
    assert np.all(addr // (64 * PITCH) == t >> 6)         # a wave writes only its own quarter of a buffer


def exchange(reg, pattern, check_banks):
    """write (t, rho) for every register, then read pattern 'A' (16 hi + j, lo) or 'B' (16 m + lo, hi)"""
    t = np.arange(256)
    hi, lo = t >> 4, t & 15
    buf = np.full(BUF, np.nan, complex)
    for rho in range(16):
        addr = t * PITCH + rho
        own_quarter(addr, t)
        if check_banks:
            bank_conflicts(addr, "write")
        buf[addr] = reg[:, rho]
    out = np.empty_like(reg)
    for m in range(16):
        if pattern == "A":
            addr = hi * 16 * PITCH + lo + PITCH * m        # (16 hi + m) * 17 + lo: a row-private read
            own_quarter(addr, t)
        else:
            addr = lo * PITCH + hi + 16 * PITCH * m        # (16 m + lo) * 17 + hi: reads every wave's quarter
        if check_banks:
            bank_conflicts(addr, "read " + pattern, allow=1 if pattern == "A" else 2)
        out[:, m] = buf[addr]
    return out


def fwd(x, check_banks=True):
    """x: natural time order (4096,) -> registers in the frequency layout [t][r]"""
    tw1, tw2 = tables()
    t = np.arange(256)
    reg = x.reshape(16, 256).T.copy()                     # reg[t][r] = x[256 r + t]
    reg = dft16(reg, +1) * tw1[:, t >> 4].T               # S1 over n1 -> k1, T1: [t][k1] *= tw1[k1][n2 = hi]
    reg = exchange(reg, "B", check_banks)                 # X1 (block barrier between write and read)
    reg = dft16(reg, +1) * tw2                            # S2 over n2 -> k2, T2
    reg = exchange(reg, "A", check_banks)                 # X2 (row private)
    return dft16(reg, +1)                                 # S3 over n3 -> k3


def inv(Y, check_banks=True):
    """Y: registers in the frequency layout [t][r] -> natural time order (4096,), unnormalised (x 4096)"""
    tw1, tw2 = tables()
    t = np.arange(256)
    reg = dft16(Y, -1)                                    # over k3 -> n3
    reg = exchange(reg, "A", check_banks) * np.conj(tw2)  # X2, conj T2 (register = k2)
    reg = dft16(reg, -1)                                  # over k2 -> n2
    reg = exchange(reg, "B", check_banks) * np.conj(tw1[:, t >> 4].T)   # X1, conj T1 (register = k1)
    reg = dft16(reg, -1)                                  # over k1 -> n1
    return reg.T.reshape(-1)                              # x[256 r + t]


def freq_bin(t, r):
    return (t >> 4) + 16 * (t & 15) + 256 * r


def main():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(P) + 1j * rng.standard_normal(P)
    X = np.fft.fft(x)
    F = fwd(x)
    t = np.arange(256)[:, None]
    r = np.arange(16)[None, :]
    ref = X[freq_bin(t, r)]
    e = np.abs(F - ref).max() / np.abs(ref).max()
    print("forward vs numpy.fft in the permuted layout:", e)
    assert e < 1e-12
    y = inv(F)
    e = np.abs(y / P - x).max()
    print("inverse(forward(x)) / 4096 - x:", e)
    assert e < 1e-12
    # correlation through the permuted layout: lags 0..L of sum_n conj(u[n]) v[n + l] (the CAF / LS use)
    u = np.zeros(P, complex)
    u[:3000] = rng.standard_normal(3000) + 1j * rng.standard_normal(3000)
    v = rng.standard_normal(P) + 1j * rng.standard_normal(P)
    c = inv(np.conj(fwd(u, False)) * fwd(v, False), False) / P
    direct = np.array([np.sum(np.conj(u[:3000]) * v[l:l + 3000]) for l in range(0, 1097, 137)])
    e = np.abs(c[0:1097:137] - direct).max() / np.abs(direct).max()
    print("lag products through the permuted layout:", e)
    assert e < 1e-12
    print("LDS phases as documented (one 2-way conflict per lane group on the cross-wave read only); OK")


if __name__ == "__main__":
    main()
