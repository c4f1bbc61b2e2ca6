"""NumPy model of the wavefront FFT used by caf_fft.hip / ls_fft.hip (index math only).

P = 64*r points live in one wavefront as r complex values per lane.  Forward:
  S1 lane n2 holds x[64*n1 + n2], n1<r        -> radix-r DFT over n1          (registers)
  S2 multiply by W_P^(n2*k1)
  S3 LDS transpose: lane l -> (k1 = l // g, j = l % g), g = 64 // r lanes per 64-point FFT,
     register m holds A[k1][j + g*m], m < r
  S4 radix-r DFT over m -> m'                                                  (registers)
  S5 multiply by W_64^(j*m')
  S6 radix-g DFT over j across the g lanes of a group -> j'
  result: lane 4*k1 + j' (g=4), register m'  <->  frequency k = k1 + r*m' + r*r*j'  (P=1024: k1+16m'+256j')
The inverse runs the same stages backwards with conjugated twiddles and returns natural order.
Pointwise products (correlation) are done in the permuted layout, so no reordering is needed.
"""
import numpy as np


def forward(x, r=16):
    P = 64 * r
    g = 64 // r
    a = x.reshape(r, 64).T.copy()                       # a[n2, n1]
    n1 = np.arange(r)
    A = a @ np.exp(-2j * np.pi * np.outer(n1, n1) / r)  # A[n2, k1]
    A = A * np.exp(-2j * np.pi * np.outer(np.arange(64), n1) / P)
    buf = A.T.copy()                                    # buf[k1, n2]
    lanes = np.arange(64)
    k1 = lanes // g
    j = lanes % g
    c = np.stack([buf[k1, j + g * m] for m in range(r)], axis=1)     # c[lane, m]
    C = c @ np.exp(-2j * np.pi * np.outer(n1, n1) / r)               # C[lane, m']
    C = C * np.exp(-2j * np.pi * np.outer(j, n1) / 64)
    D = np.empty_like(C)
    for l in lanes:
        base = l - j[l]
        jp = j[l]
        D[l] = sum(C[base + jj] * np.exp(-2j * np.pi * jj * jp / g) for jj in range(g))
    return D                                            # D[lane, m']


def freq_index(r=16):
    g = 64 // r
    lanes = np.arange(64)[:, None]
    m = np.arange(r)[None, :]
    return (lanes // g) + r * m + r * r * (lanes % g)


def inverse(D, r=16):
    P = 64 * r
    g = 64 // r
    lanes = np.arange(64)
    j = lanes % g
    k1 = lanes // g
    n1 = np.arange(r)
    C = np.empty_like(D)
    for l in lanes:
        base = l - j[l]
        C[l] = sum(D[base + jp] * np.exp(+2j * np.pi * j[l] * jp / g) for jp in range(g))
    C = C * np.exp(+2j * np.pi * np.outer(j, n1) / 64)
    c = C @ np.exp(+2j * np.pi * np.outer(n1, n1) / r)                # c[lane, m]
    buf = np.empty((r, 64), dtype=complex)
    for m in range(r):
        buf[k1, j + g * m] = c[:, m]
    A = buf.T * np.exp(+2j * np.pi * np.outer(np.arange(64), n1) / P)  # A[n2, k1]
    a = A @ np.exp(+2j * np.pi * np.outer(n1, n1) / r)                # a[n2, n1]
    return a.T.reshape(P) / P


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for r in (8, 16, 32):
        P = 64 * r
        x = rng.standard_normal(P) + 1j * rng.standard_normal(P)
        D = forward(x, r)
        X = np.fft.fft(x)
        idx = freq_index(r)
        print(r, "fwd", np.abs(D - X[idx]).max(), "inv", np.abs(inverse(D, r) - x).max())
        # correlation check: g[l] = sum_i conj(u[i]) v[i+l]
        B, R = P - P // 4, P // 4
        u = np.zeros(P, complex); u[:B] = rng.standard_normal(B) + 1j * rng.standard_normal(B)
        v = rng.standard_normal(P) + 1j * rng.standard_normal(P)
        gcor = inverse(np.conj(forward(u, r)) * forward(v, r), r)
        ref = np.array([np.vdot(u[:B], v[l:l + B]) for l in range(R + 1)])
        print("   corr", np.abs(gcor[:R + 1] - ref).max())


# ---- the quad (radix-4 across four lanes) stage as fft_wave.h issues it -------------------------------------
# Each butterfly r = s*c + c_partner with a per-lane sign s is ONE v_fmac_f32_dpp, d += dpp(d) * (-s), when it is
# written on the sign-scaled value: the partner's sign is the opposite of one's own.  The scaling sg = sA*sB is
# folded into the twiddle table in front of the stage (forward) or into the caller's spectrum (inverse).
def quad_forward_fused(c):
    """c[lane]: values after the W_64^(j m') twiddle for one register, 64 lanes; returns lane j' <- X[bitrev2(j')]"""
    lanes = np.arange(64)
    j = lanes & 3
    sA = np.where(j < 2, 1.0, -1.0)
    sB = np.where(j & 1, -1.0, 1.0)
    u = (sA * sB) * c                          # folded into TW2S
    u = u - sA * u[lanes ^ 2]                  # v_fmac_f32_dpp quad_perm:[2,3,0,1]
    u = np.where(j == 3, -1j * u, u)           # lane 3 carries the -i twiddle
    u = u - sB * u[lanes ^ 1]                  # v_fmac_f32_dpp quad_perm:[1,0,3,2]
    return u


def quad_inverse_fused(y):
    lanes = np.arange(64)
    j = lanes & 3
    sA = np.where(j < 2, 1.0, -1.0)
    sB = np.where(j & 1, -1.0, 1.0)
    p = (sA * sB) * y                          # PRESCALED: folded into the filter spectrum by the caller
    p = p - sB * p[lanes ^ 1]
    p = np.where(j == 3, 1j * p, p)
    p = p - sA * p[lanes ^ 2]
    return p


def _check_quad():
    rng = np.random.default_rng(1)
    c = rng.standard_normal(64) + 1j * rng.standard_normal(64)
    lanes = np.arange(64)
    base, j = lanes & ~3, lanes & 3
    bitrev2 = np.array([0, 2, 1, 3])
    want = np.array([sum(c[base[l] + jj] * np.exp(-2j * np.pi * jj * bitrev2[j[l]] / 4) for jj in range(4)) for l in lanes])
    got = quad_forward_fused(c)
    back = quad_inverse_fused(got) / 4
    print("quad stage, fused-DPP form: forward vs 4-point DFT (bit-reversed lanes)", np.abs(got - want).max(),
          " inverse(forward) vs input", np.abs(back - c).max())


if __name__ == "__main__":
    _check_quad()
