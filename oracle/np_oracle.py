"""CPU oracle for the range-Doppler hot path -- TEST INFRASTRUCTURE ONLY.

This module is a closed-form NumPy restatement of what the reference computes on
the path named by BASELINE.json's north_star.  It is the *checker*: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``passiveradar_amd/`` imports it, and the product
path raises when the HIP library is missing instead of falling back to this.

Parity pin: the reference ships no tests or golden vectors, so
the pin is made by us: ``oracle/gen_golden.py`` imports the reference's own
modules in the build container and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against them.
Third-party arithmetic on the path lives in SciPy 1.15.3 / NumPy 2.2.6 (not
pinned by the reference, ``environment.yaml:5-16``); the SciPy calls the
reference makes are restated here in closed form:

* ``scipy.signal.decimate(x, q, ftype=dlti(h,1))`` (zero_phase=True) ==
  ``resample_poly(x, 1, q, window=h)`` == ``y[j] = sum_m h[m] x[j*q + (len(h)-1)//2 - m]``
  with zeros outside ``[0, N)`` and ``ceil(N/q)`` outputs, accumulated in
  complex128.
* ``scipy.signal.correlate(s1, pad(s2,(nlag,nlead)), 'valid')`` ==
  ``z[i] = sum_n s1[n] conj(s2[n-(i-nlead)])``.
* ``scipy.linalg.solve_toeplitz(c, b)`` == Levinson recursion on the Hermitian
  Toeplitz matrix with first column ``c`` (complex128).

Each function cites the reference lines it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "decimation_taps", "caf_segment_sums", "fast_xambg", "fast_xambg_libcalls",
    "direct_xambg", "xcorr", "frequency_shift", "levinson_hermitian",
    "LS_Filter_Toeplitz", "LS_Filter_Multiple", "LS_Filter", "NLMS_filter",
    "decimate_iir", "find_channel_offset", "CFAR_2D", "front_end", "resample", "deinterleave_IQ",
    "overlap_frames", "process_stream",
]


# --------------------------------------------------------------------------
# CAF  (range_doppler_processing.py:12-90)
# --------------------------------------------------------------------------

def decimation_taps(q: int, shortFilt: bool = True) -> np.ndarray:
    """Decimation FIR the reference builds at range_doppler_processing.py:69-78.

    shortFilt -> ones(q+1) (:72); otherwise firwin(10q+1, 1/q, 'flattop') (:76).
    """
    if shortFilt:
        return np.ones(q + 1, dtype=np.float64)
    from scipy.signal import firwin
    return firwin(10 * q + 1, 1.0 / q, window="flattop")


def _resolve_window(window, n):
    # range_doppler_processing.py:57-58 -- str/tuple windows go through get_window
    if isinstance(window, (tuple, str)):
        from scipy.signal import get_window
        return get_window(window, n)
    return window


def caf_segment_sums(ref, srv, rangeBins, freqBins, window=None, taps=None):
    """Lagged conjugate product + decimating FIR, before the Doppler FFT.

    Follows range_doppler_processing.py:61-86.  Returns y[F, R+1] complex128 with
    y[j, k] = sum_m h[m] * p_k[j*q + half - m],  p_k[n] = ref[n] conj(srv[(n+R-k) mod N]) w[n]
    (p_k is rounded to complex64 as the reference's array arithmetic does).
    """
    ref = np.asarray(ref)
    srv = np.asarray(srv)
    N = ref.shape[0]
    q = int(N / freqBins)                                  # :61
    h = decimation_taps(q) if taps is None else np.asarray(taps, dtype=np.float64)
    half = (h.size - 1) // 2
    boxcar = bool(np.all(h == 1.0))
    y = np.zeros((freqBins, rangeBins + 1), dtype=np.complex128)
    sconj = np.conj(srv)                                   # :67
    centres = np.arange(freqBins, dtype=np.int64) * q
    lo = centres + half - (h.size - 1)                     # first sample of each window
    hi = centres + half                                    # last sample (inclusive)
    for k in range(rangeBins + 1):
        ell = rangeBins - k                                # delay in samples
        # np.roll(conj(srv), -ell)[n] == conj(srv[(n+ell) mod N])            (:82)
        prod = np.concatenate((sconj[ell:], sconj[:ell])) * ref
        if window is not None:
            prod = prod * window                           # :83-84
        if prod.dtype != np.complex128:
            # the reference keeps complex64 when srv is complex64 and the window is
            # applied in place (:84); upfirdn then accumulates in complex128.
            prod = prod.astype(np.result_type(ref.dtype, srv.dtype)).astype(np.complex128)
        if boxcar:
            cs = np.concatenate(([0.0], np.cumsum(prod)))
            a = np.clip(lo, 0, N)
            b = np.clip(hi + 1, 0, N)
            y[:, k] = cs[b] - cs[a]
        else:
            # full = np.convolve(prod, h), full[t] = sum_m h[m] prod[t-m], taken only at the F outputs t_j = j q + half
            # (resample_poly keeps those): rows of a strided view of the zero-padded products against the reversed taps
            nt = h.size
            P = np.concatenate((np.zeros(nt - 1, dtype=np.complex128), prod, np.zeros(nt, dtype=np.complex128)))
            idx = centres + half
            ok = idx < N + nt - 1
            W = np.lib.stride_tricks.as_strided(P[half:], shape=(freqBins, nt),
                                                strides=(q * P.itemsize, P.itemsize), writeable=False)
            yk = W @ h[::-1]
            y[ok, k] = yk[ok]
    return y


def fast_xambg(refChannel, srvChannel, rangeBins, freqBins, inputLen=None,
               window=None, shortFilt=True):
    """Closed-form restatement of range_doppler_processing.py:12-90.

    Output (freqBins, rangeBins+1, 1) complex64: column k <-> delay R-k, rows are
    fftshift(FFT over the decimated slow-time axis) computed in single precision
    (scipy.fftpack.fft on complex64, :6,:89).
    """
    refChannel = np.asarray(refChannel)
    srvChannel = np.asarray(srvChannel)
    if refChannel.shape != srvChannel.shape:               # :46-49
        raise ValueError("Input vectors must have the same length")
    if inputLen is not None and refChannel.shape[0] != inputLen:   # :52-55
        pad = inputLen - refChannel.shape[0]
        refChannel = np.pad(refChannel, (0, pad))
        srvChannel = np.pad(srvChannel, (0, pad))
    window = _resolve_window(window, inputLen)
    q = int(refChannel.shape[0] / freqBins)
    y = caf_segment_sums(refChannel, srvChannel, rangeBins, freqBins, window,
                         decimation_taps(q, shortFilt))
    y64 = y.astype(np.complex64)                           # store into complex64 xambg (:64,:86)
    from scipy.fft import fft as _fft                      # single precision for complex64 input
    X = np.fft.fftshift(_fft(y64, axis=0), axes=0)         # :89
    return X.reshape(freqBins, rangeBins + 1, 1).astype(np.complex64)


def fast_xambg_libcalls(ref, srv, rangeBins, freqBins, window=None):
    """Same result as fast_xambg, but issuing the library operations the reference
    issues per lag (roll, multiply, window, polyphase boxcar decimation, FFT) with
    the SciPy-1.15 ``dlti._as_zpk -> np.roots`` artefact left out (SURVEY.md section 0:
    the artefact does not change the output).  This is the honest CPU cost of the
    reference path and is what bench.py times as ``cpu_baseline``.
    """
    from scipy.signal import resample_poly
    from scipy.fft import fft as _fft
    n = ref.shape[0]
    q = int(n / freqBins)
    box = np.ones(q + 1)
    out = np.empty((freqBins, rangeBins + 1), dtype=np.complex64)
    sc = np.conjugate(srv)
    for col in range(rangeBins + 1):
        shifted = np.roll(sc, col - rangeBins)
        shifted *= ref
        if window is not None:
            shifted *= window
        out[:, col] = resample_poly(shifted, 1, q, window=box)[:freqBins]
    return np.fft.fftshift(_fft(out, axis=0), axes=0).reshape(freqBins, rangeBins + 1, 1)


def direct_xambg(refChannel, srvChannel, rangeBins, freqBins, sampleRate):
    """Time-domain CAF, range_doppler_processing.py:93-124 (independent peak check only).

    Row i is Doppler (i - F/2)/CPI (not mirrored), column k is delay R-k.
    """
    ref = np.asarray(refChannel)
    srv = np.asarray(srvChannel)
    if ref.shape != srv.shape:
        raise ValueError("Input vectors must have the same length")
    cpi = ref.shape[0] / sampleRate                        # :110
    out = np.zeros((freqBins, rangeBins + 1, 1), dtype=np.complex64)
    for i in range(freqBins):
        df = (i - 0.5 * freqBins) / cpi                    # :118
        out[i, :, 0] = xcorr(frequency_shift(ref, df, sampleRate), srv, rangeBins, 0)  # :120-122
    return out


# --------------------------------------------------------------------------
# helpers from signal_utils.py that sit on the path
# --------------------------------------------------------------------------

def xcorr(s1, s2, nlead, nlag):
    """signal_utils.py:29-32: correlate(s1, pad(s2, (nlag, nlead)), mode='valid').  For equal lengths (every call site
    of the reference) z[i] = sum_n s1[n] conj(s2[n - (i - nlead)]), i = 0..nlag+nlead, terms outside either array
    dropped.  For unequal lengths SciPy's 'valid' mode slides the shorter of (s1, padded s2) over the longer: with
    y = pad(s2), m = len(y), K = |m - len(s1)| it returns K + 1 values,
        m >= len(s1):  z[i] = sum_l s1[l] conj(y[l + K - i])      (correlate swaps its inputs, conjugates, reverses)
        m <  len(s1):  z[k] = sum_l s1[l + k] conj(y[l]).
    Accumulated in complex128, returned in the inputs' result type (complex64 for complex64 inputs, as SciPy's direct
    method does)."""
    s1 = np.asarray(s1)
    s2 = np.asarray(s2)
    n1, n2 = s1.shape[0], s2.shape[0]
    a = s1.astype(np.complex128)
    b = s2.astype(np.complex128)
    m = n2 + nlag + nlead
    K = abs(m - n1)
    out = np.zeros(K + 1, dtype=np.complex128)
    for i in range(K + 1):
        # pair s1[n] with s2[n - d]
        d = (i + nlag - K) if m >= n1 else (nlag + i)
        n_lo = max(0, d)
        n_hi = min(n1, n2 + d)
        if n_hi > n_lo:
            out[i] = np.vdot(b[n_lo - d:n_hi - d], a[n_lo:n_hi])
    return out.astype(np.result_type(s1.dtype, s2.dtype, np.complex64))


def frequency_shift(x, fc, Fs, phase_offset=0):
    """signal_utils.py:24-27.  The sample index is held as complex64, so the phase ramp is evaluated in float32
    (NumPy-2 weak-scalar promotion); this restatement spells the float32 steps out:
        ph[n] = fl32( fl32( fl32(2*pi*fc) * n ) * fl32(1/fl32(Fs)) )
    (complex64 / real goes through NumPy's Smith division = multiply by the float32 reciprocal).  What is added to it
    decides the rest: a scalar phase_offset (or a float32 / float16 array) keeps everything in float32 -- float32 sum,
    complex64 exponential and product; a float64 or integer ARRAY (main.py:137,146 pass one phase per dask block, the
    expression takes one per sample just as well) promotes the sum to double: float32 ramp + double phase, complex128
    exponential and product."""
    x = np.asarray(x)
    n = np.arange(x.shape[0], dtype=np.float32)
    a = np.float32(2 * np.pi * fc)
    rcp = np.float32(1.0) / np.float32(Fs)
    ph = (a * n) * rcp
    if np.ndim(phase_offset) != 0:
        p = np.asarray(phase_offset)
        if p.dtype in (np.float32, np.float16):
            ph = ph + p.astype(np.float32)
            return x * (np.cos(ph) + 1j * np.sin(ph)).astype(np.complex64)
        ph = ph.astype(np.float64) + p.astype(np.float64)
        return x * (np.cos(ph) + 1j * np.sin(ph))
    if phase_offset != 0:
        ph = ph + np.float32(phase_offset)
    rot = (np.cos(ph) + 1j * np.sin(ph)).astype(np.complex64)
    return x * rot


# --------------------------------------------------------------------------
# Clutter filters (clutter_removal.py)
# --------------------------------------------------------------------------

def levinson_hermitian(c, b):
    """Solve T w = b, T[i,j] = c[i-j] (i>=j), conj(c[j-i]) (i<j); complex128.

    Restates scipy.linalg.solve_toeplitz(c, b) (clutter_removal.py:150: only the
    first column is passed, so the first row is conj(c)).  Levinson-Durbin with
    the backward predictor obtained from the forward one by conjugate reversal.
    """
    c = np.asarray(c, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    n = c.size
    w = np.zeros(n, dtype=np.complex128)
    # a: forward predictor polynomial, a[0]=1;  T_m a = [E_m, 0..0]^T
    a = np.zeros(n, dtype=np.complex128)
    a[0] = 1.0
    err = c[0].real
    w[0] = b[0] / c[0]
    for m in range(1, n):
        # reflection coefficient for order m
        acc = np.dot(a[:m], c[m:0:-1])              # sum_i a[i] c[m-i]
        k = -acc / err
        prev = a[:m + 1].copy()
        a[:m + 1] = prev + k * np.conj(prev[::-1])
        err = err * (1.0 - (k * np.conj(k)).real)
        # update the solution: residual of row m with current w
        res = b[m] - np.dot(c[m:0:-1], w[:m])       # sum_j c[m-j] w[j]
        g = res / err
        w[:m + 1] += g * np.conj(a[m::-1])
    return w


def LS_Filter_Toeplitz(refChannel, srvChannel, filterLen, peek=10, return_filter=False):
    """clutter_removal.py:109-160.  T = filterLen+peek taps; tap k <-> delay k-peek.

    r = roll(ref, -peek) (:139, circular); c[k] = sum_{n>=k} r[n] conj(r[n-k]),
    b[k] = sum_{n>=k} s[n] conj(r[n-k]) (:142-147); Hermitian-Toeplitz solve (:150);
    out[n] = s[n] - sum_{k<=min(n,T-1)} w[k] r[n-k] (:153-155).  complex128 out.
    """
    ref = np.asarray(refChannel)
    srv = np.asarray(srvChannel)
    if ref.shape != srv.shape:                              # :133-135
        raise ValueError("Input vectors must have the same length")
    T = filterLen + peek
    r = np.concatenate((ref[peek:], ref[:peek])) if peek else ref
    c = xcorr(r, r, 0, T - 1)
    b = xcorr(srv, r, 0, T - 1)
    w = levinson_hermitian(c, b)
    clutter = np.convolve(r.astype(np.complex128), w)[:srv.shape[0]]
    out = srv.astype(np.complex128) - clutter
    return (out, w) if return_filter else out


def LS_Filter_Multiple(refChannel, srvChannel, filterLen, sampleRate, dopplerBins=(0,)):
    """clutter_removal.py:162-187: chain LS_Filter_Toeplitz over Doppler bins, each on the
    previous bin's output, reference frequency-shifted for non-zero bins (:184)."""
    out = srvChannel
    for fd in dopplerBins:
        r = refChannel if fd == 0 else frequency_shift(refChannel, fd, sampleRate)
        out = LS_Filter_Toeplitz(r, out, filterLen)
    return out


def LS_Filter_Multiple_libcalls(ref, srv, filterLen, sampleRate, dopplerBins=(0,), peek=10):
    """Same result as LS_Filter_Multiple, but issuing the library calls the reference issues
    (scipy correlate 'valid' on zero-padded reference, scipy solve_toeplitz, np.convolve,
    complex64 phase ramp) -- the honest CPU cost of clutter_removal.py:162-187 for bench.py."""
    from scipy.linalg import solve_toeplitz
    from scipy.signal import correlate
    T = filterLen + peek
    cur = srv
    idx = np.arange(ref.shape[0], dtype=np.complex64)
    for fd in dopplerBins:
        shifted = ref if fd == 0 else ref * np.exp(1j * 2 * np.pi * fd * idx / sampleRate)
        r = np.roll(shifted, -peek)
        rp = np.pad(r, (T - 1, 0))
        col = correlate(r, rp, mode="valid")
        rhs = correlate(cur, rp, mode="valid")
        taps = solve_toeplitz(col, rhs)
        cur = cur - np.convolve(r, taps)[:cur.shape[0]]
    return cur


def LS_Filter(refChannel, srvChannel, filterLen, reg=1.0, peek=10, return_filter=False):
    """clutter_removal.py:6-56 without materialising the N x T data matrix.

    A[:,k] = roll(ref, k-peek) (:33-36) so (A^H A)[i,j] = g[i-j] with the *circular*
    autocorrelation g[d] = sum_n ref[n] conj(ref[(n-d) mod N]) -- exactly Hermitian
    Toeplitz -- and (A^H s)[k] = sum_n s[n] conj(ref[(n-k+peek) mod N]).  The
    regularised system (:45) is solved here in complex128 (the reference uses
    LAPACK in complex64); out = s - A w (:51) is a circular FIR.  complex64 out.
    """
    ref = np.asarray(refChannel)
    srv = np.asarray(srvChannel)
    if ref.shape != srv.shape:                              # :28-29
        raise ValueError("Input vectors must have the same length")
    N = ref.shape[0]
    T = filterLen + peek
    r128 = ref.astype(np.complex128)
    s128 = srv.astype(np.complex128)
    g = np.empty(T, dtype=np.complex128)
    rhs = np.empty(T, dtype=np.complex128)
    for d in range(T):
        g[d] = np.vdot(np.roll(r128, d), r128)             # sum_n r[n] conj(r[n-d])
        rhs[d] = np.vdot(np.roll(r128, d - peek), s128)    # sum_n s[n] conj(r[n-(d-peek)])
    g[0] += reg
    w = levinson_hermitian(g, rhs)
    clutter = np.zeros(N, dtype=np.complex128)
    for k in range(T):
        clutter += w[k] * np.roll(r128, k - peek)
    out = (s128 - clutter).astype(np.complex64)
    return (out, w.astype(np.complex64)) if return_filter else out


def NLMS_filter(refChannel, srvChannel, filterLen, mu, peek=10, initialTaps=None,
                returnFilter=False):
    """clutter_removal.py:189-249 (complex64 state, sample-recursive):
        u_k[i] = ref[L+k+peek-i], i=0..T-1;  e = srv[k+L] - w^H u;  w += mu*u*conj(e)/(u^H u)
        out[L+k] = e,  k = 0..N-T-1;  out is zero elsewhere.
    Pure-Python loop: small cases only (oracle/c/nlms.c is the fast twin)."""
    ref = np.asarray(refChannel)
    srv = np.asarray(srvChannel)
    if initialTaps is None:                                 # :218-225
        w = np.zeros(filterLen + peek, dtype=np.complex64)
    else:
        w = np.asarray(initialTaps).copy()
        filterLen = w.shape[0] - peek
    T = filterLen + peek
    N = srv.shape[0]
    out = np.zeros(N, dtype=np.complex64)
    mu32 = np.float32(mu)
    for k in range(N - T):                                  # :234
        u = ref[k + 1:k + T + 1][::-1]                      # :228,:237 sliding window, newest first
        e = srv[k + filterLen] - np.vdot(w, u)              # :212
        w = (w + mu32 * u * np.conj(e) / np.vdot(u, u)).astype(np.complex64)   # :213
        out[filterLen + k] = e                              # :244
    return (out, w) if returnFilter else out


# --------------------------------------------------------------------------
# Block pipeline semantics of main.py:169-194 (the harness row of SURVEY 8c)
# --------------------------------------------------------------------------

def overlap_frames(stream, chunk, depth):
    """dask.array.overlap.overlap(x, depth, boundary=0) on 1-D chunks (main.py:178-181):
    frame i = stream[i*chunk - depth : (i+1)*chunk + depth], zeros beyond the ends."""
    stream = np.asarray(stream)
    nchunks = stream.shape[0] // chunk
    padded = np.concatenate((np.zeros(depth, stream.dtype), stream[:nchunks * chunk],
                             np.zeros(depth, stream.dtype)))
    return [padded[i * chunk:(i + 1) * chunk + 2 * depth] for i in range(nchunks)]


def process_stream(ref, srv, cpi_samples, num_range_cells, num_doppler_cells, IF_sample_rate,
                   dopplerBins=(0, 1, -1, 2, -2), window=("kaiser", 5.0), return_cleaned=False):
    """main.py:169-194 on an in-memory IF stream: per-chunk LS_Filter_Multiple (chunk =
    cpi/2, :169-176), zero-boundary overlap of depth cpi/4 (:178-181), Kaiser window (:183),
    fast_xambg per frame (:186-194), frames stacked on axis 2."""
    from scipy.signal import get_window
    C = cpi_samples // 2
    depth = cpi_samples // 4
    nchunks = ref.shape[0] // C
    cleaned = np.concatenate([
        LS_Filter_Multiple(ref[i * C:(i + 1) * C], srv[i * C:(i + 1) * C], num_range_cells,
                           IF_sample_rate, list(dopplerBins)) for i in range(nchunks)])
    w = get_window(window, cpi_samples) if isinstance(window, (tuple, str)) else window
    rf = overlap_frames(ref, C, depth)
    sf = overlap_frames(cleaned, C, depth)
    frames = [fast_xambg(a, b, num_range_cells, num_doppler_cells, cpi_samples, w)
              for a, b in zip(rf, sf)]
    out = np.concatenate(frames, axis=2)
    return (out, cleaned) if return_cleaned else out


# --------------------------------------------------------------------------
# Front end (SURVEY 8f "next" #1): signal_utils.py:15-22 and main.py:105-166
# --------------------------------------------------------------------------

def deinterleave_IQ(interleavedIQ):
    """signal_utils.py:19-22: [I0, Q0, I1, Q1, ...] scalars -> complex64 (a trailing odd scalar is dropped)."""
    x = np.asarray(interleavedIQ)
    n = x.shape[0] // 2
    return (x[0:2 * n:2].astype(np.float64) + 1j * x[1:2 * n:2].astype(np.float64)).astype(np.complex64)


def resample_design(up, dn):
    """The FIR resample_poly designs for signal_utils.resample (signal_utils.py:15-17: window
    ('kaiser', 5.0), 20*max(up,dn)+1 taps, cutoff 1/max(up,dn), gain up) and its alignment:
    returns (h zero-padded in front, n_pre_remove)."""
    from math import gcd
    from scipy.signal import firwin
    g = gcd(up, dn)
    up, dn = up // g, dn // g
    max_rate = max(up, dn)
    half_len = 10 * max_rate
    h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)) * up
    n_pre_pad = dn - half_len % dn
    n_pre_remove = (half_len + n_pre_pad) // dn
    return np.concatenate((np.zeros(n_pre_pad), h)), n_pre_remove, up, dn


def resample(x, up, dn):
    """signal_utils.py:15-17: scipy.signal.resample_poly(x, up, dn, padtype='line') restated as a
    polyphase sum:  y[m] = sum_j h[(t mod up) + up j] xe[t div up - j],  t = (m + n_pre_remove) dn,
    xe = x extended linearly through its first and last sample (upfirdn mode 'line')."""
    x = np.asarray(x)
    from math import gcd
    if int(up) // gcd(int(up), int(dn)) == 1 and int(dn) // gcd(int(up), int(dn)) == 1:
        return x.copy()                                   # resample_poly: "if up == down == 1: return x.copy()"
    hp, n_pre_remove, up, dn = resample_design(up, dn)
    n_in = x.shape[0]
    n_out = n_in * up
    n_out = n_out // dn + bool(n_out % dn)
    xd = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)
    slope = (xd[-1] - xd[0]) / (n_in - 1)
    J = -(-hp.size // up)
    pad = J + 2
    left = xd[0] + slope * np.arange(-pad, 0)
    right = xd[-1] + slope * np.arange(1, pad + dn // up + 2)
    xe = np.concatenate((left, xd, right))
    t = (np.arange(n_out, dtype=np.int64) + n_pre_remove) * dn
    phase = t % up
    i0 = t // up + pad                      # index into xe
    y = np.zeros(n_out, dtype=xd.dtype)
    for p in range(up):
        taps = hp[p::up]
        sel = np.nonzero(phase == p)[0]
        if sel.size == 0:
            continue
        # y[m] = sum_j taps[j] xe[i0 - j]  = (xe conv taps)[i0]
        full = np.convolve(xe, taps)
        y[sel] = full[i0[sel]]
    return y.astype(x.dtype if np.issubdtype(x.dtype, np.inexact) else y.dtype)


def block_phase_offsets(nblocks, input_chunk_length, input_sample_rate, offset_freq):
    """main.py:125-130: starting phase of every block so that block-wise tuning is phase continuous."""
    mod_period = input_sample_rate // offset_freq
    per_block = (input_chunk_length // 2) % mod_period
    return 2 * np.pi * np.arange(nblocks) * per_block * (offset_freq / input_sample_rate)


def front_end(raw, input_chunk_length, offset_freq, input_sample_rate, up, dn):
    """main.py:105-166 for one channel held in memory: per block of input_chunk_length raw scalars
    deinterleave -> frequency_shift(offset_freq, block phase) -> resample(up, dn); blocks concatenated."""
    raw = np.asarray(raw)
    nblocks = raw.shape[0] // input_chunk_length
    ph = block_phase_offsets(nblocks, input_chunk_length, input_sample_rate, offset_freq)
    out = []
    for i in range(nblocks):
        blk = deinterleave_IQ(raw[i * input_chunk_length:(i + 1) * input_chunk_length])
        tuned = frequency_shift(blk, offset_freq, input_sample_rate, np.array([ph[i]]))
        out.append(resample(tuned, up, dn))
    return np.concatenate(out)


# --------------------------------------------------------------------------
# CFAR (SURVEY 8f "next" #3): target_detection.py:683-703
# --------------------------------------------------------------------------

def CFAR_2D(X, fw, gw, thresh=None):
    """target_detection.py:683-703 with scipy.signal.convolve2d(mode='same', boundary='wrap') spelled
    out:  box[i,j] = sum_{a,b} T[a,b] X[(i + (fw-1)//2 - a) mod H, (j + (fw-1)//2 - b) mod W],
    T = 1/(fw^2-gw^2) outside the guard block [e1,e2)^2, e1 = (fw-gw)//2, e2 = fw-e1+1."""
    X = np.asarray(X)
    e1 = (fw - gw) // 2
    e2 = fw - e1 + 1
    c = (fw - 1) // 2
    box = np.zeros(X.shape, dtype=np.float64)
    Xd = X.astype(np.float64)
    for a in range(fw):
        for b in range(fw):
            if e1 <= a < e2 and e1 <= b < e2:
                continue
            box += np.roll(np.roll(Xd, a - c, axis=0), b - c, axis=1)     # X[i + c - a, j + c - b]
    box /= (fw ** 2 - gw ** 2)
    cr = (X / np.mean(np.abs(X).flatten())) / (box + 1e-10)
    return cr if thresh is None else cr > thresh


# --------------------------------------------------------------------------
# Channel offset estimation (SURVEY 8f "next" #2): signal_utils.py:73-78
# --------------------------------------------------------------------------
# scipy.signal.decimate(x, q) with its defaults (SciPy 1.15.3 _signaltools.py: ftype='iir', n=8,
# zero_phase=True) is  sosfiltfilt(cheby1(8, 0.05, 0.8/q, output='sos'), x)[::q].  Restated below from the
# published algorithms: Chebyshev-I analog prototype + bilinear transform, sosfilt_zi's steady state,
# sosfiltfilt's odd extension of 3*(2*n_sections+1) samples, transposed-direct-form-II recursion
# (oracle/c/oracle.c:orc_sosfilt, or the Python loop for small inputs).

def cheby1_lowpass_sos(order, rp, wn):
    """Second-order sections [b0 b1 b2 1 a1 a2] of scipy.signal.cheby1(order, rp, wn) (even order): conjugate
    pole pairs, a double zero at z=-1 per section, the gain on the first.  Section order does not change
    the response (SciPy orders by pole radius)."""
    eps = np.sqrt(10.0 ** (0.1 * rp) - 1.0)
    mu = np.arcsinh(1.0 / eps) / order
    theta = np.pi * np.arange(-order + 1, order, 2) / (2 * order)
    p = -np.sinh(mu + 1j * theta)                       # analog prototype poles
    k = np.prod(-p).real
    if order % 2 == 0:
        k /= np.sqrt(1.0 + eps * eps)
    warped = 4.0 * np.tan(np.pi * wn / 2.0)             # pre-warp, fs = 2
    p = p * warped
    k = k * warped ** order
    pz = (4.0 + p) / (4.0 - p)                          # bilinear
    kz = k * np.real(1.0 / np.prod(4.0 - p))
    assert order % 2 == 0
    up = sorted([q for q in pz if q.imag > 0], key=lambda q: abs(q))
    sos = np.zeros((order // 2, 6))
    for i, q in enumerate(up):
        sos[i] = [1.0, 2.0, 1.0, 1.0, -2.0 * q.real, abs(q) ** 2]
    sos[0, :3] *= kz
    return sos


def sosfilt_zi(sos):
    """scipy.signal.sosfilt_zi: per-section lfilter_zi (step-response steady state), each scaled by the DC
    gain of the sections before it."""
    zi = np.zeros((sos.shape[0], 2))
    scale = 1.0
    for s, (b0, b1, b2, a0, a1, a2) in enumerate(sos):
        # (I - A^T) zi = B with A the companion matrix of a, B = b[1:] - a[1:] b0
        m = np.array([[1.0 + a1, -1.0], [a2, 1.0]])
        zi[s] = scale * np.linalg.solve(m, np.array([b1 - a1 * b0, b2 - a2 * b0]))
        scale *= (b0 + b1 + b2) / (a0 + a1 + a2)
    return zi


def _sosfilt(sos, x, zi):
    try:
        from . import c_oracle
    except ImportError:                                   # imported as a top-level module
        import c_oracle
    try:
        return c_oracle.sosfilt(sos, x, zi)
    except OSError:
        y = np.array(x, dtype=np.complex128)
        z = np.array(zi, dtype=np.complex128)
        for i in range(y.shape[0]):
            cur = y[i]
            for s in range(sos.shape[0]):
                nw = sos[s, 0] * cur + z[s, 0]
                z[s, 0] = sos[s, 1] * cur - sos[s, 4] * nw + z[s, 1]
                z[s, 1] = sos[s, 2] * cur - sos[s, 5] * nw
                cur = nw
            y[i] = cur
        return y, z


def decimate_iir(x, q):
    """scipy.signal.decimate(x, q) as find_channel_offset calls it (signal_utils.py:75-76), in complex128
    (the reference's recursion runs in the input's complex64)."""
    if int(q) != q or q < 1:
        raise ValueError("q must be a positive integer")
    x = np.asarray(x)
    sos = cheby1_lowpass_sos(8, 0.05, 0.8 / int(q))
    edge = 3 * (2 * sos.shape[0] + 1)
    if x.shape[0] <= edge:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % edge)
    xd = x.astype(np.complex128)
    ext = np.concatenate((2 * xd[0] - xd[edge:0:-1], xd, 2 * xd[-1] - xd[-2:-edge - 2:-1]))
    zi = sosfilt_zi(sos)
    y, _ = _sosfilt(sos, ext, zi * ext[0])
    y = y[::-1]
    y, _ = _sosfilt(sos, y, zi * y[0])
    return y[::-1][edge:-edge][::int(q)]


def channel_xcorr(B1, B2, nl):
    """|scipy.signal.correlate(B1, pad(B2, nl), 'valid')| when the padded B2 is the longer input:
    xc[i] = |sum_l B1[l] conj(B2[l + m2 - m1 + nl - i])|, i = 0 .. m2 + 2 nl - m1."""
    m1, m2 = B1.shape[0], B2.shape[0]
    K = m2 + 2 * nl - m1
    if K < 0:
        raise ValueError("decimate(s2) padded by nl is shorter than decimate(s1)")
    size = 1
    while size < m2 + 2 * nl:
        size *= 2
    pad = np.zeros(size, dtype=np.complex128)
    pad[nl:nl + m2] = B2
    c = np.fft.ifft(np.fft.fft(pad) * np.conj(np.fft.fft(B1, size)))[:K + 1]     # c[k] = sum_l pad[l+k] conj(B1[l])
    return np.abs(c[::-1])


def find_channel_offset(s1, s2, nd, nl, return_xc=False):
    """signal_utils.py:73-78."""
    B1 = decimate_iir(s1, nd)
    B2 = decimate_iir(s2, nd)
    xc = channel_xcorr(B1, B2, int(nl))
    off = (int(np.argmax(xc)) - int(nl)) * int(nd)
    return (off, xc) if return_xc else off


def find_channel_offset_libcalls(s1, s2, nd, nl):
    """The same three SciPy calls the reference makes (the CPU baseline of the start-up step)."""
    from scipy import signal
    B1 = signal.decimate(s1, nd)
    B2 = np.pad(signal.decimate(s2, nd), (nl, nl), "constant")
    xc = np.abs(signal.correlate(B1, B2, mode="valid"))
    return (np.argmax(xc) - nl) * nd
