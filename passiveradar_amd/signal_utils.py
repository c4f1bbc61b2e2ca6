"""Drop-ins for the two ``passiveRadar/signal_utils.py`` helpers that sit on the hot path."""
from __future__ import annotations

import numpy as np

from . import _lib, engine
from ._lib import check, lib

__all__ = ["xcorr", "frequency_shift", "deinterleave_IQ", "resample", "front_end", "find_channel_offset",
           "decimate_iir"]


def xcorr(s1, s2, nlead, nlag):
    """signal_utils.py:29-32: z[i] = sum_n s1[n] conj(s2[n-(i-nlead)]), i = 0..nlag+nlead (complex64)."""
    s1 = np.ascontiguousarray(s1, dtype=np.complex64)
    s2 = np.ascontiguousarray(s2, dtype=np.complex64)
    if s1.shape != s2.shape:
        raise ValueError("Input vectors must have the same length")
    n = s1.shape[0]
    st = engine.staging()
    d1 = st.get("xc_1", 8 * n)
    d2 = st.get("xc_2", 8 * n)
    do = st.get("xc_o", 8 * (nlag + nlead + 1))
    d1.upload(s1)
    d2.upload(s2)
    check(lib().prc_xcorr(d1.ptr, d2.ptr, n, int(nlead), int(nlag), do.ptr, None))
    return do.download((nlag + nlead + 1,), np.complex64)


def frequency_shift(x, fc, Fs, phase_offset=0):
    """signal_utils.py:24-27 with the reference's float32 phase ramp; scalar phase_offset only
    (the array form is the front end's block-phase trick, main.py:125-149, outside this path)."""
    if np.ndim(phase_offset) != 0:
        # array phase offset (main.py:133-149: one starting phase per dask block): NumPy promotes the
        # result to complex128 -- float32 ramp + double phase, exponential in double
        if np.size(phase_offset) != 1:
            raise ValueError("operands could not be broadcast together: phase_offset must hold one value per block")
        x = np.ascontiguousarray(x, dtype=np.complex64)
        n = x.shape[0]
        st = engine.staging()
        dx = st.get("fs_x", 8 * n)
        dy = st.get("fs_y128", 16 * n)
        dx.upload(x)
        check(lib().prc_frequency_shift_block(dx.ptr, dy.ptr, n, float(fc), float(Fs),
                                              float(np.asarray(phase_offset).reshape(-1)[0]), None))
        return dy.download((n,), np.complex128)
    x = np.ascontiguousarray(x, dtype=np.complex64)
    n = x.shape[0]
    st = engine.staging()
    dx = st.get("fs_x", 8 * n)
    dy = st.get("fs_y", 8 * n)
    dx.upload(x)
    check(lib().prc_frequency_shift(dx.ptr, dy.ptr, n, float(fc), float(Fs), float(phase_offset), None))
    return dy.download((n,), np.complex64)


def deinterleave_IQ(interleavedIQ):
    """signal_utils.py:19-22: interleaved I,Q scalars (int8 / uint8 / int16 / float32) -> complex64."""
    raw = np.ascontiguousarray(interleavedIQ)
    if str(raw.dtype) not in ("int8", "uint8", "int16", "float32"):
        raw = raw.astype(np.float32)
    n = raw.shape[0] // 2
    st = engine.staging()
    dr = st.get("di_raw", raw.nbytes)
    do = st.get("di_out", 8 * max(n, 1))
    dr.upload(raw)
    check(lib().prc_deinterleave(dr.ptr, _lib.RAW_DTYPES[str(raw.dtype)], n, do.ptr, None))
    return do.download((n,), np.complex64)


def decimate_iir(x, q):
    """scipy.signal.decimate(x, q) with its defaults, as find_channel_offset uses it (signal_utils.py:75-76):
    zero-phase order-8 Chebyshev-I low-pass, every q-th sample, complex64."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    n = x.shape[0]
    dec = engine.cached_plan(("iirdec", int(q)), lambda: engine.IirDecimator(q))
    st = engine.staging()
    dx = st.get("dec_x", 8 * max(n, 1))
    dy = st.get("dec_y", 8 * max(dec.out_len(n), 1))
    dx.upload(x)
    dec.decimate(dx, n, dy)
    return dy.download((dec.out_len(n),), np.complex64)


def find_channel_offset(s1, s2, nd, nl, return_xc=False):
    """signal_utils.py:73-78: (argmax|correlate(decimate(s1, nd), pad(decimate(s2, nd), nl), 'valid')| - nl)*nd.
    With ``return_xc`` also the correlation magnitudes (float32, m2 + 2 nl - m1 + 1 lags)."""
    s1 = np.ascontiguousarray(s1, dtype=np.complex64)
    s2 = np.ascontiguousarray(s2, dtype=np.complex64)
    if s1.ndim != 1 or s2.ndim != 1:
        raise ValueError("find_channel_offset takes one-dimensional signals")
    nl = int(nl)
    dec = engine.cached_plan(("iirdec", int(nd)), lambda: engine.IirDecimator(nd))
    n1, n2 = s1.shape[0], s2.shape[0]
    st = engine.staging()
    d1 = st.get("co_1", 8 * max(n1, 1))
    d2 = st.get("co_2", 8 * max(n2, 1))
    d1.upload(s1)
    d2.upload(s2)
    dxc = None
    if return_xc:
        dxc = st.get("co_xc", 4 * max(dec.n_lags(n1, n2, nl), 1))
    am, n_xc = dec.channel_offset(d1, n1, d2, n2, nl, dxc)
    offset = (am - nl) * int(nd)
    if return_xc:
        return offset, dxc.download((n_xc,), np.float32)
    return offset


def resample(x, up, dn):
    """signal_utils.py:15-17: rational resampling, scipy.signal.resample_poly(x, up, dn, padtype='line').
    The stream is complex64 on the device; the result is returned in the input's dtype (the reference
    keeps complex128 when fed the tuned complex128 stream)."""
    xin = np.asarray(x)
    from math import gcd
    g = gcd(int(up), int(dn))
    if int(up) // g == 1 and int(dn) // g == 1:
        return xin.copy()                                  # scipy.signal.resample_poly: up == down -> a copy
    xc = np.ascontiguousarray(xin, dtype=np.complex64)
    n = xc.shape[0]
    plan = engine.cached_plan(("fe", n, "complex64", int(up), int(dn)),
                              lambda: engine.FrontendPlan(n, "complex64", up, dn, 1))
    st = engine.staging()
    dx = st.get("rs_x", 8 * n)
    do = st.get("rs_o", 8 * plan.n_out)
    dx.upload(xc)
    plan.execute(dx, do, 1, n, plan.n_out, mix=False)
    y = do.download((plan.n_out,), np.complex64)
    return y.astype(xin.dtype) if np.iscomplexobj(xin) else y.real.astype(xin.dtype)


def front_end(raw, input_chunk_length, offset_freq, input_sample_rate, up, dn, max_blocks=16):
    """main.py:105-166 for one channel: per block of ``input_chunk_length`` raw scalars
    deinterleave -> tune by ``offset_freq`` with the block starting phase (main.py:125-130) ->
    resample(up, dn), ONE fused kernel per batch of blocks.  Returns the concatenated complex64 IF stream."""
    raw = np.ascontiguousarray(raw)
    if str(raw.dtype) not in ("int8", "uint8", "int16", "float32"):
        raw = raw.astype(np.float32)
    icl = int(input_chunk_length)
    nblocks = raw.shape[0] // icl
    n_in = icl // 2
    mod_period = input_sample_rate // offset_freq
    per_block = n_in % mod_period
    phases = 2 * np.pi * np.arange(nblocks) * per_block * (offset_freq / input_sample_rate)
    plan = engine.cached_plan(("fe", n_in, str(raw.dtype), int(up), int(dn), max_blocks),
                              lambda: engine.FrontendPlan(n_in, str(raw.dtype), up, dn, max_blocks))
    st = engine.staging()
    dr = st.get("fe_raw", raw.nbytes)
    do = st.get("fe_out", 8 * plan.n_out * max(nblocks, 1))
    dr.upload(raw)
    isz = raw.dtype.itemsize
    for b0 in range(0, nblocks, max_blocks):
        nb = min(max_blocks, nblocks - b0)
        plan.execute(dr.ptr + b0 * icl * isz, do.ptr + 8 * b0 * plan.n_out, nb, icl, plan.n_out,
                     offset_freq, input_sample_rate, phases[b0:b0 + nb], True)
    return do.download((nblocks * plan.n_out,), np.complex64)
