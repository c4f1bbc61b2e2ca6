"""Drop-ins for the two ``passiveRadar/signal_utils.py`` helpers that sit on the hot path."""
from __future__ import annotations

import numpy as np

from . import _lib, engine
from ._lib import check, lib

__all__ = ["xcorr", "frequency_shift", "deinterleave_IQ", "resample", "front_end", "find_channel_offset",
           "decimate_iir"]


def _xcorr_equal_lengths(s1, s2, nlead, nlag):
    """xcorr of two signals of different lengths as the equal-length sum on zero-extended copies (host logic, no device
    call): returns (e1, e2, nlead', nlag') with xcorr(s1, s2, nlead, nlag) == xcorr(e1, e2, nlead', nlag')"""
    n1, n2 = s1.shape[0], s2.shape[0]
    m = n2 + nlag + nlead
    K = abs(m - n1)
    if m >= n1:
        # offsets K - nlag - i of s2 against s1: lead' = K - nlag (when negative, s2 is delayed by that much first)
        lead = K - nlag
        front = max(-lead, 0)
        e2 = np.zeros(max(n1, n2 + front), np.complex64)
        e2[front:front + n2] = s2
        e1 = np.zeros(e2.shape[0], np.complex64)
        e1[:n1] = s1
        return e1, e2, lead + front, K - (lead + front)
    e2 = np.zeros(n1, np.complex64)                         # the padded s2, zero-extended to len(s1)
    e2[nlag:nlag + n2] = s2
    return s1, e2, 0, K


def xcorr(s1, s2, nlead, nlag):
    """signal_utils.py:29-32: ``correlate(s1, pad(s2, (nlag, nlead)), mode='valid')``, complex64.

    Equal lengths (every call site of the reference): z[i] = sum_n s1[n] conj(s2[n-(i-nlead)]), i = 0..nlag+nlead.
    The reference's expression accepts ANY two lengths; with m = len(s2) + nlag + nlead and K = |m - len(s1)| SciPy's
    'valid' mode returns K + 1 values:
      m >= len(s1):  z[i] = sum_l s1[l]   conj(s2[l + K - i - nlag]),  i = 0..K   (the padded s2 slides over s1)
      m <  len(s1):  z[k] = sum_l s1[l+k] conj(s2[l - nlag]),          k = 0..K   (s1 slides over the padded s2)
    Both are the equal-length sum on zero-extended copies (zeros add nothing), which is how they reach prc_xcorr."""
    s1 = np.ascontiguousarray(s1, dtype=np.complex64)
    s2 = np.ascontiguousarray(s2, dtype=np.complex64)
    if s1.ndim != 1 or s2.ndim != 1:
        raise ValueError("xcorr takes one-dimensional signals")
    nlead, nlag = int(nlead), int(nlag)
    if nlead < 0 or nlag < 0:
        raise ValueError("index can't contain negative values")          # np.pad's complaint about (nlag, nlead)
    if s1.shape[0] != s2.shape[0]:
        s1, s2, nlead, nlag = _xcorr_equal_lengths(s1, s2, nlead, nlag)
    n1 = s1.shape[0]
    n = n1
    st = engine.staging()
    d1 = st.get("xc_1", 8 * n)
    d2 = st.get("xc_2", 8 * n)
    do = st.get("xc_o", 8 * (nlag + nlead + 1))
    d1.upload(s1)
    d2.upload(s2)
    check(lib().prc_xcorr(d1.ptr, d2.ptr, n, nlead, nlag, do.ptr, None))
    return do.download((nlag + nlead + 1,), np.complex64)


def frequency_shift(x, fc, Fs, phase_offset=0):
    """signal_utils.py:24-27 with the reference's float32 phase ramp.  phase_offset: a scalar (complex64 result), or an
    array NumPy broadcasts against x -- one value (main.py:133-149: the starting phase of a dask block) or one per
    sample; a float64 / integer array promotes the result to complex128 (float32 ramp + double phase, exponential in
    double), a float32 array keeps it complex64, as in the reference."""
    if np.ndim(phase_offset) != 0:
        ph = np.asarray(phase_offset)
        x = np.ascontiguousarray(x, dtype=np.complex64)
        n = x.shape[0]
        if ph.ndim != 1 or ph.shape[0] not in (1, n) or x.ndim != 1:
            raise ValueError(f"operands could not be broadcast together with shapes ({n},) {ph.shape}")
        f32 = ph.dtype in (np.float32, np.float16)
        st = engine.staging()
        dx = st.get("fs_x", 8 * n)
        dx.upload(x)
        if ph.shape[0] == 1 and not f32:
            dy = st.get("fs_y128", 16 * n)
            check(lib().prc_frequency_shift_block(dx.ptr, dy.ptr, n, float(fc), float(Fs), float(ph[0]), None))
            return dy.download((n,), np.complex128)
        ph = np.ascontiguousarray(np.broadcast_to(ph, (n,)), dtype=np.float32 if f32 else np.float64)
        dp = st.get("fs_ph", ph.nbytes)
        dp.upload(ph)
        dy = st.get("fs_y128", (8 if f32 else 16) * n)
        check(lib().prc_frequency_shift_phases(dx.ptr, dy.ptr, n, float(fc), float(Fs), dp.ptr, int(f32), None))
        return dy.download((n,), np.complex64 if f32 else np.complex128)
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
