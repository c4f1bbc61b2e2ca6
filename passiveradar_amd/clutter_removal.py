"""Drop-in for the clutter filters of the reference's ``passiveRadar/clutter_removal.py`` that
are on the north-star path: LS_Filter (:6-56), LS_Filter_Toeplitz (:109-160),
LS_Filter_Multiple (:162-187), NLMS_filter (:189-249).  Same signatures, return dtypes and
ValueError on mismatched inputs; the arithmetic runs in libprcore.so (complex64 streams,
complex128 Levinson solve on device).  LS_Filter_SVD and GAL_JPE are not on the path
(never called by the reference) and are deliberately absent.
"""
from __future__ import annotations

import numpy as np

from . import _lib, engine

__all__ = ["LS_Filter", "LS_Filter_Toeplitz", "LS_Filter_Multiple", "NLMS_filter", "set_default_ls_method"]

_LS_METHOD = {"m": 0}     # 0 auto | 1 time-domain kernels | 2 FFT kernels (tests flip it)


def set_default_ls_method(method):
    _LS_METHOD["m"] = int(method)


def _check_same(ref, srv):
    if tuple(ref.shape) != tuple(srv.shape):
        raise ValueError("Input vectors must have the same length")


def _ls_run(ref, srv, filterLen, peek, circular, sampleRate, bins, reg, want_taps):
    ref = np.ascontiguousarray(ref, dtype=np.complex64)
    srv = np.ascontiguousarray(srv, dtype=np.complex64)
    n = ref.shape[0]
    T = int(filterLen) + int(peek)
    meth = _LS_METHOD["m"]
    plan = engine.cached_plan(("ls", n, int(filterLen), int(peek), bool(circular), meth),
                              lambda: engine.LsPlan(n, filterLen, peek, circular, 1, meth))
    st = engine.staging()
    d_ref = st.get("ls_ref", 8 * n)
    d_srv = st.get("ls_srv", 8 * n)
    d_out = st.get("ls_out", 8 * n)
    d_taps = st.get("ls_taps", 16 * T)
    d_ref.upload(ref)
    d_srv.upload(srv)
    plan.execute(d_ref, d_srv, d_out, 1, n, n, sampleRate, bins, reg, d_taps)
    out = d_out.download((n,), np.complex64)
    taps = d_taps.download((T,), np.complex128) if want_taps else None
    return out, taps


def LS_Filter_Toeplitz(refChannel, srvChannel, filterLen, peek=10, return_filter=False):
    """Block LS canceller, Toeplitz/Levinson form (:109-160).  Returns complex128 like the reference
    (values are the device's complex64 stream; taps are complex128 from the fp64 solve)."""
    _check_same(refChannel, srvChannel)
    out, taps = _ls_run(refChannel, srvChannel, filterLen, peek, False, 1.0, (0.0,), 0.0, return_filter)
    out = out.astype(np.complex128)
    return (out, taps) if return_filter else out


def LS_Filter_Multiple(refChannel, srvChannel, filterLen, sampleRate, dopplerBins=[0]):
    """LS_Filter_Toeplitz chained over Doppler bins (:162-187); the whole chain stays on the GPU."""
    _check_same(refChannel, srvChannel)
    bins = [float(b) for b in dopplerBins]
    if not bins:
        return srvChannel
    out, _ = _ls_run(refChannel, srvChannel, filterLen, 10, False, float(sampleRate), bins, 0.0, False)
    return out.astype(np.complex128)


def LS_Filter(refChannel, srvChannel, filterLen, reg=1.0, peek=10, return_filter=False):
    """Direct-matrix block LS (:6-56) without forming the N x T matrix: its Gram matrix is the
    circular-autocorrelation Toeplitz matrix, so this is the Toeplitz path with circular indexing
    and ``reg`` on the diagonal.  complex64 out and taps, like the reference."""
    _check_same(refChannel, srvChannel)
    out, taps = _ls_run(refChannel, srvChannel, filterLen, peek, True, 1.0, (0.0,), float(reg),
                        return_filter)
    return (out, taps.astype(np.complex64)) if return_filter else out


def NLMS_filter(refChannel, srvChannel, filterLen, mu, peek=10, initialTaps=None, returnFilter=False):
    """Normalised LMS canceller (:189-249), one wavefront on the GPU.  complex64 out/taps."""
    ref = np.ascontiguousarray(refChannel, dtype=np.complex64)
    srv = np.ascontiguousarray(srvChannel, dtype=np.complex64)
    if ref.shape[0] < srv.shape[0]:
        raise IndexError("refChannel shorter than srvChannel")
    n = srv.shape[0]
    st = engine.staging()
    d_tin = None
    if initialTaps is not None:                               # :218-225
        taps0 = np.ascontiguousarray(initialTaps, dtype=np.complex64)
        filterLen = taps0.shape[0] - peek
        d_tin = st.get("nlms_tin", 8 * taps0.shape[0])
        d_tin.upload(taps0)
    T = int(filterLen) + int(peek)
    d_ref = st.get("nlms_ref", 8 * n)
    d_srv = st.get("nlms_srv", 8 * n)
    d_out = st.get("nlms_out", 8 * n)
    d_tout = st.get("nlms_tout", 8 * T)
    d_ref.upload(ref[:n])
    d_srv.upload(srv)
    engine.nlms_execute(d_ref, d_srv, d_out, n, filterLen, mu, peek, d_tin, d_tout, 1)
    out = d_out.download((n,), np.complex64)
    if returnFilter:
        return out, d_tout.download((T,), np.complex64)
    return out
