"""Thin Python owners of the libprcore plans (device-pointer level).

Everything here takes raw device pointers (ints) + a hipStream_t, so it works both for the
NumPy-facing drop-in functions (which stage through ``_lib.DeviceBuffer``) and for callers that
already hold torch device tensors (``tensor.data_ptr()``; bench.py, stream.py).
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import check, lib


def _ptr(x):
    """int / c_void_p / DeviceBuffer / torch tensor -> c_void_p"""
    if x is None:
        return None
    if isinstance(x, _lib.DeviceBuffer):
        return C.c_void_p(x.ptr)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if isinstance(x, C.c_void_p):
        return x
    return C.c_void_p(int(x))


class CafPlan:
    """prc_caf_plan: fast_xambg for up to ``max_frames`` frames per launch."""

    def __init__(self, n, range_bins, freq_bins, max_frames=1, method=_lib.CAF_AUTO,
                 doppler=_lib.DOPPLER_AUTO, taps=None, multi=_lib.CAF_MULTI_AUTO):
        self.n, self.range_bins, self.freq_bins = int(n), int(range_bins), int(freq_bins)
        self.max_frames = int(max_frames)
        d = _lib.CafDesc()
        d.n, d.range_bins, d.freq_bins = self.n, self.range_bins, self.freq_bins
        d.max_frames, d.method, d.doppler = self.max_frames, int(method), int(doppler)
        d.multi = _lib.CAF_MULTI_MODES[multi] if isinstance(multi, str) else int(multi)
        self._taps = None
        if taps is not None:
            self._taps = np.ascontiguousarray(taps, dtype=np.float32)
            d.ntaps = self._taps.size
            d.taps_host = self._taps.ctypes.data_as(C.POINTER(C.c_float))
        else:
            d.ntaps = 0
            d.taps_host = None
        h = C.c_void_p()
        check(lib().prc_caf_plan_create(C.byref(h), C.byref(d)))
        self._h = h
        m, dp, ws = C.c_int32(), C.c_int32(), C.c_int64()
        check(lib().prc_caf_plan_info(self._h, C.byref(m), C.byref(dp), C.byref(ws)))
        self.method, self.doppler, self.workspace_bytes = m.value, dp.value, ws.value
        mm = C.c_int32()
        check(lib().prc_caf_plan_multi_mode(self._h, C.byref(mm)))
        self.multi = mm.value               # what AUTO resolved to (turns / shared / pairs)

    @property
    def out_shape(self):
        return (self.freq_bins, self.range_bins + 1)

    def execute(self, ref, srv, out, nframes=1, frame_stride=None, n_valid=None, window=None,
                stream=None):
        check(lib().prc_caf_execute(self._h, _ptr(ref), _ptr(srv),
                                    self.n if frame_stride is None else int(frame_stride),
                                    self.n if n_valid is None else int(n_valid),
                                    _ptr(window), _ptr(out), int(nframes), stream))

    def execute_multi(self, refs, srv, outs, nframes=1, frame_stride=None, n_valid=None, window=None, stream=None):
        """prc_caf_execute_multi: every reference channel of ``refs`` against ONE surveillance channel in one call;
        ``outs[i]`` receives what ``execute(refs[i], srv, ...)`` would write (len(refs) * nframes <= max_frames)."""
        nref = len(refs)
        rp = (C.c_void_p * nref)(*[_ptr(r) for r in refs])
        op = (C.c_void_p * nref)(*[_ptr(o) for o in outs])
        check(lib().prc_caf_execute_multi(self._h, rp, nref, _ptr(srv),
                                          self.n if frame_stride is None else int(frame_stride),
                                          self.n if n_valid is None else int(n_valid),
                                          _ptr(window), op, int(nframes), stream))

    def execute_segments(self, ref, srv, nframes=1, frame_stride=None, n_valid=None, window=None,
                         stream=None):
        check(lib().prc_caf_execute_segments(self._h, _ptr(ref), _ptr(srv),
                                             self.n if frame_stride is None else int(frame_stride),
                                             self.n if n_valid is None else int(n_valid),
                                             _ptr(window), int(nframes), stream))

    def execute_doppler(self, out, nframes=1, stream=None):
        check(lib().prc_caf_execute_doppler(self._h, _ptr(out), int(nframes), stream))

    def close(self):
        if getattr(self, "_h", None):
            lib().prc_caf_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LsPlan:
    """prc_ls_plan: LS_Filter_Toeplitz / LS_Filter_Multiple / LS_Filter on up to max_blocks blocks."""

    def __init__(self, n, filter_len, peek=10, circular=False, max_blocks=1, method=0):
        self.n, self.filter_len, self.peek = int(n), int(filter_len), int(peek)
        self.ntaps = self.filter_len + self.peek
        self.max_blocks = int(max_blocks)
        d = _lib.LsDesc()
        d.n, d.filter_len, d.peek = self.n, self.filter_len, self.peek
        d.circular, d.max_blocks, d.method = int(bool(circular)), self.max_blocks, int(method)
        h = C.c_void_p()
        check(lib().prc_ls_plan_create(C.byref(h), C.byref(d)))
        self._h = h

    def execute(self, ref, srv, out, nblocks=1, stride=None, out_stride=None, sample_rate=1.0,
                doppler_bins=(0,), reg=0.0, taps_out=None, stream=None):
        bins = (C.c_double * len(doppler_bins))(*[float(b) for b in doppler_bins])
        check(lib().prc_ls_execute(self._h, _ptr(ref), _ptr(srv),
                                   self.n if stride is None else int(stride), _ptr(out),
                                   self.n if out_stride is None else int(out_stride), int(nblocks),
                                   float(sample_rate), bins, len(doppler_bins), float(reg),
                                   _ptr(taps_out), stream))

    def set_profiling(self, enable=True):
        check(lib().prc_ls_set_profiling(self._h, int(bool(enable))))

    def get_profile(self):
        """((ms_corr, ms_solve, ms_fir), (launches_corr, launches_solve, launches_fir)) of the last
        execute; in the cached chain the correlation of bin i+1 runs inside the FIR kernel of bin i"""
        ms = (C.c_double * 3)()
        k = (C.c_int32 * 3)()
        check(lib().prc_ls_get_profile(self._h, ms, k))
        return (ms[0], ms[1], ms[2]), (k[0], k[1], k[2])

    def close(self):
        if getattr(self, "_h", None):
            lib().prc_ls_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cheby1_lowpass_zpk(order, rp, wn):
    """Digital Chebyshev type-I low-pass as scipy.signal.cheby1(order, rp, wn, output='zpk') designs it:
    analog prototype poles -sinh(mu + j theta_m), cut-off pre-warped to 4 tan(pi wn / 2) (fs = 2), bilinear
    transform; all zeros at z = -1.  Returns (zeros, poles, gain)."""
    eps = np.sqrt(10.0 ** (0.1 * rp) - 1.0)
    mu = np.arcsinh(1.0 / eps) / order
    theta = np.pi * np.arange(-order + 1, order, 2) / (2 * order)
    p = -np.sinh(mu + 1j * theta)
    k = np.prod(-p).real
    if order % 2 == 0:
        k /= np.sqrt(1.0 + eps * eps)
    warped = 4.0 * np.tan(np.pi * wn / 2.0)
    p = p * warped
    k = k * warped ** order
    pz = (4.0 + p) / (4.0 - p)
    kz = k * np.real(1.0 / np.prod(4.0 - p))
    return -np.ones(order, dtype=np.complex128), pz, float(kz)


class IirDecimator:
    """prc_iir_desc of scipy.signal.decimate(x, q) with its defaults (ftype='iir', n=8, zero_phase=True):
    cheby1(8, 0.05, 0.8/q) through sosfiltfilt, whose odd extension is 3*(2*4+1) = 27 samples."""

    def __init__(self, q, order=8):
        if int(q) != q or q < 1:
            raise ValueError("q must be a positive integer")       # scipy: operator.index(q)
        z, p, k = cheby1_lowpass_zpk(order, 0.05, 0.8 / int(q))
        nsec = (order + 1) // 2
        self.q = int(q)
        self.padlen = 3 * (2 * nsec + 1)
        # |pole|^settle < 1e-9: the recursion's memory, replaced by an explicit constant extension
        self.settle = int(np.ceil(np.log(1e-9) / np.log(np.abs(p).max()) / 64.0)) * 64
        self._z = np.ascontiguousarray(np.stack([z.real, z.imag], -1).reshape(-1))
        self._p = np.ascontiguousarray(np.stack([p.real, p.imag], -1).reshape(-1))
        d = _lib.IirDesc()
        d.q, d.padlen, d.settle, d.nzeros, d.npoles, d.gain = self.q, self.padlen, self.settle, z.size, p.size, k
        d.zeros_host = self._z.ctypes.data_as(C.POINTER(C.c_double))
        d.poles_host = self._p.ctypes.data_as(C.POINTER(C.c_double))
        self.desc = d

    def out_len(self, n):
        return -(-int(n) // self.q)

    def decimate(self, x, n, y, stream=None):
        """y[j] = sosfiltfilt(x)[j q] (complex64 device buffers)"""
        check(lib().prc_decimate_iir(_ptr(x), int(n), C.byref(self.desc), _ptr(y), stream))

    def channel_offset(self, s1, n1, s2, n2, nl, xc_out=None, stream=None):
        """(argmax index, number of lags) of |correlate(decimate(s1), pad(decimate(s2), nl), 'valid')|"""
        n_xc, am = C.c_int64(), C.c_int64()
        check(lib().prc_channel_offset(_ptr(s1), int(n1), _ptr(s2), int(n2), C.byref(self.desc), int(nl),
                                       _ptr(xc_out), C.byref(n_xc), C.byref(am), stream))
        return am.value, n_xc.value

    def n_lags(self, n1, n2, nl):
        return self.out_len(n2) + 2 * int(nl) - self.out_len(n1) + 1

    def close(self):
        """nothing device-side to release (the plan cache calls this on eviction)"""


def resample_design(up, dn):
    """scipy.signal.resample_poly's filter and alignment for signal_utils.resample (signal_utils.py:15-17):
    returns (taps with n_pre_pad leading zeros, float32; n_pre_remove; up; dn) with up/dn reduced."""
    from math import gcd
    from scipy.signal import firwin
    g = gcd(int(up), int(dn))
    up, dn = int(up) // g, int(dn) // g
    max_rate = max(up, dn)
    half_len = 10 * max_rate
    h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)) * up
    n_pre_pad = dn - half_len % dn
    taps = np.concatenate((np.zeros(n_pre_pad), h)).astype(np.float32)
    return taps, (half_len + n_pre_pad) // dn, up, dn


class FrontendPlan:
    """prc_frontend_plan: deinterleave -> tune (block phase) -> resample, fused, up to max_blocks blocks."""

    def __init__(self, n_in, raw_dtype, up, dn, max_blocks=1):
        taps, npr, up, dn = resample_design(up, dn)
        self.n_in, self.up, self.dn = int(n_in), up, dn
        self.raw_is_complex = str(raw_dtype) == "complex64"     # raw_stride then counts complex samples
        self._taps = np.ascontiguousarray(taps)
        d = _lib.FrontendDesc()
        d.n_in, d.raw_dtype, d.up, d.down = self.n_in, _lib.RAW_DTYPES[str(raw_dtype)], up, dn
        d.ntaps, d.n_pre_remove, d.max_blocks = self._taps.size, npr, int(max_blocks)
        d.taps_host = self._taps.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()
        check(lib().prc_frontend_plan_create(C.byref(h), C.byref(d)))
        self._h = h
        n = C.c_int64()
        check(lib().prc_frontend_out_len(self._h, C.byref(n)))
        self.n_out = n.value

    def execute(self, raw, out, nblocks=1, raw_stride=None, out_stride=None, fc=0.0, fs=1.0, phases=None,
                mix=True, stream=None):
        ph = None
        if phases is not None:
            ph = (C.c_double * nblocks)(*[float(p) for p in phases])
        itemscalars = 1 if self.raw_is_complex else 2
        check(lib().prc_frontend_execute(self._h, _ptr(raw),
                                         int(self.n_in * itemscalars if raw_stride is None else raw_stride),
                                         int(bool(mix)), float(fc), float(fs), ph, _ptr(out),
                                         int(self.n_out if out_stride is None else out_stride), int(nblocks), stream))

    def execute2(self, raw_a, raw_b, out_a, out_b, nblocks=1, raw_stride=None, out_stride=None, fc=0.0, fs=1.0,
                 phases=None, mix=True, stream=None):
        """prc_frontend_execute2: both channels of every block in one launch (same strides, same phases)"""
        ph = None
        if phases is not None:
            ph = (C.c_double * nblocks)(*[float(p) for p in phases])
        itemscalars = 1 if self.raw_is_complex else 2
        check(lib().prc_frontend_execute2(self._h, _ptr(raw_a), _ptr(raw_b),
                                          int(self.n_in * itemscalars if raw_stride is None else raw_stride),
                                          int(bool(mix)), float(fc), float(fs), ph, _ptr(out_a), _ptr(out_b),
                                          int(self.n_out if out_stride is None else out_stride), int(nblocks), stream))

    def close(self):
        if getattr(self, "_h", None):
            lib().prc_frontend_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def nlms_execute(ref, srv, out, n, filter_len, mu, peek=10, taps_in=None, taps_out=None,
                 nstreams=1, stride=None, out_stride=None, stream=None):
    check(lib().prc_nlms_execute(_ptr(ref), _ptr(srv), int(n), int(n if stride is None else stride),
                                 int(filter_len), int(peek), float(mu), _ptr(taps_in), _ptr(out),
                                 int(n if out_stride is None else out_stride), _ptr(taps_out),
                                 int(nstreams), stream))


# ---- per-thread plan cache (dask-style concurrent callers each get their own plans) ------
_tls = threading.local()


def cached_plan(key, factory, limit=8, stream=None):
    """One plan per (thread, device, stream, shape key): a plan's workspaces live on the device that was current
    when it was made, and its single slow-time / tap workspace must not be shared by launches on two streams."""
    key = (_lib.current_device(), getattr(stream, "value", stream)) + tuple(key)
    cache = getattr(_tls, "plans", None)
    if cache is None:
        cache = _tls.plans = {}
    plan = cache.get(key)
    if plan is None:
        if len(cache) >= limit:
            old_key = next(iter(cache))
            cache.pop(old_key).close()
        plan = cache[key] = factory()
    return plan


class Staging:
    """Per-thread grow-only device scratch for the NumPy-facing functions."""

    def __init__(self):
        self.bufs = {}

    def get(self, name, nbytes):
        b = self.bufs.get(name)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = self.bufs[name] = _lib.DeviceBuffer(nbytes)
        return b


def staging():
    """per thread and per device"""
    per_dev = getattr(_tls, "staging", None)
    if per_dev is None:
        per_dev = _tls.staging = {}
    dev = _lib.current_device()
    st = per_dev.get(dev)
    if st is None:
        st = per_dev[dev] = Staging()
    return st
