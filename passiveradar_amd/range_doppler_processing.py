"""Drop-in for the reference's ``passiveRadar/range_doppler_processing.py`` on MI355X.

``fast_xambg`` keeps the reference signature, argument meaning, return shape/dtype and the
ValueError on mismatched inputs (range_doppler_processing.py:12-90) and runs on the HIP kernels
behind libprcore.so.  Inputs may be NumPy arrays (staged to the GPU and back) or torch device
tensors (zero copy, result returned as a torch tensor on the same device).
"""
from __future__ import annotations

import functools

import numpy as np

from . import _lib, engine

__all__ = ["fast_xambg", "fast_xambg_multi", "direct_xambg", "caf_plan_for", "set_default_methods"]

# Kernel selection used by fast_xambg (0 = let the plan decide).  Not part of the reference
# signature; tests flip it to run the same cases through every kernel family.
_DEFAULTS = {"caf": _lib.CAF_AUTO, "doppler": _lib.DOPPLER_AUTO}


def set_default_methods(caf=None, doppler=None):
    """caf: 0 auto | 1 direct | 2 fft (1024-point) | 3 fft (4096-point team);
    doppler: 0 auto | 1 transpose + rocfft + shift | 2 one column-FFT kernel (freq_bins 256..4096, powers of two)."""
    if caf is not None:
        _DEFAULTS["caf"] = int(caf)
    if doppler is not None:
        _DEFAULTS["doppler"] = int(doppler)


@functools.lru_cache(maxsize=16)
def _named_window(spec, n):
    # range_doppler_processing.py:57-58 -- host side, computed once per (spec, length)
    from scipy.signal import get_window
    return np.ascontiguousarray(get_window(spec, n), dtype=np.float32)


@functools.lru_cache(maxsize=4)
def _long_taps(q):
    # range_doppler_processing.py:76 -- flat-top FIR of 10q+1 taps (shortFilt=False)
    from scipy.signal import firwin
    return np.ascontiguousarray(firwin(10 * q + 1, 1.0 / q, window="flattop"), dtype=np.float32)


def caf_plan_for(n, rangeBins, freqBins, shortFilt=True, max_frames=1, method=None, doppler=None, stream=None,
                 multi=_lib.CAF_MULTI_AUTO):
    method = _DEFAULTS["caf"] if method is None else method
    doppler = _DEFAULTS["doppler"] if doppler is None else doppler
    if not shortFilt and (method == _lib.CAF_FFT4096 or (method == _lib.CAF_FFT and rangeBins + 1 > 769)):
        method = _lib.CAF_AUTO          # the long FIR runs on the 1024-point FFT kernel up to 769 lags, else time-domain
    q = int(n / freqBins) if freqBins else 0
    multi = _lib.CAF_MULTI_MODES[multi] if isinstance(multi, str) else int(multi)
    # plans copy the process-wide options when they are made: a cached plan must not outlive a change of them
    opts = (_lib.get_option(_lib.OPT_CAF_MULTI_MODE), _lib.get_option(_lib.OPT_CAF_GROUP_MB))
    key = ("caf", n, rangeBins, freqBins, bool(shortFilt), max_frames, method, doppler, multi) + opts
    taps = None if shortFilt else _long_taps(q)
    return engine.cached_plan(key, lambda: engine.CafPlan(n, rangeBins, freqBins, max_frames,
                                                          method, doppler, taps, multi), stream=stream)


def fast_xambg(refChannel, srvChannel, rangeBins, freqBins, inputLen=None, window=None,
               shortFilt=True):
    """Fast cross-ambiguity function (same contract as the reference, :12-90).

    Returns an array of shape (freqBins, rangeBins+1, 1), complex64: column k is delay
    rangeBins-k samples, rows are fftshift-ed Doppler bins.
    """
    if tuple(refChannel.shape) != tuple(srvChannel.shape):                  # :46-49
        raise ValueError("Input vectors must have the same length")
    n_in = int(refChannel.shape[0])
    n = n_in if inputLen is None else int(inputLen)
    if n_in > n:
        raise ValueError("index can't contain negative values")              # np.pad at :53
    if isinstance(window, (tuple, str)):
        window = _named_window(window, n)
    elif window is not None and not _lib.is_device_tensor(window):
        window = np.ascontiguousarray(window, dtype=np.float32)
        if window.shape[0] != n:
            raise ValueError(f"operands could not be broadcast together with shapes ({n},) "
                             f"({window.shape[0]},)")
    if _lib.is_device_tensor(refChannel):
        import torch
        # everything on the tensors' own device and on torch's current stream THERE (not on whatever device
        # happens to be current): the plan, its workspaces and the launch
        with torch.cuda.device(refChannel.device):
            st = _lib.torch_stream_ptr(refChannel.device)
            plan = caf_plan_for(n, int(rangeBins), int(freqBins), shortFilt, stream=st)
            ref = refChannel.to(torch.complex64).contiguous()
            srv = srvChannel.to(device=ref.device, dtype=torch.complex64).contiguous()
            win = None
            if window is not None:
                win = window if _lib.is_device_tensor(window) else torch.from_numpy(window)
                win = win.to(device=ref.device, dtype=torch.float32).contiguous()
            out = torch.empty((int(freqBins), int(rangeBins) + 1, 1), dtype=torch.complex64,
                              device=ref.device)
            plan.execute(ref, srv, out, 1, n, n_in, win, stream=st)
        return out

    plan = caf_plan_for(n, int(rangeBins), int(freqBins), shortFilt)
    ref = np.ascontiguousarray(refChannel, dtype=np.complex64)
    srv = np.ascontiguousarray(srvChannel, dtype=np.complex64)   # complex128 srv is narrowed
    st = engine.staging()
    d_ref = st.get("caf_ref", 8 * n_in)
    d_srv = st.get("caf_srv", 8 * n_in)
    d_out = st.get("caf_out", 8 * int(freqBins) * (int(rangeBins) + 1))
    d_ref.upload(ref)
    d_srv.upload(srv)
    d_win = None
    if window is not None:
        d_win = st.get("caf_win", 4 * n)
        d_win.upload(window)
    plan.execute(d_ref, d_srv, d_out, 1, n, n_in, d_win)
    return d_out.download((int(freqBins), int(rangeBins) + 1, 1), np.complex64)


def direct_xambg(refChannel, srvChannel, rangeBins, freqBins, sampleRate):
    """Direct (time-domain) cross-ambiguity function, range_doppler_processing.py:93-124: for every Doppler bin i the
    reference is shifted by (i - freqBins/2) / CPI Hz (frequency_shift, float32 phase ramp) and correlated with the
    surveillance channel at lags 0..rangeBins (xcorr).  The reference never calls it on its processing path (SURVEY 8a:
    an independent peak check); it is here so that swapping the import line never raises.  A thin composition of
    prc_frequency_shift + prc_xcorr on device buffers: both channels go up once, the surface comes down once.

    Returns (freqBins, rangeBins+1, 1) complex64; row i is Doppler (i - freqBins/2)/CPI (NOT mirrored like
    fast_xambg's rows), column k is delay rangeBins - k."""
    if tuple(refChannel.shape) != tuple(srvChannel.shape):                  # :107-108
        raise ValueError("Input vectors must have the same length")
    ref = np.ascontiguousarray(refChannel.cpu().numpy() if _lib.is_device_tensor(refChannel) else refChannel, dtype=np.complex64)
    srv = np.ascontiguousarray(srvChannel.cpu().numpy() if _lib.is_device_tensor(srvChannel) else srvChannel, dtype=np.complex64)
    if ref.ndim != 1:
        raise ValueError("direct_xambg takes one-dimensional channels")
    n, R, F = ref.shape[0], int(rangeBins), int(freqBins)
    if R < 0:
        raise ValueError("index can't contain negative values")              # np.pad inside xcorr
    cpi = n / sampleRate                                                     # :111
    st = engine.staging()
    d_ref = st.get("dx_ref", 8 * n)
    d_srv = st.get("dx_srv", 8 * n)
    d_shift = st.get("dx_shift", 8 * n)
    d_out = st.get("dx_out", 8 * max(F, 1) * (R + 1))
    d_ref.upload(ref)
    d_srv.upload(srv)
    L = _lib.lib()
    for i in range(F):
        df = (i - 0.5 * F) / cpi                                             # :119
        _lib.check(L.prc_frequency_shift(d_ref.ptr, d_shift.ptr, n, float(df), float(sampleRate), 0.0, None))
        _lib.check(L.prc_xcorr(d_shift.ptr, d_srv.ptr, n, R, 0, d_out.ptr + 8 * i * (R + 1), None))   # :123
    return d_out.download((F, R + 1, 1), np.complex64)


def fast_xambg_multi(refChannels, srvChannel, rangeBins, freqBins, inputLen=None, window=None, shortFilt=True,
                     mode="auto"):
    """``[fast_xambg(r, srvChannel, ...) for r in refChannels]`` in one call: several illuminators against ONE
    surveillance channel (BASELINE config 5).  The reference calls fast_xambg once per (reference, surveillance) pair
    (range_doppler_processing.py:12-90); the pairs of one frame share the surveillance channel and the window, so the
    device can transform the surveillance pieces once per segment for all of them (prc_caf_execute_multi; ``mode``:
    "auto" = the library's measured choice, "turns" = one single-reference pass per illuminator, "shared" / "pairs" =
    surveillance spectra shared by all / by pairs of illuminators -- prc_caf_desc.multi).  Same argument
    meaning, errors and per-surface result as fast_xambg; returns a list of (freqBins, rangeBins+1, 1) complex64."""
    refs = list(refChannels)
    if not refs:
        return []
    if len(refs) > _lib.CAF_MAX_REFS:
        out = []
        for i in range(0, len(refs), _lib.CAF_MAX_REFS):
            out += fast_xambg_multi(refs[i:i + _lib.CAF_MAX_REFS], srvChannel, rangeBins, freqBins, inputLen, window,
                                    shortFilt, mode)
        return out
    for r in refs:
        if tuple(r.shape) != tuple(srvChannel.shape):                         # :46-49
            raise ValueError("Input vectors must have the same length")
    n_in = int(srvChannel.shape[0])
    n = n_in if inputLen is None else int(inputLen)
    if n_in > n:
        raise ValueError("index can't contain negative values")
    if isinstance(window, (tuple, str)):
        window = _named_window(window, n)
    elif window is not None and not _lib.is_device_tensor(window):
        window = np.ascontiguousarray(window, dtype=np.float32)
        if window.shape[0] != n:
            raise ValueError(f"operands could not be broadcast together with shapes ({n},) ({window.shape[0]},)")
    nref, F, R = len(refs), int(freqBins), int(rangeBins)
    if _lib.is_device_tensor(srvChannel):
        import torch
        dev = srvChannel.device
        with torch.cuda.device(dev):
            st = _lib.torch_stream_ptr(dev)
            plan = caf_plan_for(n, R, F, shortFilt, max_frames=nref, stream=st, multi=mode)
            srv = srvChannel.to(torch.complex64).contiguous()
            rr = [r.to(device=dev, dtype=torch.complex64).contiguous() for r in refs]
            win = None
            if window is not None:
                win = window if _lib.is_device_tensor(window) else torch.from_numpy(window)
                win = win.to(device=dev, dtype=torch.float32).contiguous()
            outs = [torch.empty((F, R + 1, 1), dtype=torch.complex64, device=dev) for _ in refs]
            plan.execute_multi(rr, srv, outs, 1, n, n_in, win, stream=st)
        return outs
    plan = caf_plan_for(n, R, F, shortFilt, max_frames=nref, multi=mode)
    st = engine.staging()
    d_srv = st.get("caf_srv", 8 * n_in)
    d_srv.upload(np.ascontiguousarray(srvChannel, dtype=np.complex64))
    d_refs = st.get("caf_refs", 8 * n_in * nref)
    d_outs = st.get("caf_outs", 8 * F * (R + 1) * nref)
    for i, r in enumerate(refs):
        d_refs.upload(np.ascontiguousarray(r, dtype=np.complex64), offset_bytes=8 * n_in * i)
    d_win = None
    if window is not None:
        d_win = st.get("caf_win", 4 * n)
        d_win.upload(window)
    plan.execute_multi([d_refs.ptr + 8 * n_in * i for i in range(nref)], d_srv,
                       [d_outs.ptr + 8 * F * (R + 1) * i for i in range(nref)], 1, n, n_in, d_win)
    return [d_outs.download((F, R + 1, 1), np.complex64, offset_bytes=8 * F * (R + 1) * i) for i in range(nref)]
