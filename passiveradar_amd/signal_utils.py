"""Drop-ins for the two ``passiveRadar/signal_utils.py`` helpers that sit on the hot path."""
from __future__ import annotations

import numpy as np

from . import _lib, engine
from ._lib import check, lib

__all__ = ["xcorr", "frequency_shift"]


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
        if np.size(phase_offset) != 1:
            raise NotImplementedError("array phase_offset belongs to the front end (out of scope)")
        phase_offset = float(np.asarray(phase_offset).reshape(()))
    x = np.ascontiguousarray(x, dtype=np.complex64)
    n = x.shape[0]
    st = engine.staging()
    dx = st.get("fs_x", 8 * n)
    dy = st.get("fs_y", 8 * n)
    dx.upload(x)
    check(lib().prc_frequency_shift(dx.ptr, dy.ptr, n, float(fc), float(Fs), float(phase_offset), None))
    return dy.download((n,), np.complex64)
