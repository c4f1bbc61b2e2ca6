"""ctypes wrapper of oracle/liboracle.so (C twin of np_oracle) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("ORACLE_LIB", os.path.join(_HERE, "liboracle.so"))    # ORACLE_LIB: the sanitizer build (make ASAN=1)
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        _lib.orc_nlms.restype = C.c_int
        _lib.orc_nlms.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_caf_segments.restype = C.c_int
        _lib.orc_caf_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p]
        _lib.orc_sosfilt.restype = C.c_int
        _lib.orc_sosfilt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    return _lib


def sosfilt(sos, x, zi):
    """scipy.signal.sosfilt(sos, x, zi=zi) -> (y, zf), complex128, one-dimensional."""
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    y = np.array(x, dtype=np.complex128, order="C")
    z = np.array(zi, dtype=np.complex128, order="C")
    assert sos.shape[1] == 6 and z.shape == (sos.shape[0], 2)
    rc = lib().orc_sosfilt(sos.ctypes.data, sos.shape[0], y.ctypes.data, y.shape[0], z.ctypes.data)
    assert rc == 0
    return y, z


def nlms(ref, srv, L, mu, peek=10, initialTaps=None):
    """NLMS_filter (clutter_removal.py:189-249) -> (out, taps), complex64."""
    ref = np.ascontiguousarray(ref, dtype=np.complex64)
    srv = np.ascontiguousarray(srv, dtype=np.complex64)
    tin = None
    if initialTaps is not None:
        tin = np.ascontiguousarray(initialTaps, dtype=np.complex64)
        L = tin.shape[0] - peek
    T = L + peek
    out = np.empty(srv.shape[0], dtype=np.complex64)
    taps = np.empty(T, dtype=np.complex64)
    rc = lib().orc_nlms(ref.ctypes.data, srv.ctypes.data, srv.shape[0], L, peek, mu,
                        None if tin is None else tin.ctypes.data, out.ctypes.data, taps.ctypes.data)
    assert rc == 0
    return out, taps


def fast_xambg(ref, srv, rangeBins, freqBins, window=None):
    """fast_xambg with the boxcar decimator (range_doppler_processing.py:12-90), no padding branch."""
    from scipy.fft import fft
    ref = np.ascontiguousarray(ref, dtype=np.complex64)
    srv = np.ascontiguousarray(srv, dtype=np.complex64)
    n = ref.shape[0]
    y = np.empty((freqBins, rangeBins + 1), dtype=np.complex128)
    w = None if window is None else np.ascontiguousarray(window, dtype=np.float64)
    rc = lib().orc_caf_segments(ref.ctypes.data, srv.ctypes.data, n, rangeBins, freqBins,
                                None if w is None else w.ctypes.data, y.ctypes.data)
    assert rc == 0
    X = np.fft.fftshift(fft(y.astype(np.complex64), axis=0), axes=0)
    return X.reshape(freqBins, rangeBins + 1, 1)
