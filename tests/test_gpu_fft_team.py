"""The 4096-point team transform (passiveradar_amd/csrc/fft_team.h) on its own, through a test-only probe library
(tests/csrc), against numpy.fft and the layouts of tools/fft4096_model.py."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=["libfftprobe.so", "libfftprobe1.so"],
                ids=["two_buffers_one_barrier", "one_buffer_two_barriers"])
def probe(request, gpu_ready):
    LIB = os.path.join(HERE, "csrc", request.param)
    if not os.path.exists(LIB):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc")])
    from passiveradar_amd import _lib
    _lib.lib()                              # one HIP runtime per process (loads torch's copy first when present)
    h = ctypes.CDLL(LIB)
    h.fft_probe.restype = ctypes.c_int
    h.fft_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]

    def run(x, y, mode):
        x = np.ascontiguousarray(x, np.complex64)
        y = np.ascontiguousarray(y, np.complex64)
        out = np.empty_like(x)
        rc = h.fft_probe(x.ctypes.data, y.ctypes.data, out.ctypes.data, x.shape[0], mode)
        assert rc == 0, rc
        return out
    return run


def _rand(nb, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((nb, 4096)) + 1j * rng.standard_normal((nb, 4096))).astype(np.complex64)


def test_forward_matches_numpy_in_the_permuted_layout(probe):
    x = _rand(600, 1)                       # more workgroups than the chip holds at once
    F = probe(x, x, 0).reshape(-1, 256, 16)
    X = np.fft.fft(x.astype(np.complex128), axis=1)
    t = np.arange(256)[:, None]
    r = np.arange(16)[None, :]
    ref = X[:, (t >> 4) + 16 * (t & 15) + 256 * r]
    e = np.abs(F - ref).max() / np.abs(ref).max()
    assert e < 2e-6, e


def test_inverse_of_forward_is_identity(probe):
    x = _rand(600, 2)
    y = probe(x, x, 1) / 4096.0
    assert np.abs(y - x).max() < 2e-5


def test_back_to_back_transforms_like_the_caf_kernel(probe):
    """(fwd, fwd, accumulate) x 3, inverse -- twice -- with no extra barrier: the schedule the kernels rely on"""
    u, v = _rand(600, 3), _rand(600, 4)
    got = probe(u, v, 2)
    U = np.fft.fft(u.astype(np.complex128), axis=1)
    V = np.fft.fft(v.astype(np.complex128), axis=1)
    exp = np.fft.ifft(3 * np.conj(U) * V, axis=1) * 4096.0
    e = np.abs(got - exp).max() / np.abs(exp).max()
    assert e < 5e-6, e


# ---- the eight-wavefront transform (fft_team8.h; layouts of tools/fft4096x8_model.py) ----------------------------------
@pytest.fixture(scope="module")
def probe8(gpu_ready):
    LIB = os.path.join(HERE, "csrc", "libfftprobe8.so")
    if not os.path.exists(LIB):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc")])
    from passiveradar_amd import _lib
    _lib.lib()
    h = ctypes.CDLL(LIB)
    h.fft_probe8.restype = ctypes.c_int
    h.fft_probe8.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]

    def run(x, y, mode):
        x = np.ascontiguousarray(x, np.complex64)
        y = np.ascontiguousarray(y, np.complex64)
        out = np.empty_like(x)
        rc = h.fft_probe8(x.ctypes.data, y.ctypes.data, out.ctypes.data, x.shape[0], mode)
        assert rc == 0, rc
        return out
    return run


def test_team8_forward_matches_numpy_in_its_frequency_layout(probe8):
    x = _rand(900, 11)                      # more workgroups than the chip holds at once
    F = probe8(x, x, 0).reshape(-1, 512, 8)
    X = np.fft.fft(x.astype(np.complex128), axis=1)
    t = np.arange(512)[:, None]
    r = np.arange(8)[None, :]
    w, l = t >> 6, t & 63
    ref = X[:, w + 8 * (l >> 3) + 64 * (l & 7) + 512 * r]          # wave k1, lane 8 k2 + k3, register k4
    e = np.abs(F - ref).max() / np.abs(ref).max()
    assert e < 2e-6, e
    # a piece whose upper half is zero through the short first pass (NZ = 4) gives the same spectrum
    z = x.copy()
    z[:, 2048:] = 0
    a, b = probe8(z, z, 0), probe8(z, z, 3)
    assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max()


def test_team8_inverse_of_forward_is_identity(probe8):
    x = _rand(900, 12)
    y = probe8(x, x, 1) / 4096.0
    assert np.abs(y - x).max() < 2e-5


def test_team8_back_to_back_transforms_like_the_caf_kernel(probe8):
    """(fwd, fwd, accumulate) x 3, inverse -- twice -- and inverse straight after inverse: the barrier schedule"""
    u, v = _rand(900, 13), _rand(900, 14)
    got = probe8(u, v, 2)
    U = np.fft.fft(u.astype(np.complex128), axis=1)
    V = np.fft.fft(v.astype(np.complex128), axis=1)
    exp = np.fft.ifft(3 * np.conj(U) * V, axis=1) * 4096.0
    e = np.abs(got - exp).max() / np.abs(exp).max()
    assert e < 5e-6, e
    got = probe8(u, v, 4) / 4096.0
    assert np.abs(got - (u + 2 * v)).max() < 6e-5
