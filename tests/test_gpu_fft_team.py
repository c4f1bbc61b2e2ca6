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
