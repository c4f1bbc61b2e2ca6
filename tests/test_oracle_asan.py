"""SURVEY section 5 (race / memory checking): the C twin of the oracle (oracle/c/oracle.c -- the NLMS recursion,
the direct CAF segment sums, the sosfilt recursion) built with AddressSanitizer + UndefinedBehaviorSanitizer
(make -C oracle ASAN=1) and run on edge-sized inputs in a python that has libasan preloaded.  Any out-of-bounds access,
use after free or undefined arithmetic in the checker aborts the child process."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import numpy as np
from oracle import c_oracle, np_oracle as O
from passiveradar_amd import scene
rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-30))
# NLMS: cold, warm start, the shortest stream that takes one step, a stream too short to take any
for n, L, warm in ((700, 24, False), (700, 1, True), (36, 24, False), (30, 24, False)):
    ref, srv = scene.make_scene(n, 1e4, 8, 11)
    t0 = (np.arange(L + 10) * 0.01).astype(np.complex64) if warm else None
    out, taps = c_oracle.nlms(ref, srv, L, 0.05, 10, t0)
    e, et = O.NLMS_filter(ref, srv, L, 0.05, 10, t0, True)
    assert rel(out, e) < 1e-4 and rel(taps, et) < 1e-4, (n, L, warm)
# direct CAF segments: odd q, N not a multiple of F, lags beyond q, with and without window
for n, R, F, win in ((4096, 7, 64, True), (6001, 6, 64, False), (4100, 40, 256, True), (64, 3, 2, False)):
    ref, srv = scene.make_scene(n, 8000.0, min(R, 8), 12)
    w = np.kaiser(n, 5.0) if win else None
    assert rel(c_oracle.fast_xambg(ref, srv, R, F, w), O.fast_xambg(ref, srv, R, F, n, w)) < 1e-5, (n, R, F)
# sosfilt recursion under find_channel_offset's decimator
from scipy import signal
sos = signal.cheby1(8, 0.05, 0.8 / 4, output="sos")
x = scene.white_reference(500, 13).astype(np.complex128)
zi = np.zeros((sos.shape[0], 2), np.complex128)
y, zf = c_oracle.sosfilt(sos, x, zi)
ye, ze = signal.sosfilt(sos, x, zi=zi)
assert rel(y, ye) < 1e-12 and rel(zf, ze) < 1e-10
y0, _ = c_oracle.sosfilt(sos, x[:1], zi)
assert y0.shape == (1,)
print("asan ok")
'''


def test_c_twin_under_address_and_ub_sanitizers():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "ASAN=1", "-s"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               ORACLE_LIB=os.path.join(REPO, "oracle", "liboracle_asan.so"), OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert r.returncode == 0 and "asan ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
