"""Lifetime and tooling: plans do not leak device memory when created, run and destroyed from
concurrent host threads (the reference's callers are dask worker threads, main.py:169-194), and the library's entry
points show up as roctx ranges when asked to."""
import gc
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gpu_ready")]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plans_created_and_destroyed_from_eight_threads_leak_nothing():
    """200 rounds of (CAF plan with either Doppler stage, LS plan of every kernel family, front-end plan): created,
    executed once, destroyed -- 25 rounds on each of 8 threads.  hipMemGetInfo's free memory is back within 1 MB."""
    from passiveradar_amd import _lib, engine
    n, R, F, C = 16384, 24, 64, 65536

    def one_round(i, bufs):
        ref, srv, out, raw, fe_out = bufs
        # 256 Doppler bins: the column kernel (slow-time buffer); 48: the rocFFT path (two buffers, a rocFFT plan + work area)
        Fi = (256, 48)[i % 2]
        caf = engine.CafPlan(n, R, Fi, 2, (_lib.CAF_AUTO, _lib.CAF_DIRECT, _lib.CAF_FFT)[i % 3])
        caf.execute(ref, srv, out, 2, n // 2, n, None, None)
        caf.close()
        # method 1 time-domain, 2 FFT, 3 FFT + spectrum cache, 4 the cached chain on 4096-point transforms
        L = (24, 24, 24, 300)[i % 4]
        ls = engine.LsPlan(C, L, 10, False, 3, (1, 2, 3, 4)[i % 4])
        ls.execute(ref, srv, out, 3, C, C, 2.4e5, (0, 1, -1) if i % 4 >= 2 else (0,), 0.0, None, None)
        ls.close()
        fe = engine.FrontendPlan(30000, "int8", 13, 119, 2)
        fe.execute(raw, fe_out, 2, 60000, fe.n_out, 1.0e5, 2.4e6, np.array([0.0, 0.4]), True, None)
        fe.close()

    def buffers():
        b = [_lib.DeviceBuffer(8 * 3 * C) for _ in range(3)] + [_lib.DeviceBuffer(2 * 60000), _lib.DeviceBuffer(8 * 2 * 4000)]
        for x in b:
            x.zero()
        return b

    warm = buffers()
    for i in range(12):                       # every kernel's code object, rocFFT's caches, per-thread scratch: loaded once
        one_round(i, warm)
    _lib.check(_lib.lib().prc_stream_sync(None))
    per_thread = [buffers() for _ in range(8)]
    gc.collect()
    free0, total = _lib.mem_info()
    errors = []

    def worker(t):
        try:
            for i in range(25):
                one_round(t + i, per_thread[t])
            _lib.check(_lib.lib().prc_stream_sync(None))
        except Exception as e:                # noqa: BLE001 -- reported below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    gc.collect()
    free1, _ = _lib.mem_info()
    print(f"free before {free0 / 2**20:.1f} MiB, after 200 plan rounds on 8 threads {free1 / 2**20:.1f} MiB of {total / 2**30:.0f} GiB")
    assert free0 - free1 < (1 << 20), (free0, free1)


def test_entry_points_open_roctx_ranges_when_asked_to(tmp_path):
    """PRC_OPT_MARKERS: the option binds the roctx library at run time and every prc_*_execute opens a range; off (the
    default) nothing is loaded.  Run in a fresh process so that 'not loaded' can be observed."""
    code = r'''
import numpy as np
from passiveradar_amd import _lib
from passiveradar_amd.range_doppler_processing import fast_xambg
from passiveradar_amd import scene
maps = lambda: open("/proc/self/maps").read()
ref, srv = scene.make_scene(8192, 1e4, 8, 3)
a = fast_xambg(ref, srv, 8, 32)
assert "roctx" not in maps(), "the marker library must not be loaded by default"
assert _lib.get_option(_lib.OPT_MARKERS) == 0
_lib.set_option(_lib.OPT_MARKERS, 1)
assert "roctx" in maps()
b = fast_xambg(ref, srv, 8, 32)
_lib.set_option(_lib.OPT_MARKERS, 0)
assert np.array_equal(a, b)
print("markers ok")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode == 0 and "markers ok" in r.stdout, r.stderr[-2000:]
