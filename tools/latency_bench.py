"""Single-call latency of the drop-in functions at config-2 sizes (host arrays in, host arrays out,
and device tensors in/out): what a dask worker calling once per chunk sees."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from passiveradar_amd import scene
from passiveradar_amd.clutter_removal import LS_Filter_Multiple
from passiveradar_amd.range_doppler_processing import fast_xambg
from scipy.signal import get_window

n, R, F, fs = 2400000, 256, 512, 2.4e6
ref, srv = scene.make_scene(n, fs, R, 1)
w = get_window(("kaiser", 5.0), n)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
print(f"fast_xambg  host->host  : {timeit(lambda: fast_xambg(ref, srv, R, F, n, w)):.2f} ms")
rt, st, wt = torch.from_numpy(ref).cuda(), torch.from_numpy(srv).cuda(), torch.from_numpy(w).float().cuda()
print(f"fast_xambg  device      : {timeit(lambda: fast_xambg(rt, st, R, F, n, wt)):.3f} ms")
C = n // 2
print(f"LS_Filter_Multiple host : {timeit(lambda: LS_Filter_Multiple(ref[:C], srv[:C], R, fs, [0, 1, -1, 2, -2])):.2f} ms")
