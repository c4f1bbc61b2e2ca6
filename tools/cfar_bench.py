"""CFAR_2D(18, 4) on PRconfig.yaml maps (1024 Doppler x 177 range cells, 256 frames per call): both kernel forms
(PRC_OPT_CFAR_METHOD 1 = every tap per cell, 0 = separable), HIP events; bytes = the map read once + the ratio written."""
import sys
import torch
sys.path.insert(0, ".")
from passiveradar_amd import _lib
from passiveradar_amd.target_detection import CFAR_2D

nf, H, W = 256, 1024, 177
X = torch.rand((nf, H, W), device="cuda") + 0.1
for method in (1, 0):
    _lib.set_option(_lib.OPT_CFAR_METHOD, method)
    for _ in range(2):
        out = CFAR_2D(X, 18, 4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = CFAR_2D(X, 18, 4)
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / 5
    print(f"CFAR_2D(18,4) method {method}: {nf} maps of {H}x{W} in {dt*1e3:.3f} ms -> {dt/nf*1e6:.2f} us per map, "
          f"{3 * 4.0 * nf * H * W / dt / 1e12:.3f} TB/s (mean pass + box pass reads, ratio write)")
_lib.set_option(_lib.OPT_CFAR_METHOD, 0)
# CFAR_2D(np.abs(X)) as range_doppler_plot.py:56-57 calls it: |X| in its own kernel (torch), then CFAR_2D; or CFAR_2D_abs, one kernel
from passiveradar_amd.target_detection import CFAR_2D_abs
Xc = torch.view_as_complex(torch.rand((nf, H, W, 2), device="cuda") + 0.1)
for name, fn in (("abs + CFAR_2D", lambda: CFAR_2D(Xc.abs(), 18, 4)), ("CFAR_2D_abs", lambda: CFAR_2D_abs(Xc, 18, 4))):
    for _ in range(2):
        out = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / 5
    print(f"{name:14s} on complex maps: {dt*1e3:.3f} ms -> {dt/nf*1e6:.2f} us per map")
