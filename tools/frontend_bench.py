"""Front-end throughput at the shipped PRconfig.yaml block size (4 799 250 raw int8 scalars -> 262 144 IF samples), HIP
events around the launches.  Variants on one box:
  per-output   PRC_OPT_FE_METHOD 1: one output per thread
  group bal=0  the group kernel (`up` outputs per thread), equal runs of tap rows at full width (PRC_OPT_FE_BALANCE 0)
  group bal=N  the same with banded segments dealt to the wavefronts by cost 2 w + N per trip (PRC_OPT_FE_BALANCE N)
  ... x2       both channels of a block in one workgroup (prc_frontend_execute2), per block-CHANNEL
    [FE_BALANCES=0,8,34] [PRCORE_LIB=build/libprcore_<variant>.so] python tools/frontend_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from passiveradar_amd import _lib  # noqa: E402
from passiveradar_amd.stream import HipBackend  # noqa: E402

icl, nblk = 4799250, 32
raw = torch.randint(-100, 100, (icl * nblk,), dtype=torch.int8, device="cuda")
raw2 = torch.randint(-100, 100, (icl * nblk,), dtype=torch.int8, device="cuda")
args = (icl, 100000, 2400000, 13, 119)
outs = {}


def timed(fn, reps=5):
    for _ in range(2):
        out = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, out


import os
balances = [int(v) for v in os.environ.get("FE_BALANCES", "0,34").split(",")]     # PRC_OPT_FE_BALANCE values (0 = equal runs)
variants = [("per-output", 1, 0, False)]
for b in balances:
    variants += [(f"group bal={b}", 2, b, False), (f"group bal={b} x2", 2, b, True)]
folds = [int(v) for v in os.environ.get("FE_FOLDS", "1").split(",")]        # PRC_OPT_FE_FOLD values (round 6: folded tap rows)
variants = [(f"{n} fold={f}" if m == 2 else n, m, b, t, f) for (n, m, b, t) in variants for f in (folds if m == 2 else [1])]
for name, method, balance, two, fold in variants:
    _lib.set_option(_lib.OPT_FE_FOLD, fold)
    _lib.set_option(_lib.OPT_FE_METHOD, method)
    _lib.set_option(_lib.OPT_FE_BALANCE, balance)
    be = HipBackend(524288, 175, 1024, 262184.87, batch=4, clutter=None)       # a fresh plan: the balance option is read at creation
    if two:
        dt, out = timed(lambda: be.front_end2(raw, raw2, *args))
        dt, out = dt / 2, out[0]
    else:
        dt, out = timed(lambda: be.front_end(raw, *args))
    outs[name] = out.clone()
    nout = out.shape[0] // nblk
    gflop = nblk * (nout * 4.0 * 2381 / 13 + (icl // 2) * 6.0) / 1e9   # real tap x complex sample multiply-adds on non-zero taps + one rotation product per input
    print(f"front end {name:24s}: {dt / nblk * 1e6:6.2f} us per block-channel ({icl * nblk / dt / 1e9:.1f} GB/s raw in, "
          f"{gflop / dt / 1e3:.1f} TFLOP/s of FIR + rotation products)", flush=True)
_lib.set_option(_lib.OPT_FE_METHOD, 0)
_lib.set_option(_lib.OPT_FE_BALANCE, 0)
ref = outs["per-output"]
for name, out in outs.items():
    print(f"max |{name} - per-output| / peak = {(out - ref).abs().max().item() / ref.abs().max().item():.2e}")
