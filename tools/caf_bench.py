#!/usr/bin/env python3
"""CAF-only micro-benchmark for A/B library builds (PRCORE_LIB selects the .so):

    python tools/caf_bench.py --shape cfg5 --frames 8 --nref 4 [--reps 5] [--doppler 0|1|2] [--group-mb 96]

Times (HIP events on torch's current stream) the segment kernel, the Doppler stage, the grouped prc_caf_execute, and
-- with --nref > 1 -- prc_caf_execute_multi against nref single calls.  Random complex64 inputs resident in HBM."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {"cfg2": (2400000, 256, 512), "cfg3": (5000000, 1024, 1024), "cfg5": (1 << 23, 2048, 2048),
          "cfg1": (262144, 256, 256),
          # Doppler-stage probes: surfaces one tile wide, so that every row segment of a workgroup's tile is contiguous in
          # memory on both sides (what a tile-major slow-time buffer would give the loads)
          "dop2048x8": (131072, 7, 2048), "dop512x16": (32768, 15, 512)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="cfg5", choices=sorted(SHAPES))
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--nref", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--caf-method", type=int, default=0)
    ap.add_argument("--doppler", type=int, default=0)
    ap.add_argument("--group-mb", type=int, default=None)
    ap.add_argument("--multi", default="auto", choices=["auto", "turns", "shared", "pairs"])
    ap.add_argument("--tag", default="")
    ap.add_argument("--pair-frames", type=int, default=None, help="PRC_OPT_CAF_PAIR_FRAMES: 0 or 1")
    ap.add_argument("--xcd-contig", type=int, default=None, help="PRC_OPT_CAF_XCD_CONTIG: 0 or 1")
    ap.add_argument("--team8", type=int, default=None, help="PRC_OPT_CAF_TEAM8: 0 = teams of four wavefronts, 1 = of eight")
    ap.add_argument("--long-fir", action="store_true", help="shortFilt=False: firwin(10 q + 1, 1/q, flattop) instead of the boxcar")
    args = ap.parse_args()
    import torch
    from passiveradar_amd import _lib, engine
    _lib.require_gpu()
    if args.group_mb is not None:
        _lib.set_option(_lib.OPT_CAF_GROUP_MB, int(args.group_mb))
    if args.pair_frames is not None:
        _lib.set_option(_lib.OPT_CAF_PAIR_FRAMES, args.pair_frames)
    if args.xcd_contig is not None:
        _lib.set_option(_lib.OPT_CAF_XCD_CONTIG, args.xcd_contig)
    if args.team8 is not None:
        _lib.set_option(_lib.OPT_CAF_TEAM8, args.team8)
    n, R, F = SHAPES[args.shape]
    C = n // 2
    nf, nref = args.frames, args.nref
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    L = (nf + 1) * C
    mk = lambda: torch.view_as_complex(torch.randn((L, 2), generator=g, device=dev, dtype=torch.float32))
    refs = [mk() for _ in range(nref)]
    srv = mk()
    win = torch.from_numpy(np.kaiser(n, 5.0).astype(np.float32)).to(dev)
    taps = None
    if args.long_fir:
        from scipy.signal import firwin
        q = n // F
        taps = firwin(10 * q + 1, 1.0 / q, window="flattop").astype(np.float32)
    plan = engine.CafPlan(n, R, F, nf * nref, args.caf_method, args.doppler, taps=taps, multi=args.multi)
    outs = [torch.empty((nf, F, R + 1), dtype=torch.complex64, device=dev) for _ in range(nref)]
    s = _lib.torch_stream_ptr()
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            a, b = ev(), ev()
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    res = {"lib": os.environ.get("PRCORE_LIB", "default"), "tag": args.tag, "shape": args.shape, "frames": nf,
           "nref": nref, "long_fir": bool(args.long_fir), "xcd_contig": args.xcd_contig, "pair_frames": args.pair_frames, "team8": _lib.get_option(_lib.OPT_CAF_TEAM8), "method": plan.method, "doppler": plan.doppler, "multi": plan.multi}
    res["segments_ms"] = timeit(lambda: plan.execute_segments(refs[0], srv, nf, C, n, win, s))
    res["doppler_ms"] = timeit(lambda: plan.execute_doppler(outs[0], nf, s))
    res["execute_ms"] = timeit(lambda: plan.execute(refs[0], srv, outs[0], nf, C, n, win, s))
    bytes_surface = 20.0 * n + 8.0 * F * (R + 1)
    res["seg_us_per_surface"] = res["segments_ms"] * 1e3 / nf
    res["seg_GBps"] = bytes_surface * nf / (res["segments_ms"] * 1e-3) / 1e9
    res["exec_us_per_surface"] = res["execute_ms"] * 1e3 / nf
    if nref > 1:
        res["multi_ms"] = timeit(lambda: plan.execute_multi(refs, srv, outs, nf, C, n, win, s))
        res["multi_us_per_frame"] = res["multi_ms"] * 1e3 / nf
        shared = nref * 8.0 * n + 12.0 * n + nref * 8.0 * F * (R + 1)
        res["multi_GBps_shared_bytes"] = shared * nf / (res["multi_ms"] * 1e-3) / 1e9
        res["singles_ms"] = timeit(lambda: [plan.execute(r, srv, o, nf, C, n, win, s) for r, o in zip(refs, outs)])
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
