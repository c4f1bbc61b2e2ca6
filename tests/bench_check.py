"""Test helper: hold what bench.py computed at its benchmarked size against an independent pass.

bench.py --dump writes the first, second, middle and last map of its last timed step and every frame's sum.  The
independent pass here regenerates the benchmark's stream segment by segment (bench.stream_segment: seeds are keyed by
the global chunk index, so any part of a 54 GB stream can be made again on its own), runs it through another
``HipBackend`` -- windows of a few segments with one halo chunk each side, another sub-batch size, clutter canceller
and CAF back to back on one stream, one plan -- and returns every frame's sum, the picked maps and the raw chunks
under one frame for the CPU oracle.  main.py:169-194 is the semantics of both.
"""
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, dump, timeout=1500):
    """bench.py as the driver runs it (own process), two timed steps; returns (JSON line, dump)"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu", "--steps", "2", "--warmup", "1",
                        "--dump", dump] + list(args), capture_output=True, text=True, timeout=timeout, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    return line, np.load(dump)


def independent_pass(bench, torch, wl, nchunks, seed0, strong, pick, oracle_frame=None, window_segs=2, batch=64):
    """-> (sums [nchunks] complex64, {frame: map}, {chunk: (ref, srv, cleaned-on-the-device)} host chunks oracle_frame touches)"""
    from passiveradar_amd.stream import HipBackend
    fs, n, R, F, clutter, _ = bench.WORKLOADS[wl]
    C = n // 2
    dev = torch.device("cuda", 0)
    seg = bench.seg_chunks_for(C, strong)
    nseg = -(-nchunks // seg)
    be = HipBackend(n, R, F, fs, clutter=clutter, batch=batch, overlap=False, device=dev)
    gen = lambda k: bench.stream_segment(torch, k, nchunks, C, fs, R, seed0, dev, strong=strong)
    sums = np.zeros(nchunks, np.complex64)
    maps_at, chunks_at = {}, {}
    want_chunks = () if oracle_frame is None else [c for c in (oracle_frame - 1, oracle_frame, oracle_frame + 1)
                                                  if 0 <= c < nchunks]
    tail = None                # last chunk of the segment before this window
    ahead = None               # (k, segment) generated for a window's upper halo: the next window starts with it
    for k0 in range(0, nseg, window_segs):
        k1 = min(k0 + window_segs, nseg)
        c_lo, c_hi = k0 * seg, min(k1 * seg, nchunks)
        h_lo, h_hi = int(c_lo > 0), int(c_hi < nchunks)
        nlocal = h_lo + (c_hi - c_lo) + h_hi
        ref_pad = torch.zeros(nlocal * C + C, dtype=torch.complex64, device=dev)
        srv_pad = torch.zeros_like(ref_pad)
        pos = C // 2
        if h_lo:
            ref_pad[pos:pos + C], srv_pad[pos:pos + C] = tail
            pos += C
        for k in range(k0, k1):
            a, s = ahead[1] if (ahead is not None and ahead[0] == k) else gen(k)
            ref_pad[pos:pos + a.shape[0]], srv_pad[pos:pos + a.shape[0]] = a, s
            pos += a.shape[0]
            if k == k1 - 1:
                tail = (a[-C:].clone(), s[-C:].clone())
            del a, s
        ahead = None
        if h_hi:
            ahead = (k1, gen(k1))
            ref_pad[pos:pos + C], srv_pad[pos:pos + C] = ahead[1][0][:C], ahead[1][1][:C]
        maps = be.run(ref_pad, srv_pad, nlocal, h_lo * C, c_hi - c_lo)
        sums[c_lo:c_hi] = maps.sum(dim=(1, 2)).cpu().numpy()
        for f in pick:
            if c_lo <= f < c_hi:
                maps_at[int(f)] = maps[f - c_lo].cpu().numpy()
        for c in want_chunks:
            if c_lo - h_lo <= c < c_hi + h_hi and c not in chunks_at:
                o = C // 2 + (c - c_lo + h_lo) * C
                cl = be._clean_buf if be.clutter is not None else srv_pad      # the cleaned stream the frames were made of
                chunks_at[c] = (ref_pad[o:o + C].cpu().numpy(), srv_pad[o:o + C].cpu().numpy(), cl[o:o + C].cpu().numpy())
        del maps, ref_pad, srv_pad
    return sums, maps_at, chunks_at


def oracle_frame_map(bench, wl, nchunks, frame, chunks_at, lags=None, device_cleaned=False):
    """One frame end to end on the CPU (test infrastructure: oracle/): the clutter canceller of the workload on each of
    the hop chunks the frame touches, the overlapped CPI with zeros beyond the stream's ends (main.py:178-181), the Kaiser
    window, fast_xambg.  lags: only delays 0..lags (the LAST lags + 1 columns of the map), for sizes whose full CAF
    takes minutes on one core.  device_cleaned: the CAF stage alone -- the oracle's fast_xambg on the stream the DEVICE
    cleaned.  Returns (map, [the cleaned chunks that went into it])."""
    from scipy.signal import get_window
    from oracle import np_oracle as O
    fs, n, R, F, clutter, _ = bench.WORKLOADS[wl]
    C = n // 2
    zero = np.zeros(C, np.complex64)
    refs, cleans = [], []
    for c in (frame - 1, frame, frame + 1):
        if not (0 <= c < nchunks):
            refs.append(zero)
            cleans.append(zero)
            continue
        a, s, dev_clean = chunks_at[c]
        if device_cleaned:
            y = dev_clean
        elif clutter == "ls":
            y = O.LS_Filter_Multiple(a, s, R, fs, [0, 1, -1, 2, -2])
        elif clutter == "nlms":
            from oracle import c_oracle
            y = c_oracle.nlms(a, s, R, 0.02, 10)[0]
        else:
            y = s
        refs.append(a)
        cleans.append(np.asarray(y).astype(np.complex64))
    r3, c3 = np.concatenate(refs), np.concatenate(cleans)
    lo = C // 2                                               # frame i = stream[i C - C/2 : i C + 3C/2]
    w = get_window(("kaiser", 5.0), n)
    return O.fast_xambg(r3[lo:lo + n], c3[lo:lo + n], R if lags is None else lags, F, n, w)[:, :, 0], cleans
