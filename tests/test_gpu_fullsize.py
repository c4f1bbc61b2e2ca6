"""The BENCHMARKED configurations at their real sizes: bench.py's default 5632-frame step
(6.8e9 samples per stream, byte offsets beyond 2^35), the 1199-frame 600 s stream and config 3's 3072-hop NLMS launch
are held, frame for frame, against an independent pass over the regenerated stream and one frame end to end against
the CPU oracle.  Semantics: main.py:169-194 (LS_Filter_Multiple / NLMS_filter per hop chunk, fast_xambg per
50 %-overlapped CPI)."""
import os
import sys

import numpy as np
import pytest

from conftest import rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gpu_ready")]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(tmp_path, wl, bench_args, nframes, strong, oracle_lags=None, window_segs=2, batch=64, nlms=False):
    import torch
    sys.path.insert(0, REPO)
    import bench
    import bench_check
    torch.cuda.empty_cache()                      # the benchmark keeps 160-190 GB resident: give back what earlier tests cached
    line, d = bench_check.run_bench(bench_args, str(tmp_path / f"{wl}.npz"))
    assert line["config"]["frames_per_step_total"] == nframes and int(d["nframes"]) == nframes
    pick = [int(f) for f in d["frame_index"]]
    assert pick == sorted({0, 1, nframes // 2, nframes - 1})
    last = nframes - 1
    sums, maps_at, chunks_at = bench_check.independent_pass(bench, torch, wl, nframes, int(d["seed0"]), strong, pick,
                                                            oracle_frame=last, window_segs=window_segs, batch=batch)
    torch.cuda.empty_cache()
    got = d["ill0_frames"]
    errs = {f: rel_err(got[j], maps_at[f]) for j, f in enumerate(pick)}
    # not bit-equal by construction: a plan of 258 blocks and a plan of 66 cut a chunk into different runs of pieces per
    # team, so the float32 partial sums of the correlations group differently (1e-6 measured at 600 frames)
    assert max(errs.values()) < 5e-6, errs
    # every frame's sum: each of the maps came from the right chunks (the cells' 1e-6 differences add incoherently
    # while the sum itself mostly cancels -- 1.2e-5 of the largest sum measured at 600 frames)
    e_sum = rel_err(d["ill0_sums"], sums)
    assert e_sum < 1e-4, e_sum
    # the LAST frame (the highest byte offsets of the resident streams) end to end on the CPU
    peak = float(np.abs(got[pick.index(last)]).max())
    mine = got[pick.index(last)]
    if oracle_lags is not None:
        mine = mine[:, -(oracle_lags + 1):]
    exp, cleans = bench_check.oracle_frame_map(bench, wl, nframes, last, chunks_at, lags=oracle_lags)
    e_or = float(np.abs(mine - exp).max() / peak)
    msg = (f"{wl}: {nframes} frames, picked maps vs independent pass {errs}, sums {e_sum:.2e}, last frame vs oracle end to end "
           f"{e_or:.2e}; {line['value']:.0f} frames/s in the dumped run")
    if not nlms:
        # LS: the device sums its correlations in double where it matters and the oracle is a float64 evaluation
        print(msg)
        assert e_or < 1e-4, e_or
        return
    # NLMS is a float32 recursion in the reference, in the C twin and on the device alike.  Two summation orders leave
    # ~2e-5 of the stream's level as differences in the taps' jitter; the jitter modulates the cancelled clutter, so the
    # differences are coherent with the reference channel and spread over the adaptation bandwidth (mu fs / T, ~100 Doppler
    # rows here) of a map whose peak is a 1 % target.  So the end-to-end number is reported and held to 5e-4, and the two
    # stages are held to the bar separately: the cleaned stream against the C twin at the stream's own scale (as
    # test_nlms_full_cfg3_hop_vs_c_oracle), the CAF of the DEVICE-cleaned stream against the oracle's fast_xambg.
    exp_caf, _ = bench_check.oracle_frame_map(bench, wl, nframes, last, chunks_at, lags=oracle_lags, device_cleaned=True)
    e_caf = float(np.abs(mine - exp_caf).max() / peak)
    e_clean = 0.0
    for k, c in enumerate((last - 1, last, last + 1)):
        if 0 <= c < nframes:
            e_clean = max(e_clean, float(np.abs(chunks_at[c][2] - cleans[k]).max() / np.abs(chunks_at[c][1]).max()))
    print(msg + f"; CAF stage alone {e_caf:.2e}, cleaned stream vs the C twin {e_clean:.2e} of the stream's peak")
    assert e_caf < 1e-4 and e_clean < 1e-4 and e_or < 5e-4, (e_or, e_caf, e_clean)


def test_default_bench_step_at_its_real_size(tmp_path):
    """bench.py with no arguments but --dump: 5632 frames per step, 22 sub-batches of 256 over three LS plans"""
    _check(tmp_path, "cfg2", [], 5632, False)


def test_600_s_stream_at_its_real_size(tmp_path):
    """bench.py --workload cfg4: all 1199 frames of the 600 s stream (the test of rounds 3-4 ran 21)"""
    _check(tmp_path, "cfg4", ["--workload", "cfg4"], 1199, True)


def test_config3_nlms_launch_at_its_real_size(tmp_path):
    """bench.py --workload cfg3: 3072 hop chunks of 2.5 M samples through ONE NLMS launch (three wavefronts per SIMD),
    then 3072 frames of 1024 x 1025; the oracle leg runs the C twin's NLMS over the two hops under the last frame and
    the CAF on delays 0..127 (the full 1025-lag CAF takes two minutes on one host core)"""
    _check(tmp_path, "cfg3", ["--workload", "cfg3"], 3072, False, oracle_lags=127, window_segs=5, batch=32, nlms=True)
