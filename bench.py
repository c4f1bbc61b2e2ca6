#!/usr/bin/env python3
"""bench.py -- CAF frames/s of the range-Doppler hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg4|cfg5|cfg3|cfg1|cfg2p2] [--frames B]

A *step* is one pass of the hot path (main.py:169-194 semantics) over one batch of B frames per GPU
of synthetic two-channel complex64 IQ already resident in HBM: LS_Filter_Multiple on the B hop chunks
(5 Doppler bins) + fast_xambg on the B 50 %-overlapped CPI frames, in sub-batches of 256.

Workloads (BASELINE.json configs):
  cfg2 (default) 2.4 MS/s, 1 s CPI (N = 2 400 000), 256 range x 512 Doppler, LS canceller.  N>1: every
                 rank processes its own B frames (weak scaling); the maps go to rank 0 with one RCCL
                 gather per 256-frame sub-batch, inside the timed region, overlapped with compute.
  cfg4           a 600 s stream of cfg-2 IQ = 1199 overlapped frames (main.py:116-120), frames sharded
                 contiguously over the ranks (stream.plan_shard: strong scaling, one halo chunk each side
                 re-filtered locally), ONE gather of the maps per pass.  A step = the whole stream.
  cfg5           20 MS/s, N = 2^23, 2048 x 2048, four illuminators against one surveillance channel, CAF
                 only; with N>1 the illuminators are sharded over the ranks (min(N,4)-way) and the frame
                 groups over what is left; the 33.6 MB surfaces stay on the GPU that made them.
  cfg3           10 MS/s, N = 5e6, 1024 x 1024, NLMS canceller (one wavefront per hop chunk).
  cfg1, cfg2p2   the reference's own CPU-sized case / the power-of-two variant of cfg2.
  prconfig       the reference's published workload (README.md:24, PRconfig.yaml as shipped) from raw int8 in host
                 memory to maps + CFAR back in host memory (and once more with the zarr store): its own JSON line.

`--gpus N` with N > 1 and no launcher around it starts the N ranks itself (torch.distributed.run on 127.0.0.1) and refuses
to run when the host has fewer GPUs; under a launcher it checks WORLD_SIZE against N.

One JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline     : the dominant kernel's rate on the HBM bytes it MOVED (2 x FETCH_SIZE + WRITE_SIZE of this run's own
                 rocprofv3 --pmc child passes; the committed per-unit figures or the analytic bytes when they cannot run --
                 `frac_basis` says which) over its average launch time (HIP events on the launch stream, kernel alone),
                 against the 8 TB/s HBM peak; SURVEY 8d's two-pass algorithmic accounting, which credits a fused kernel with
                 bytes it does not move, is reported as `frac_two_pass_accounting`; `frac_in_step` is the same kernel inside
                 the overlapped step.  (NLMS: flops against the 157.3 TFLOP/s fp32 vector peak -- VALU-issue-bound.)
  kernels      : per kernel family the solo launch time, the duration inside the overlapped step (kernel-trace child pass)
                 and the bytes moved; kernel_time_accounting relates their sums to ms_per_step
  cpu_baseline : the reference's CPU path (oracle restatement issuing the same NumPy/SciPy calls, SciPy-1.15 np.roots
                 artefact left out) MEASURED on this box's host cores: one core, and one frame per worker process on the
                 worker count that gives the strongest figure (usable cores = min(affinity, cgroup CPU quota): 16 on this
                 pool's boxes, which report 256; searched from min(usable, 32) workers, doubling / halving while the
                 throughput grows; every leg is listed; 256 workers measured five times WEAKER than 16:
                 profiles/r06_bench_default_allcore_legs.json)
  secondary    : BASELINE configs 3 and 5, a few steps each in child processes (default single-GPU line only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# the host driver of this pool only supports dmabuf IPC: without this RCCL's (and torch's) buffer sharing between the
# ranks of a node fails with "hipIpcGetMemHandle: invalid argument".  Already exported on the boxes; kept for any other env.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector peak
MIN_TIMED_SECONDS = 5.0        # auto-sized runs time at least this long

WORKLOADS = {
    # name: (Fs, N, R, F, clutter, default frames per GPU per step)
    "cfg2": (2.4e6, 2400000, 256, 512, "ls", 5632),
    "cfg2p2": (2.4e6, 2097152, 256, 512, "ls", 2048),
    "cfg1": (262184.87, 262144, 256, 256, "ls", 4096),
    # 3072 hop chunks per step: NLMS is one wavefront per chunk, and three wavefronts per SIMD (3 x 1024 SIMDs) issue
    # 1.36x the steps per second of one (61 GB per stream resident)
    "cfg3": (1.0e7, 5000000, 1024, 1024, "nlms", 3072),
    # config 4: 600 s of cfg-2 IQ, hop 0.5 s -> 1200 chunks, trimmed by one (main.py:116-120) -> 1199 frames
    "cfg4": (2.4e6, 2400000, 256, 512, "ls", 1199),
    # config 5: 20 MS/s, 2048 x 2048, four illuminators against one surveillance channel, CAF only
    # (16 frames = 64 surfaces per step: with 8 the segment kernel's last wave of workgroups is a visible tail)
    "cfg5": (2.0e7, 1 << 23, 2048, 2048, None, 16),
}
N_ILLUMINATORS = {"cfg5": 4}
SUB_BATCH = 256                # frames per CAF launch, per LS launch and per gather


def synth_segment(torch, nchunks, C, fs, R, seed, device, t0=0.0):
    """Device-side synthetic scene with the structure of passiveradar_amd.scene.make_scene
    (delays are circular shifts inside the segment)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = nchunks * C
    ref = torch.view_as_complex(torch.randn((n, 2), generator=g, device=device, dtype=torch.float32)
                                * np.float32(np.sqrt(0.5)))
    srv = torch.roll(ref, 2) + 0.3 * torch.roll(ref, 9) + 0.1 * torch.roll(ref, 40)
    t = torch.arange(n, device=device, dtype=torch.float64) / fs + t0
    for d, fd, a in ((60, 80.0, 0.01), (R // 2, -35.0, 0.003), (R - 5, 120.0, 0.003)):
        ph = (2 * np.pi * fd) * t
        rot = torch.complex(torch.cos(ph), torch.sin(ph)).to(torch.complex64)
        srv = srv + a * torch.roll(ref, d) * rot
    del t
    noise = torch.view_as_complex(torch.randn((n, 2), generator=g, device=device, dtype=torch.float32)
                                  * np.float32(np.sqrt(0.5)))
    srv = srv + 0.003 * noise
    return ref, srv.to(torch.complex64)


def seg_chunks_for(C, strong=False):
    """chunks per generated segment: 64 for the sharded 600 s stream (cfg4), else <= 128 and <= 2 GB per temporary"""
    return 64 if strong else max(1, min(128, (1 << 28) // C))


def stream_segment(torch, k, nchunks, C, fs, R, seed, device, strong=False):
    """Segment k of a synthetic stream of ``nchunks`` hop chunks, (ref, srv).  The seed is keyed by the segment's first
    GLOBAL chunk index, so a stream is the same however it is sharded and a test can regenerate any part of it.  Weak
    workloads cut the last segment short (its circular delays wrap inside what exists); the sharded stream (strong)
    always generates whole 64-chunk segments and keeps what lies inside the stream."""
    seg = seg_chunks_for(C, strong)
    c0 = k * seg
    m = seg if strong else min(seg, nchunks - c0)
    a, s = synth_segment(torch, m, C, fs, R, seed * 1000003 + c0, device, t0=c0 * C / fs)
    keep = min(m, nchunks - c0)
    return (a, s) if keep == m else (a[:keep * C], s[:keep * C])


def synth_padded(torch, nchunks, C, fs, R, seed, device, add_to=None):
    """[C/2 zeros | nchunks chunks | C/2 zeros] reference and surveillance streams, generated segment by
    segment (a 5632-chunk stream is 54 GB per channel; the generator's temporaries are kept to one
    segment).  add_to: an existing padded surveillance stream to accumulate into (further illuminators)."""
    ref_pad = torch.zeros(nchunks * C + C, dtype=torch.complex64, device=device)
    srv_pad = add_to if add_to is not None else torch.zeros(nchunks * C + C, dtype=torch.complex64, device=device)
    seg = seg_chunks_for(C)
    for k in range(-(-nchunks // seg)):
        a, s = stream_segment(torch, k, nchunks, C, fs, R, seed, device)
        lo = C // 2 + k * seg * C
        ref_pad[lo:lo + a.shape[0]] = a
        srv_pad[lo:lo + a.shape[0]] += s
        del a, s
    return ref_pad, srv_pad


# ---- CPU baseline: the reference's NumPy/SciPy path on this box's host cores ------------------------
def _cpu_frame_worker(args):
    """One whole frame of the workload (one LS hop / NLMS sample + one fast_xambg) in this process."""
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    workload, lags, seed_off = args
    from scipy.signal import get_window
    from oracle import np_oracle as O
    from passiveradar_amd import scene
    fs, n, R, F, clutter, _ = WORKLOADS[workload]
    ref, srv = scene.make_scene(n, fs, R, scene.scene_seed(2) + seed_off)
    w = get_window(("kaiser", 5.0), n)
    C = n // 2
    # lazy imports and first-call set-up of SciPy (resample_poly, correlate, solve_toeplitz) stay out of the clock
    O.fast_xambg_libcalls(ref[:4096], srv[:4096], 3, 16, w[:4096])
    if clutter == "ls":
        O.LS_Filter_Multiple_libcalls(ref[:4096], srv[:4096], 8, fs, [0, 1])
    t0 = time.perf_counter()
    O.fast_xambg_libcalls(ref, srv, lags, F, w)
    t_caf = (time.perf_counter() - t0) * (R + 1) / (lags + 1)
    t0 = time.perf_counter()
    if clutter == "ls":
        O.LS_Filter_Multiple_libcalls(ref[:C], srv[:C], R, fs, [0, 1, -1, 2, -2])
        t_cl = time.perf_counter() - t0
    elif clutter == "nlms":
        from oracle import c_oracle
        m = 20000
        c_oracle.nlms(ref[:m], srv[:m], R, 0.02, 10)
        t_cl = (time.perf_counter() - t0) * (C / (m - R - 10))
    else:
        t_cl = 0.0
    return t_caf, t_cl


def _cpu_quota():
    """CPUs' worth of time the cgroup grants this process (v2 cpu.max, v1 cfs quota), None when unlimited / unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def _usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = _cpu_quota()
    return n if q is None else max(1, min(n, int(np.ceil(q))))


def _host_memory_budget():
    """bytes this process tree may use: the smaller of MemAvailable and what the cgroup (v2 or v1) still allows"""
    avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v != "max" and int(v) < (1 << 60):
                room = int(v) - int(open(cur).read().strip())
                avail = room if avail is None else min(avail, room)
        except (OSError, ValueError):
            pass
    return avail


def cpu_baseline(workload, max_workers=None, asis=False, force_all=False):
    """(i) one core, bounded sample; (ii) one frame per worker process, concurrently, on the worker count that makes the
    STRONGEST baseline (the x50 claim is against the stronger multi-core number): searched from min(usable
    cores, 32), see below; never more than ``max_workers`` or than host memory holds; ``force_all`` adds the every-core leg."""
    import multiprocessing as mp
    fs, n, R, F, clutter, _ = WORKLOADS[workload]
    cores = _usable_cores()
    nill = N_ILLUMINATORS.get(workload, 1)
    # bounded one-core sample: 64 of the lag columns (whole LS hop), scaled to the full lag count
    lags1 = min(R, 63)
    W = max(1, cores if max_workers is None else min(cores, max_workers))
    # a worker holds its frame (ref, srv complex64, window float64) and fast_xambg's per-lag temporaries; measured peak
    # RSS at config 2: 296 MB (interpreter + NumPy/SciPy ~100 MB of it).  Bound: six N-sample complex128 arrays + 150 MB,
    # and the workers together may take half of what the host has free
    per_worker = 96.0 * n + 1.5e8
    budget = _host_memory_budget()
    mem_note = ""
    if budget is not None and W * per_worker > 0.5 * budget:
        W_mem = max(1, int(0.5 * budget // per_worker))
        mem_note = f"; {W} usable cores, capped to {W_mem} workers by host memory ({budget / 2**30:.0f} GiB free)"
        W = min(W, W_mem)
    # whole frames up to 64 workers; beyond, a quarter of the lag columns, scaled (the cost per column does not depend on the
    # column) so that a leg stays well under a minute even when memory bandwidth stretches every worker
    ctx = mp.get_context("spawn")
    # one thread per worker: the variables must be in the environment the children START with (their NumPy / OpenBLAS load
    # before any code of ours runs in them; round 3's 256-worker leg set them too late and measured 0.55 frames/s against
    # 2.2 on 32 workers)
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"
    legs = []
    # the one-core sample runs in a worker process of its own as well: in THIS process NumPy's BLAS is already up with
    # one thread per core (np.correlate's dot products would not be a one-core figure)
    with ctx.Pool(1) as pool1:
        t_caf1, t_cl1 = pool1.map(_cpu_frame_worker, [(workload, lags1, 0)])[0]
    one = 1.0 / (nill * t_caf1 + t_cl1)

    def leg(Wl):
        whole = workload in ("cfg2", "cfg2p2", "cfg1", "cfg4") and Wl <= 64
        lagsW = R if whole else min(R, 63)
        t0 = time.perf_counter()
        pool = ctx.Pool(Wl)
        try:
            per = pool.map_async(_cpu_frame_worker, [(workload, lagsW, i) for i in range(Wl)], chunksize=1).get(timeout=420)
            pool.close()
        except mp.TimeoutError:
            pool.terminate()
            legs.append({"workers": Wl, "error": "timed out after 420 s"})
            return None
        finally:
            pool.join()
        wall = time.perf_counter() - t0
        # every worker did one frame's worth of (scaled) work concurrently; throughput = workers / the slowest worker
        slow = max(nill * a_ + b_ for a_, b_ in per)
        rec = {"workers": Wl, "value": Wl / slow, "lag_columns_timed": lagsW + 1, "wall_seconds": wall,
               "slowest_worker_frame_seconds": slow, "fastest_worker_frame_seconds": min(nill * a_ + b_ for a_, b_ in per),
               "slowdown_vs_one_core": slow * one}
        legs.append(rec)
        return rec
    try:
        # the x50 claim is against the STRONGEST CPU number.  More workers are not always stronger: the MI355X
        # boxes of this pool report 256 logical cores and grant a cgroup CPU quota of 16 -- 256 workers made 0.45 frames/s,
        # 128 0.64, 64 1.43, 32 2.01, 16 2.45 (profiles/r06_bench_default_allcore_legs.json, before _usable_cores read the
        # quota: every worker slowed 97x .. 1.1x).  So: W = min(affinity, quota); start at min(W, 32), double while the
        # workers do not slow each other down and the throughput still grows, else halve while it grows; `value` is the
        # best leg and every leg is listed.  --cpu-workers all forces a leg on every usable core too.
        best = None
        tried = set()

        def take(Wl):
            nonlocal best
            tried.add(Wl)
            rec = leg(Wl)
            better = rec is not None and (best is None or rec["value"] > 1.03 * best["value"])
            if better:
                best = rec
            return better
        start = min(W, 32)
        take(start)
        grew, Wl = False, start
        # more workers only while the ones there are do not already slow each other down (a frame taking 1.5 x its one-core
        # time says the host is out of memory bandwidth or of real cores: doubling then only lengthens the run)
        while Wl < W and legs[-1].get("slowdown_vs_one_core", 9.9) < 1.5 and take(min(2 * Wl, W)):
            Wl, grew = min(2 * Wl, W), True
        if not grew:
            Wl = start
            while Wl > 2 and take(Wl // 2):
                Wl //= 2
        if force_all and W not in tried:
            take(W)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if best is None:
        best = {"workers": 1, "value": one, "lag_columns_timed": lags1 + 1, "wall_seconds": 0.0,
                "slowest_worker_frame_seconds": 1.0 / one, "fastest_worker_frame_seconds": 1.0 / one, "slowdown_vs_one_core": 1.0}
    note_cl = {"ls": "1 LS_Filter_Multiple hop (5 bins)", "nlms": "NLMS on 20k samples (C twin), scaled to one hop",
               None: "no canceller"}[clutter]
    return {
        "value": best["value"], "unit": "frames/s", "cores": best["workers"], "kind": "port",
        "sample": f"{best['workers']} worker processes x 1 frame each, concurrently ({best['lag_columns_timed']} of {R + 1} lag "
                  f"columns timed per fast_xambg, scaled; {note_cl}; 1 thread per process); NumPy/SciPy calls of the reference, "
                  f"np.roots artefact bypassed; the strongest of the legs listed under `legs` (all usable cores first){mem_note}",
        "one_core_value": one, "one_core_caf_seconds": t_caf1, "one_core_clutter_seconds": t_cl1,
        "one_core_sample": f"{lags1 + 1} of {R + 1} lag columns + {note_cl}",
        "host_cores": os.cpu_count() or 1, "usable_cores": cores, "workers": best["workers"], "legs": legs,
        "all_usable_cores_value": next((l_.get("value") for l_ in legs if l_["workers"] == W), None),
        "cpu_quota_cores": _cpu_quota(),
        "workers_wall_seconds": best["wall_seconds"], "slowest_worker_frame_seconds": best["slowest_worker_frame_seconds"],
    }


KERNEL_FAMILY = {"caf_segments": ("caf_fft_kernel", "caf_fft_team_kernel", "caf_fft_team_multi_kernel", "caf_direct_kernel"),
                 "caf_doppler": ("doppler_col_kernel", "shift_transpose_kernel", "transpose_jk_kj_kernel"),
                 "ls_correlate": ("ls_corr_cached_kernel", "ls_corr_cached_team_kernel"),
                 "ls_fir_subtract": ("ls_fused_cached_kernel", "ls_fused_cached_team_kernel"),
                 "ls_solve": ("ls_solve_gs_kernel", "ls_prepare_kernel"),
                 "nlms": ("nlms_kernel", "nlms_generic_kernel")}


def _kernel_family(name):
    k = name.replace("void ", "").split("(")[0].split("<")[0].strip()
    for fam, names in KERNEL_FAMILY.items():
        if k in names:
            return fam, k
    return None, k


def _child_cmd(wl, frames, steps, warmup, extra_args):
    return [sys.executable, os.path.join(REPO, "bench.py"), "--workload", wl, "--frames", str(frames), "--steps", str(steps),
            "--warmup", str(warmup), "--no-cpu", "--traffic", "none", "--secondary", "none", "--in-step", "none", "--no-kernel-timing"] + list(extra_args)


def measure_traffic(wl, frames, extra_args=(), timeout=150):
    """HBM bytes per launch of EVERY kernel family of the step, MEASURED by this run: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters only, no trace domain, as
    MI355X_MICROARCH.md prescribes -- on ``frames`` frames of the same workload (one launch of a kernel covers as many
    units as in the timed run), stages back to back on one stream.  Returns ({family: 2 x FETCH_SIZE + WRITE_SIZE per
    launch (the guide's gfx950 correction for wide coalesced reads), summed over the kernels of a family that follow one
    another within a stage (prepare + solve), averaged over the launches with the largest grid}, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="prc_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + \
                _child_cmd(wl, frames, 1, 1, ["--no-overlap"] + list(extra_args))
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} child failed (rc {r.returncode})"
            vals = {}
            for fcsv in files:
                for row in csv.DictReader(open(fcsv)):
                    fam, k = _kernel_family(row["Kernel_Name"])
                    if row["Counter_Name"] == ctr and fam:
                        vals.setdefault((fam, k), []).append((int(row["Grid_Size"]), float(row["Counter_Value"])))
            if not vals:
                return None, f"no library kernel in the --pmc {ctr} pass"
            got[ctr] = {}
            for (fam, k), v in vals.items():
                big = max(g for g, _ in v)
                sel = [x for g, x in v if g == big]
                got[ctr][(fam, k)] = sum(sel) / len(sel) * 1e3            # rocprofv3 reports KB
            # every library launch of the child belongs to one of its two steps (--no-kernel-timing)
            got[ctr + "_all"] = sum(x for v in vals.values() for _, x in v) * 1e3 / (2.0 * frames)
        except Exception as e:                                       # noqa: BLE001 -- a measurement aid must never take the line down
            return None, f"--pmc {ctr} pass: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out, parts = {}, {}
    out["_per_frame_total"] = 2.0 * got["FETCH_SIZE_all"] + got["WRITE_SIZE_all"]
    for key in set(got["FETCH_SIZE"]) | set(got["WRITE_SIZE"]):
        f_, w_ = got["FETCH_SIZE"].get(key, 0.0), got["WRITE_SIZE"].get(key, 0.0)
        out[key[0]] = out.get(key[0], 0.0) + 2.0 * f_ + w_
        p = parts.setdefault(key[0], [0.0, 0.0])
        p[0] += f_
        p[1] += w_
    return out, {"how": f"measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of bench.py --workload {wl} "
                        f"--frames {frames} --no-overlap, 2 x FETCH_SIZE + WRITE_SIZE per launch",
                 "fetch_write_MB_per_launch": {k: [round(v[0] / 1e6, 2), round(v[1] / 1e6, 2)] for k, v in parts.items()}}


def measure_in_step(wl, frames, extra_args=(), timeout=240):
    """The kernels' durations INSIDE the step (the solo HIP-event times describe each kernel alone on the chip;
    in the step several streams run together, every kernel takes longer and the sum of the solo times exceeds the step):
    one child run of this script under `rocprofv3 --kernel-trace --stats` on ``frames`` frames with the streams
    overlapped as in the timed region.  Returns ({family: {"avg_ms", "calls", "total_ms"}}, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    d = tempfile.mkdtemp(prefix="prc_kt_", dir="/tmp")
    try:
        steps, warmup = 3, 1
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "kt", "--"] + \
            _child_cmd(wl, frames, steps, warmup, extra_args)
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, f"rocprofv3 --kernel-trace child failed (rc {r.returncode})"
        child_line = None
        for ln in r.stdout.splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                child_line = json.loads(ln)
        fam_acc = {}
        for row in csv.DictReader(open(files[0])):
            fam, k = _kernel_family(row["Name"])
            if fam:
                a = fam_acc.setdefault(fam, {"calls": 0, "total_ms": 0.0, "kernels": {}})
                calls, tot = int(row["Calls"]), float(row["TotalDurationNs"]) / 1e6
                a["kernels"][k] = {"calls": calls, "avg_ms": tot / max(calls, 1)}
                a["total_ms"] += tot
                a["calls"] += calls
        for a in fam_acc.values():
            a["avg_ms"] = a["total_ms"] / max(a["calls"], 1)
        return fam_acc, {"how": f"rocprofv3 --kernel-trace --stats child pass of bench.py --workload {wl} --frames {frames} --steps {steps} "
                                f"--warmup {warmup} (streams overlapped as in the timed region; a traced run clocks 2-5 % lower)",
                         "child_steps": steps + warmup, "child_frames": frames,
                         "child_ms_per_step": child_line["ms_per_step"] if child_line else None}
    except Exception as e:                                           # noqa: BLE001
        return None, f"--kernel-trace pass: {e!r}"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def secondary_leg(wl, steps=None, warmup=1, timeout=420):
    """A short run of another BASELINE configuration in a child process (own HIP context, own buffers), so that the driver's
    one invocation also times configs 3 and 5.  Returns the child's line reduced to what matters."""
    import subprocess
    t0 = time.perf_counter()
    try:
        # steps: None = the workload's own sizing (>= 5 s of timed steps: 3-4 steps of config 3, ~1000 of config 5)
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--workload", wl, "--warmup", str(warmup),
               "--no-cpu", "--traffic", "file", "--secondary", "none", "--in-step", "none"] + ([] if steps is None else ["--steps", str(steps)])
        r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=timeout)
        line = None
        for ln in r.stdout.splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = json.loads(ln)
        if r.returncode != 0 or line is None:
            return {"error": f"child rc {r.returncode}", "stderr_tail": r.stderr[-400:]}
        keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data", "roofline",
                "hbm_moved_GBps", "hbm_frac_of_peak", "caf_only_frames_per_s_per_gpu", "caf_channels_per_launch")
        out = {k: line[k] for k in keep if k in line}
        out["workload"] = line["config"]["workload"]
        out["kernels"] = line.get("kernels")
        out["wall_seconds_incl_synthesis"] = time.perf_counter() - t0
        return out
    except Exception as e:                                           # noqa: BLE001 -- a side leg must never take the line down
        return {"error": repr(e)}


def fm_suppression(L=64):
    """SURVEY 8d's report-only figure: clutter suppression (dB) of the five-bin LS chain on an FM-like illuminator
    (badly conditioned Toeplitz system), device vs the oracle's per-bin complex128 Levinson (CPU, test infrastructure)."""
    from oracle import np_oracle as O
    from passiveradar_amd import scene
    from passiveradar_amd.clutter_removal import LS_Filter_Multiple
    n, fs = 262144, 262184.87
    ref, srv = scene.make_fm_scene(n, fs, L, 2024)
    bins = [0, 1, -1, 2, -2]
    core = slice(2 * L, n - 2 * L)
    p_in = np.mean(np.abs(srv[core]) ** 2)
    sup = lambda y: float(10 * np.log10(p_in / np.mean(np.abs(y[core]) ** 2)))
    return {"device": sup(LS_Filter_Multiple(ref, srv, L, fs, bins)), "oracle": sup(O.LS_Filter_Multiple(ref, srv, L, fs, bins)),
            "scene": f"FM-like reference (75 kHz deviation, 15 kHz audio), N={n}, {L + 10} taps, bins {bins}"}


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:                               # noqa: BLE001 -- cosmetic
        pass
    sys.stdout.flush()


def synth_raw_pinned(torch, nchunks, icl, fs_in, foff, device, seed=2026, per=16):
    """Raw interleaved int8 I,Q recordings of both channels in PINNED host memory, made on the device batch by batch
    (structure of passiveradar_amd.scene.make_raw_stream: band-limited noise offset_freq below the recording's centre,
    delayed copies + two moving echoes in the surveillance channel, independent int8 quantisation)."""
    n_blk = icl // 2
    raw = [torch.empty(nchunks * icl, dtype=torch.int8).pin_memory() for _ in range(2)]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    m_ = np.arange(-32, 33)
    fcut = 1.0e5 / fs_in
    lp = torch.tensor(2 * fcut * np.sinc(2 * fcut * m_) * np.hamming(65), dtype=torch.float32, device=device).reshape(1, 1, 65)
    for c0 in range(0, nchunks, per):
        m = min(per, nchunks - c0)
        n = m * n_blk
        w = torch.randn((2, 1, n + 64), generator=g, device=device, dtype=torch.float32)      # I and Q as two rows
        base = torch.nn.functional.conv1d(w, lp)                                     # 65-tap low-pass to +-100 kHz
        base = torch.complex(base[0, 0], base[1, 0])
        base = base / base.abs().pow(2).mean().sqrt()
        del w
        t = (torch.arange(n, device=device, dtype=torch.float64) + c0 * n_blk) / fs_in
        down = torch.polar(torch.ones_like(t), -2 * np.pi * foff * t).to(torch.complex64)
        srv = torch.roll(base, 18) + 0.3 * torch.roll(base, 83) + 0.1 * torch.roll(base, 366)
        for d, fd, amp in ((549, 80.0, 0.03), (1190, -35.0, 0.01)):
            srv = srv + amp * torch.roll(base, d) * torch.polar(torch.ones_like(t), 2 * np.pi * fd * t).to(torch.complex64)
        for k, x in enumerate((base, srv)):
            noise = torch.view_as_complex(torch.randn((n, 2), generator=g, device=device, dtype=torch.float32)) * 0.007
            q = torch.view_as_real((x + noise) * down).mul(24.0).round().clamp(-127, 127).to(torch.int8).reshape(-1)
            raw[k][c0 * icl:(c0 + m) * icl].copy_(q)
        del base, srv, down, t
    torch.cuda.synchronize()
    return raw


def prconfig_main(args):
    """--workload prconfig: the reference's PUBLISHED workload end to end (README.md:24: PRconfig.yaml, 1199 frames, "about
    20 minutes"): raw int8 recordings in pinned host memory -> H2D -> device front end (deinterleave, tune, 13:119) ->
    LS_Filter_Multiple x 5 bins -> fast_xambg per overlapped CPI -> CFAR_2D on |X| -> maps back to the host [-> zarr store
    + .npz axes], main.py:105-224.  A step = the whole recording.  Single GPU."""
    import torch
    from passiveradar_amd import _lib, output, stream as prstream
    from passiveradar_amd.target_detection import CFAR_2D_abs
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--workload prconfig runs on one GPU (a recording is one stream; N GPUs = N recordings)")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    _lib.require_gpu()
    cfg = json.load(open(os.path.join(REPO, "tests", "golden", "config_prconfig.json")))   # getConfiguration(PRconfig.yaml), made by the reference
    icl, fs_in, foff = cfg["input_chunk_length"], cfg["input_sample_rate"], cfg["offset_freq"]
    n, R, F, fs = cfg["cpi_samples"], cfg["num_range_cells"], cfg["num_doppler_cells"], cfg["IF_sample_rate"]
    nchunks = args.frames or (cfg["num_frames"] - 1)                     # main.py:116-120: N_chunks = ... - 1
    C = n // 2
    t0 = time.perf_counter()
    raw_ref, raw_srv = synth_raw_pinned(torch, nchunks, icl, fs_in, foff, device)
    t_synth = time.perf_counter() - t0
    be = prstream.HipBackend(n, R, F, fs, batch=min(256, nchunks), device=device, overlap=not args.no_overlap)
    sp = prstream.StreamProcessor(be)
    fe_args = (icl, foff, fs_in, cfg["resamp_up"], cfg["resamp_dn"])
    maps_h = torch.empty((nchunks, F, R + 1), dtype=torch.complex64).pin_memory()
    cfar_h = torch.empty((nchunks, F, R + 1), dtype=torch.float32).pin_memory()
    store_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "prconfig_bench_out")
    ev = lambda: torch.cuda.Event(enable_timing=True)
    marks = {}

    # resident IF streams (2.5 GB per channel at 1199 blocks), double-buffered raw staging, three streams: copies in,
    # compute, copies out.  Batch k + 1 crosses the link while batch k goes through front end -> LS -> CAF -> CFAR and
    # batch k - 1's maps travel back: the step is bound by the 11.5 GB of raw samples on the link.
    nbat = 32
    ref_pad = torch.zeros(nchunks * C + C, dtype=torch.complex64, device=device)
    srv_pad = torch.zeros_like(ref_pad)
    clean = torch.zeros_like(srv_pad)
    maps_d = torch.empty((nchunks, F, R + 1), dtype=torch.complex64, device=device)
    stage = [[torch.empty(nbat * icl, dtype=torch.int8, device=device) for _ in range(2)] for _ in range(2)]   # [slot][channel]
    s_in, s_out = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    sptr = lambda: _lib.torch_stream_ptr()

    def step(store):
        main = torch.cuda.current_stream()
        e0, e1 = ev(), ev()
        e0.record()
        freed = [None, None]
        done = 0                                              # frames computed so far
        writer = q = None
        if store:
            # the store rides along: a host thread writes the chunks of a batch of frames as soon as their copy has landed
            # (main.py:216-224 stores through dask chunk by chunk as well); the .npz axes go first
            import queue
            import threading
            t1 = time.perf_counter()
            c2 = dict(cfg, range_doppler_map_fname=os.path.join(store_dir, "xambg.zarr"), meta_fname=os.path.join(store_dir, "xambg.npz"))
            os.makedirs(store_dir, exist_ok=True)
            output.save_metadata(c2, nchunks)
            zw = output.ZarrFrameWriter(c2["range_doppler_map_fname"], F, R + 1, nchunks)
            maps_np = maps_h.numpy()
            q = queue.Queue()
            busy, failed = [0.0], []

            def drain_store():
                while True:
                    item = q.get()
                    if item is None:
                        return
                    if failed:
                        continue                                  # keep draining so that the producer never blocks
                    try:
                        landed_, lo_, hi_ = item
                        landed_.synchronize()
                        t2 = time.perf_counter()
                        zw.write(lo_, maps_np[lo_:hi_])
                        busy[0] += time.perf_counter() - t2
                    except Exception as e:                        # noqa: BLE001 -- re-raised on the main thread after the join
                        failed.append(e)
            writer = threading.Thread(target=drain_store)
            writer.start()
        for k, b0 in enumerate(range(0, nchunks, nbat)):
            m = min(nbat, nchunks - b0)
            slot = k & 1
            with torch.cuda.stream(s_in):
                if freed[slot] is not None:
                    s_in.wait_event(freed[slot])              # the front end that read this staging slot two batches ago
                stage[slot][0][:m * icl].copy_(raw_ref[b0 * icl:(b0 + m) * icl], non_blocking=True)
                stage[slot][1][:m * icl].copy_(raw_srv[b0 * icl:(b0 + m) * icl], non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(s_in)
            main.wait_event(landed)
            lo = C // 2 + b0 * C
            # main.py:133-149: both recordings tuned with the same phases -- one launch for the two channels
            be.front_end2(stage[slot][0][:m * icl], stage[slot][1][:m * icl], *fe_args, max_blocks=nbat, block0=b0,
                          out_ref=ref_pad[lo:lo + m * C], out_srv=srv_pad[lo:lo + m * C])
            freed[slot] = torch.cuda.Event()
            freed[slot].record(main)
            be._clean_range(ref_pad, srv_pad, clean, b0, m, sptr())       # LS_Filter_Multiple on the new blocks (main.py:169-176)
            # frame f needs cleaned blocks f - 1 .. f + 1 (zero boundary at both ends of the recording, main.py:178-181)
            ready = nchunks if b0 + m == nchunks else b0 + m - 1
            if ready > done:
                be.frames(ref_pad, clean, 0, nchunks, maps_d, done, ready)
                computed = torch.cuda.Event()
                computed.record(main)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(computed)
                    maps_h[done:ready].copy_(maps_d[done:ready], non_blocking=True)
                    if store:
                        home = torch.cuda.Event()
                        home.record(s_out)
                        q.put((home, done, ready))
                    for f0 in range(done, ready, 256):            # range_doppler_plot.py:56-57 per frame: CFAR_2D(|X|, 18, 4)
                        f1 = min(f0 + 256, ready)
                        cfar_h[f0:f1].copy_(CFAR_2D_abs(maps_d[f0:f1], 18, 4), non_blocking=True)
                done = ready
        main.wait_stream(s_out)
        e1.record()
        torch.cuda.synchronize()
        marks.update(step_gpu_ms=e0.elapsed_time(e1))
        if store:
            q.put(None)
            writer.join()
            if failed:
                raise failed[0]
            marks["store_s"] = busy[0]
            marks["store_wall_s"] = time.perf_counter() - t1

    def stages():
        """the same work stage by stage on one stream (nothing overlapped): where the time would go without the pipeline"""
        e = [ev() for _ in range(5)]
        e[0].record()
        a, s_ = be.front_end2(raw_ref, raw_srv, *fe_args)
        e[1].record()
        frames = sp.process(a, s_)
        e[2].record()
        for f0 in range(0, nchunks, 256):
            cfar_h[f0:f0 + 256].copy_(CFAR_2D_abs(frames[f0:f0 + 256], 18, 4), non_blocking=True)
        e[3].record()
        maps_h.copy_(frames, non_blocking=True)
        e[4].record()
        torch.cuda.synchronize()
        return {"front_end_incl_h2d_ms": e[0].elapsed_time(e[1]), "ls_caf_ms": e[1].elapsed_time(e[2]),
                "cfar_incl_d2h_ms": e[2].elapsed_time(e[3]), "maps_d2h_ms": e[3].elapsed_time(e[4])}

    def pipeline_check():
        """the pipelined step's maps against the single-pass path (StreamProcessor.process_raw's order of work) on the
        same recording: the LS launches group their blocks differently, nothing else differs"""
        piped = maps_h.clone()
        a = be.front_end(raw_ref, *fe_args)
        s_ = be.front_end(raw_srv, *fe_args)
        one = sp.process(a, s_).cpu()
        peak = float(one.abs().max())
        return float((piped - one).abs().max()) / peak

    # ---- host to host: the pipelined step (PCIe-inclusive by construction; a side figure, never `value`) ----
    for _ in range(max(args.warmup, 1)):
        step(False)
    steps = args.steps or 5
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    dt_h2h = time.perf_counter() - t0
    t0 = time.perf_counter()
    step(True)
    dt_store = time.perf_counter() - t0
    import shutil
    shutil.rmtree(store_dir, ignore_errors=True)
    step(False)
    pipe_err = pipeline_check()
    stage_ms = stages()
    raw_bytes = 2.0 * nchunks * icl

    # ---- the number of record: the same work with the recording RESIDENT in HBM when the clock starts ----
    raw_ref_d = raw_ref.to(device)
    raw_srv_d = raw_srv.to(device)
    cfar_d = torch.empty((nchunks, F, R + 1), dtype=torch.float32, device=device)
    ref_pad.zero_()
    srv_pad.zero_()
    lo = C // 2

    def step_resident():
        be.front_end2(raw_ref_d, raw_srv_d, *fe_args, max_blocks=nbat, out_ref=ref_pad[lo:lo + nchunks * C],
                      out_srv=srv_pad[lo:lo + nchunks * C])
        be.run(ref_pad, srv_pad, nchunks, 0, nchunks, maps_d)       # LS x5 + fast_xambg, sub-batches overlapped
        for f0 in range(0, nchunks, 256):
            f1 = min(f0 + 256, nchunks)
            cfar_d[f0:f1] = CFAR_2D_abs(maps_d[f0:f1], 18, 4)       # range_doppler_plot.py:56-57 in one kernel

    for _ in range(max(args.warmup, 1)):
        step_resident()
    torch.cuda.synchronize()
    res_err = float((maps_d.cpu() - maps_h).abs().max()) / float(maps_h.abs().max())   # against the pipelined step's maps
    rsteps = max(steps, int(np.ceil(2.0 / max(dt_h2h / steps * 0.35, 1e-3))))           # about two seconds of timed work
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rsteps):
        step_resident()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = nchunks * rsteps / dt
    e = [ev() for _ in range(4)]
    e[0].record()
    be.front_end2(raw_ref_d, raw_srv_d, *fe_args, max_blocks=nbat, out_ref=ref_pad[lo:lo + nchunks * C],
                  out_srv=srv_pad[lo:lo + nchunks * C])
    e[1].record()
    be.run(ref_pad, srv_pad, nchunks, 0, nchunks, maps_d)
    e[2].record()
    for f0 in range(0, nchunks, 256):
        cfar_d[f0:f0 + 256] = CFAR_2D_abs(maps_d[f0:f0 + 256], 18, 4)
    e[3].record()
    torch.cuda.synchronize()
    resident_ms = {"front_end_both_channels_ms": e[0].elapsed_time(e[1]), "ls_caf_ms": e[1].elapsed_time(e[2]),
                   "abs_cfar_ms": e[2].elapsed_time(e[3])}
    del raw_ref_d, raw_srv_d
    # kernel under the step's compute: the fused LS pass (HBM-bound), measured as in the default workload
    be.ls.set_profiling(True)
    a = be.front_end(raw_ref[:icl * min(nchunks, 64)], *fe_args)
    s_ = be.front_end(raw_srv[:icl * min(nchunks, 64)], *fe_args)
    ap_, sp_ = be.padded(a), be.padded(s_)
    clean = be.clean(ap_, sp_, min(nchunks, 64))
    samples = []
    for _ in range(4):
        be._clean_range(ap_, sp_, clean, 0, min(nchunks, 64), _lib.torch_stream_ptr())
        ms3, k3 = be.ls.get_profile()
        samples.append(ms3)
    be.ls.set_profiling(False)
    acc = np.median(np.array(samples[1:]), axis=0)
    nb_ls = min(nchunks, 64)
    fir_ms = acc[2] / k3[2]
    fir_bytes = nb_ls * (24.0 * C + 16.0 * C * (k3[2] - 1) / k3[2])
    v_h2h = nchunks * steps / dt_h2h
    # the front-end kernel (VALU-bound): one launch of nbat blocks, HIP events on the launch stream
    fe_in = raw_ref[:icl * min(nchunks, nbat)].to(device)
    fe_in2 = raw_srv[:icl * min(nchunks, nbat)].to(device)
    fe_out = torch.empty(min(nchunks, nbat) * C, dtype=torch.complex64, device=device)
    fe_out2 = torch.empty_like(fe_out)
    for _ in range(2):
        be.front_end2(fe_in, fe_in2, *fe_args, max_blocks=nbat, out_ref=fe_out, out_srv=fe_out2)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        be.front_end2(fe_in, fe_in2, *fe_args, max_blocks=nbat, out_ref=fe_out, out_srv=fe_out2)
    e1.record()
    torch.cuda.synchronize()
    fe_ms = e0.elapsed_time(e1) / 5 / 2            # per channel: the launch converts both channels of its blocks
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        be.front_end(fe_in, *fe_args, max_blocks=nbat, out=fe_out)
    e1.record()
    torch.cuda.synchronize()
    fe_ms_one_channel_launch = e0.elapsed_time(e1) / 5
    nb_fe = min(nchunks, nbat)
    ntaps = 20 * max(cfg["resamp_up"], cfg["resamp_dn"]) + 1                     # scipy.signal.resample_poly's firwin length
    fe_flops = nb_fe * (C * (ntaps / cfg["resamp_up"]) * 4.0 + (icl // 2) * 6.0)  # real tap x complex sample MACs + one complex rotation per input
    fe_step_ms = 2.0 * fe_ms * nchunks / nb_fe
    fir_step_ms = acc[2] * nchunks / nb_ls
    fe_roof = {"kernel": "frontend_group_kernel", "bound": "valu", "achieved": fe_flops / (fe_ms * 1e-3) / 1e12, "peak": VALU_PEAK_TFLOPS,
               "unit": "TFLOP/s", "frac": fe_flops / (fe_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, "traffic": None,
               "ms_per_launch": 2 * fe_ms, "blocks_per_launch": nb_fe, "channels_per_launch": 2, "ms_per_step": fe_step_ms,
               "ms_per_one_channel_launch": fe_ms_one_channel_launch,
               "note": "polyphase FIR multiply-adds that meet a non-zero tap + the rotation product, against the packed-fp32 vector peak; "
                       "the sin/cos evaluation and the multiply-adds on the zero corners of the tap rows are not counted"}
    fir_roof = {"kernel": "ls_fir_subtract", "bound": "hbm", "achieved": fir_bytes / (fir_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": fir_bytes / (fir_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms_per_step": fir_step_ms,
                "note": "T = 185: the 1024-point chain's fused FIR + correlation pass"}
    result = {
        "metric": "frames/sec, PRconfig.yaml as shipped (raw int8 recordings -> front end -> LS x5 -> CAF -> CFAR), recordings resident in HBM",
        "value": value, "unit": "frames/s", "n_gpus": 1, "steps": rsteps, "warmup": max(args.warmup, 1),
        "ms_per_step": dt / rsteps * 1e3, "higher_is_better": True, "scaling": "weak",
        # the published figure is end to end on a CPU (file IO and the store included), `value` is the compute
        # with the recordings resident -- not like for like, so no ratio is claimed here; the like-for-like one
        # (host to host with the store) is published_reference.host_to_host_with_store_over_published below
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"prconfig: PRconfig.yaml unmodified -- 2.4 MS/s int8 I/Q, {icl} raw scalars per block and channel, "
                               f"tune {foff} Hz, resample {cfg['resamp_up']}:{cfg['resamp_dn']} -> {C}-sample hops, LS x5 bins T={R + 10}, "
                               f"{F} Doppler x {R + 1} range cells, CFAR_2D(18, 4); {nchunks} frames per step (README.md:24: 1199)",
                   "frames_per_gpu_per_step": nchunks, "parallelism": "single GPU"},
        "published_reference": {"source": "README.md:24", "text": "about 20 minutes for this configuration (1199 frames, CPU, dask)",
                                "frames_per_s": 1199.0 / 1200.0,
                                "host_to_host_with_store_over_published": (nchunks / dt_store) / (1199.0 / 1200.0),
                                "resident_compute_only_over_published": value / (1199.0 / 1200.0)},
        "resident_stages_ms": resident_ms,
        "resident_vs_pipelined_maps_max_err_of_peak": res_err,
        "host_to_host": {
            "what": "the recording in pinned host memory every step -> maps + CFAR back in pinned host memory (PCIe-inclusive)",
            "frames_per_s": v_h2h, "ms_per_step": dt_h2h / steps * 1e3, "steps": steps,
            "pipeline": f"batches of {nbat} blocks: H2D of batch k+1 | front end, LS, CAF of batch k | CFAR + D2H of batch k-1, three streams",
            "pipelined_vs_single_pass_max_err_of_peak": pipe_err,
            "stages_ms_unpipelined": stage_ms,
            "with_store": {"frames_per_s": nchunks / dt_store, "seconds": dt_store, "store_busy_seconds": marks.get("store_s"),
                           "format": "zarr v2 directory store (F, R+1, nframes), chunks (F, R+1, 1) + .npz axes, main.py:200-224; "
                                     "written by a host thread batch by batch as the maps land"},
            "pcie": {"raw_bytes_per_step": raw_bytes, "h2d_GBps_if_alone": raw_bytes / (stage_ms["front_end_incl_h2d_ms"] * 1e-3) / 1e9,
                     "step_GBps_raw_in": raw_bytes / (dt_h2h / steps) / 1e9}},
        "roofline": fe_roof if fe_step_ms >= fir_step_ms else fir_roof,        # the kernel the resident step spends most time in
        "kernels": {"frontend_group_kernel": fe_roof, "ls_fir_subtract": fir_roof},
        "synth_seconds": t_synth,
    }
    if not args.no_cpu:
        # the reference path on one host core, bounded: two raw blocks per channel -> one overlapped frame
        from oracle import np_oracle as O
        from passiveradar_amd import scene
        rr, rs = scene.make_raw_stream(2, icl, fs_in, foff, scene.scene_seed(14))
        t1 = time.perf_counter()
        a_ = O.front_end(rr, icl, foff, fs_in, cfg["resamp_up"], cfg["resamp_dn"])
        s_ = O.front_end(rs, icl, foff, fs_in, cfg["resamp_up"], cfg["resamp_dn"])
        fr = O.process_stream(a_, s_, n, R, F, fs)
        O.CFAR_2D(np.abs(fr[:, :, 0]), 18, 4)
        t_cpu = time.perf_counter() - t1
        result["cpu_baseline"] = {"value": 2.0 / t_cpu, "unit": "frames/s", "cores": 1, "kind": "port",
                                  "sample": f"two raw blocks per channel through front end, LS x5, CAF (np.roots artefact bypassed) "
                                            f"and CFAR on one core: {t_cpu:.1f} s"}
    _flush_c_stdio()
    print(json.dumps(result), flush=True)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (RANK / WORLD_SIZE unset): start the N ranks here --
    re-exec under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 -- so that the line can never
    claim fewer GPUs than were asked for (it used to benchmark ONE GPU and print n_gpus 1).  Fails loudly when
    the host exposes fewer than N GPUs, or when a launcher's WORLD_SIZE disagrees with --gpus."""
    if args.gpus < 1:
        raise SystemExit(f"bench.py: --gpus {args.gpus}: at least one GPU")
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if launched:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        return
    if args.gpus == 1:
        return
    if args.workload == "prconfig":
        raise SystemExit("bench.py: --workload prconfig runs on one GPU (a recording is one stream; N GPUs = N recordings)")
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if args.share_gpu and have >= 1:
        have = args.gpus                                   # rehearsal: the ranks share what there is
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} asked for, this host exposes {have} GPU(s): refusing to run a "
                         f"{args.gpus}-GPU benchmark on fewer (no line printed)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 24 for the default workload = 5.9 s at 245 ms per step, else sized to >= 5 s)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=None, help="frames (= hop chunks) per GPU per step")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS) + ["prconfig"])
    ap.add_argument("--sub-batch", type=int, default=SUB_BATCH, help="frames per CAF launch and per gather")
    ap.add_argument("--caf-method", type=int, default=0, help="0 auto, 1 direct, 2 fft")
    ap.add_argument("--ls-method", type=int, default=0, help="0 auto, 1 time-domain, 2 FFT, 3 FFT + spectrum cache, 4 = 3 on 4096-point transforms")
    ap.add_argument("--nsub", type=int, default=1, help="LS sub-batches per CAF sub-batch when stages are pipelined")
    ap.add_argument("--ls-streams", type=int, default=3, help="LS chains in flight (alternate sub-batches on separate streams)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run LS and CAF back to back on one stream instead of pipelining sub-batches on two")
    ap.add_argument("--gather", default="auto", choices=["auto", "prc", "torch", "none"],
                    help="N>1: prc = prc_gather_frames (RCCL through the C ABI), torch = torch.distributed.gather, "
                         "auto = prc when every rank could create the communicator, else torch.  On ONE GPU an explicit "
                         "prc runs the same gather plumbing through a communicator of one rank (self-test)")
    ap.add_argument("--gather-parts", type=int, default=4,
                    help="cfg4, N>1: gather a shard's maps in this many rounds, each as soon as its frames are done "
                         "(1 = one gather after the whole shard)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--caf-lanes", type=int, default=1,
                    help="cfg5: 1 (default) = one plan, stage after stage; 2 = alternate half-batches of frames on two plans / "
                         "streams, so one piece's Doppler launch may run under the next piece's segment launch (measured in "
                         "round 5: 3690 against 3703 frames/s -- no idle share to recover)")
    ap.add_argument("--no-multi", action="store_true",
                    help="cfg5: one fast_xambg pass per illuminator instead of the shared-surveillance multi call")
    ap.add_argument("--cpu-workers", default="auto",
                    help="worker processes of the multi-core cpu_baseline legs: auto (default) = search for the strongest leg "
                         "from min(usable cores, 32) workers, doubling / halving while the throughput grows; all = the same "
                         "plus a leg on every usable core (minutes on a 256-core host); a number = at most that many")
    ap.add_argument("--cpu-asis", action="store_true",
                    help="also time the reference CAF as it ships (SciPy's per-lag np.roots call left in) on a few lag "
                         "columns, scaled: the transparency figure of SURVEY 8d")
    ap.add_argument("--no-clutter", action="store_true", help="CAF only (reported as a different metric)")
    ap.add_argument("--shard-of", default=None, metavar="RANK/WORLD",
                    help="cfg4 on ONE GPU: process only the shard rank RANK of WORLD would own (its frames + the two halo "
                         "chunks), no gather -- what one rank of an N-GPU run computes, measurable on a 1-GPU box")
    ap.add_argument("--traffic", default="auto", choices=["auto", "measure", "file", "none"],
                    help="roofline.traffic: measure = two rocprofv3 --pmc child passes of this script (FETCH_SIZE, WRITE_SIZE) on "
                         "one launch's worth of frames; file = the per-unit figures of profiles/traffic_latest.json (an earlier "
                         "run); auto = measure on one GPU when the CPU leg runs too (the default line; falls back to file, then to "
                         "the analytic bytes, when rocprofv3 is missing or a pass fails), else file; none")
    ap.add_argument("--in-step", default="auto", choices=["auto", "measure", "none"],
                    help="kernels{}.in_step_ms_per_launch: one rocprofv3 --kernel-trace --stats child pass of this script with the "
                         "streams overlapped as in the timed region; auto = on one GPU when the CPU leg runs too (the default line)")
    ap.add_argument("--secondary", default="auto", choices=["auto", "cfg3,cfg5", "cfg3", "cfg5", "none"],
                    help="short legs of BASELINE configs 3 and 5 in child processes, reported under `secondary`; auto = both, on the "
                         "default single-GPU cfg2 line only")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of an N > 1 run.  nccl (= RCCL, default) is the product; gloo is a REHEARSAL of "
                         "the multi-rank control flow where RCCL cannot run (two ranks sharing the one GPU of the test box, "
                         "--share-gpu): sharding, halo, per-sub-batch gather bookkeeping, part gathers, the rank-0 line -- with "
                         "real kernels, the maps staged through the host for the gather.  Its line says so and is not a number "
                         "of record")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rank r uses GPU r mod (number of GPUs): several ranks on one GPU (rehearsal runs only)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the per-kernel HIP-event section (the counter / trace child passes: every library launch they see "
                         "then belongs to a step)")
    ap.add_argument("--markers", action="store_true",
                    help="PRC_OPT_MARKERS: the library's entry points open roctx ranges (rocprofv3 --marker-trace)")
    ap.add_argument("--dump", default=None, metavar="NPZ",
                    help="single GPU: save the maps of the last timed step's first, second, middle and last frame "
                         "(+ every frame's sum) so a test can hold the benchmarked path against an independent pass")
    args = ap.parse_args()
    self_launch(args)
    if args.markers:
        from passiveradar_amd import _lib as _l
        _l.set_option(_l.OPT_MARKERS, 1)
    if args.workload == "prconfig":
        return prconfig_main(args)

    # BASELINE configs 3 and 5, a few steps each, in child processes BEFORE this process touches the GPU (each leg has the
    # whole 288 GB to itself; config 3 keeps 180 GB resident): the driver's one default invocation then carries driver-timed
    # numbers for them too
    secondary = None
    sec = args.secondary
    if sec == "auto":
        sec = "cfg3,cfg5" if (args.workload == "cfg2" and args.gpus == 1 and not args.no_cpu and not args.no_clutter
                              and args.frames is None and "RANK" not in os.environ) else "none"
    if sec != "none":
        secondary = {}
        for leg in sec.split(","):
            print(f"[bench] secondary leg {leg} ...", file=sys.stderr, flush=True)
            secondary[leg] = secondary_leg(leg)

    import torch
    import torch.distributed as dist
    from passiveradar_amd import _lib, stream as prstream

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                                    # self_launch() has already refused every way to get here
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) are running")
    if args.share_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rehearsal = world > 1 and (args.dist_backend != "nccl" or args.share_gpu)
    # small control tensors of the collectives: on the device for RCCL, on the host for gloo (which moves CUDA tensors only
    # for broadcast and all_reduce)
    ctrl = device if args.dist_backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib.require_gpu()

    wl = args.workload
    fs, n, R, F, clutter, B_default = WORKLOADS[wl]
    if args.no_clutter:
        clutter = None
    C = n // 2
    sub = max(1, args.sub_batch)
    nill = N_ILLUMINATORS.get(wl, 1)
    strong = wl == "cfg4"

    # ---- what this rank owns ----------------------------------------------------------------------
    ill_ways = min(world, nill) if nill > 1 else 1            # illuminator shards (cfg5)
    groups = world // ill_ways if nill > 1 else world         # frame groups
    my_ills = [i for i in range(nill) if i % ill_ways == rank % ill_ways]
    if strong:
        total = args.frames or B_default                      # frames in the whole stream
        if args.shard_of and world == 1:
            fr, fw = (int(v) for v in args.shard_of.split("/"))
            shard = prstream.plan_shard(total, fr, fw)
        else:
            shard = prstream.plan_shard(total, rank, world)
        nlocal, nframes, first = shard.nlocal_chunks, shard.nframes, shard.frame_offset(shard.frame_lo, C)
        B = nframes
        frames_per_step_total = total if not (args.shard_of and world == 1) else nframes
    else:
        B = args.frames or B_default
        if nill > 1:
            B = B * ill_ways                                  # a frame group works through ill_ways * B frames ...
        nlocal, nframes, first = B, B, 0
        frames_per_step_total = B * groups if nill > 1 else B * world   # ... each rank its own illuminators
    batch = min(sub, max(nframes, 1))
    be = prstream.HipBackend(n, R, F, fs, clutter=clutter, batch=batch, device=device,
                             caf_method=args.caf_method, overlap=not args.no_overlap,
                             ls_method=args.ls_method, nsub=args.nsub, ls_streams=args.ls_streams,
                             nref=len(my_ills) if nill > 1 else 1, caf_lanes=args.caf_lanes if nill > 1 else 1)
    multi = nill > 1 and len(my_ills) > 1 and not args.no_multi

    # ---- synthetic IQ resident in HBM --------------------------------------------------------------
    seed0 = 20260926 + (rank if not strong else 0)
    if strong:
        # every rank generates the chunks of ITS shard only (chunk_lo..chunk_hi); segment seeds are keyed by
        # the global chunk index so the stream is the same whatever the sharding
        ref_pad = torch.zeros(max(nlocal, 1) * C + C, dtype=torch.complex64, device=device)
        srv_pad = torch.zeros_like(ref_pad)
        seg = seg_chunks_for(C, True)
        for k in range(shard.chunk_lo // seg, -(-shard.chunk_hi // seg)):
            a, s = stream_segment(torch, k, total, C, fs, R, seed0, device, strong=True)
            g0 = k * seg
            lo, hi = max(g0, shard.chunk_lo), min(g0 + a.shape[0] // C, shard.chunk_hi)
            dst = C // 2 + (lo - shard.chunk_lo) * C
            ref_pad[dst:dst + (hi - lo) * C] = a[(lo - g0) * C:(hi - g0) * C]
            srv_pad[dst:dst + (hi - lo) * C] = s[(lo - g0) * C:(hi - g0) * C]
            del a, s
        refs = [ref_pad]
    elif nill > 1:
        # the surveillance channel carries every illuminator's scene; a rank keeps only the
        # reference channels of the illuminators it owns
        g = rank // ill_ways
        srv_pad, refs = None, []
        for i in range(nill):
            r_i, srv_pad = synth_padded(torch, B, C, fs, R, 777 + 13 * i + 1000 * g, device, add_to=srv_pad)
            if i in my_ills:
                refs.append(r_i)
            del r_i
    else:
        ref_pad, srv_pad = synth_padded(torch, B, C, fs, R, seed0, device)
        refs = [ref_pad]
    torch.cuda.synchronize()

    # ---- the one collective: gather of the maps to rank 0 ---------------------------------------------
    gather_mode = args.gather if (world > 1 and nill == 1) else "none"
    if world == 1 and nill == 1 and args.gather == "prc":
        # one GPU, asked for explicitly: the whole gather plumbing (cuts, events, receive buffers, prc_gather_frames)
        # through a communicator of ONE rank -- what tests/test_gpu_stream.py runs; not a number of record
        gather_mode = "prc"
    comm = None
    if rehearsal and gather_mode in ("auto", "prc"):
        if gather_mode == "prc":
            raise SystemExit("bench.py: --gather prc needs RCCL ranks on GPUs of their own (not --dist-backend gloo / --share-gpu)")
        gather_mode = "torch"                           # RCCL refuses two ranks on one GPU; gloo has no RCCL id to share
    if world == 1 and gather_mode == "prc":
        comm = prstream.FrameComm(0, 1, prstream.FrameComm.unique_id())
    elif gather_mode in ("auto", "prc"):
        ok = 1
        try:
            comm = prstream.FrameComm.from_torch_distributed()
        except Exception as e:                       # noqa: BLE001 -- any failure falls back (auto) or aborts (prc)
            ok, comm = 0, None
            if gather_mode == "prc":
                raise
            print(f"[rank {rank}] prc_comm_create failed ({e}); gathering through torch.distributed", file=sys.stderr)
        flag = torch.tensor([ok], device=ctrl, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            comm = None
        gather_mode = "prc" if comm is not None else "torch"
    s_comm = torch.cuda.Stream(device=device) if gather_mode != "none" else None
    gsize = nframes if strong else batch            # frames per gather from this rank
    if gather_mode != "none":
        if strong:
            gshard = shard
        else:
            gshard = prstream.Shard(rank, world, gsize * world, rank * gsize, (rank + 1) * gsize, 0, gsize)
        # persistent receive buffers, used in turn: four in the weak mode, whose gathers are issued sub-batch by sub-batch
        # WHILE the step is being enqueued (the host only ever waits for the gather issued four blocks earlier, so the GPU
        # always has two sub-batches of kernels queued ahead)
        nrecv = 2 if strong else 4
        recv = [torch.empty((gshard.nchunks, F, R + 1), dtype=torch.complex64, device=device) if rank == 0 else None
                for _ in range(nrecv)]
    else:
        nrecv = 2
    pending = [None] * nrecv                         # work handle of the gather that last used each buffer
    gcount = [0]

    last_gather = {}

    def gather(block, ev=None):
        k = gcount[0] % nrecv
        gcount[0] += 1
        last_gather.update(k=k, m=int(block.shape[0]), first=block.storage_offset() // (F * (R + 1)))
        if pending[k] is not None:
            pending[k][1].wait()
        if ev is not None:
            s_comm.wait_event(ev)                    # the launch that completed this block's frames
        else:
            s_comm.wait_stream(torch.cuda.current_stream())
        sh, out = gshard, recv[k]
        if not strong and block.shape[0] != gsize:   # ragged last sub-batch: every rank sends the same shorter block
            m = int(block.shape[0])
            sh = prstream.Shard(rank, world, m * world, rank * m, (rank + 1) * m, 0, m)
            out = recv[k][:m * world] if rank == 0 else None
        with torch.cuda.stream(s_comm):
            res, work = prstream.gather_frames(block, sh, async_op=True, out=out, comm=comm, stream=s_comm)
        block.record_stream(s_comm)
        pending[k] = (block, work)

    part_work, part_last = [], {}
    part_gather = (prstream.PartGather(gshard, args.gather_parts, comm=comm, stream=s_comm)
                   if (strong and gather_mode != "none" and args.gather_parts > 1) else None)

    def drain():
        for k in range(nrecv):
            if pending[k] is not None:
                pending[k][1].wait()
                pending[k] = None
        while part_work:
            part_work.pop(0).wait()

    # map buffers: two sets used alternately when the maps are gathered, so that a step never overwrites frames a
    # gather of the previous step may still be reading
    nsets = 2 if gather_mode != "none" else 1
    outs = [[torch.empty((max(nframes, 1), F, R + 1), dtype=torch.complex64, device=device) for _ in refs]
            for _ in range(nsets)]
    gathered = [None] * nsets                        # event: the gathers that read set p have been enqueued and done
    stepno = [0]

    def step():
        p = stepno[0] % nsets
        stepno[0] += 1
        if gathered[p] is not None:
            torch.cuda.current_stream().wait_event(gathered[p])
        if strong and gather_mode != "none" and args.gather_parts > 1:
            # the shard's maps leave in rounds while the rest of the shard is still being computed
            pg = part_gather
            nxt = [0]

            def on_frames(lo, hi, ev):
                while nxt[0] < pg.nparts and pg.part_range(nxt[0])[1] <= hi:
                    a_, b_ = pg.part_range(nxt[0])
                    s_comm.wait_event(ev)
                    if comm is None and part_last.get(nxt[0]) is not None:
                        part_last[nxt[0]].wait()          # torch path: the staging block of this round is free again
                    with torch.cuda.stream(s_comm):
                        w_ = pg.gather_part(nxt[0], outs[p][0][a_:b_], recv[p % 2] if rank == 0 else None, async_op=True)
                    part_work.append(w_)
                    part_last[nxt[0]] = w_
                    nxt[0] += 1
            cuts = [pg.part_range(q)[1] for q in range(pg.nparts)]
            if nframes:
                be.run(refs[0], srv_pad, nlocal, first, nframes, out=outs[p][0], cuts=cuts, on_frames=on_frames)
            on_frames(0, nframes, torch.cuda.current_stream().record_event())      # ranks with empty parts still take part
            ev_g = torch.cuda.Event()
            ev_g.record(s_comm)
            gathered[p] = ev_g
            return
        if gather_mode != "none" and not strong and not multi and nframes:
            # weak scaling: every sub-batch's maps leave for rank 0 as soon as the launch that completes them is enqueued,
            # on the communication stream behind that launch's event -- the transfers ride under the rest of the step's
            # kernels (before round 4 all 22 gathers of a step were issued after its last kernel and the host waited for
            # them before it could enqueue the next step: compute and gather in series on rank 0)
            def on_frames(lo, hi, ev):
                for f0 in range(lo, hi, gsize):
                    gather(outs[p][0][f0:min(f0 + gsize, hi)], ev)
            cuts = list(range(gsize, nframes, gsize))
            be.run(refs[0], srv_pad, nlocal, first, nframes, out=outs[p][0], cuts=cuts, on_frames=on_frames)
            ev_g = torch.cuda.Event()
            ev_g.record(s_comm)
            gathered[p] = ev_g
            return
        if multi and nframes:
            # every illuminator of this rank against the shared surveillance channel in ONE call per sub-batch
            # (prc_caf_execute_multi; how the channels share work is the plan's prc_caf_desc.multi: at config 5 AUTO = one
            # launch per stage for all channels, their workgroups co-located per XCD so that the surveillance windows are
            # fetched once)
            # join=False: the lanes are not joined to the main stream after every step -- the timed region's own fence
            # (torch.cuda.synchronize) is what ends the K steps, so step k+1's first segment launch may start under step
            # k's last Doppler launch, as in any resident pipeline
            be.frames_multi(refs, srv_pad, first, nframes, outs[p], join=False)
        else:
            for r_i, out in zip(refs, outs[p]):      # one fast_xambg pass per illuminator
                if nframes:
                    be.run(r_i, srv_pad, nlocal, first, nframes, out=out)
        if gather_mode != "none":
            if strong:
                gather(outs[p][0][:nframes])
            else:
                for f0 in range(0, nframes, gsize):
                    gather(outs[p][0][f0:f0 + gsize])
            ev_g = torch.cuda.Event()
            ev_g.record(s_comm)
            gathered[p] = ev_g

    def fence():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        step()
    fence()
    steps = args.steps
    if steps is None:
        if wl == "cfg2":
            steps = 24
        else:                                        # size the timed region to >= MIN_TIMED_SECONDS
            t0 = time.perf_counter()
            step()
            fence()
            probe = torch.tensor([time.perf_counter() - t0], device=ctrl, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(probe, op=dist.ReduceOp.MAX)
            steps = int(min(max(np.ceil(MIN_TIMED_SECONDS / max(float(probe.item()), 1e-6)), 3), 2000))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    rank_seconds = [dt]
    if world > 1:
        mine = torch.tensor([dt], device=ctrl, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                 # every rank's own clock around the same K steps
        rank_seconds = [float(t.item()) for t in every]
        dt = max(rank_seconds)
    value = frames_per_step_total * steps / dt      # a multi-illuminator frame = all its CAF surfaces
    # the communicator as RCCL sees it, and what every rank sent through it
    rccl_nranks, rccl_seen, gathered_bytes = None, None, None
    if world > 1:
        try:
            mine_n, mine_r = comm.count() if comm is not None else (0, -1)
        except Exception as e:                       # noqa: BLE001 -- reporting must never take the line down
            print(f"[rank {rank}] prc_comm_count failed: {e}", file=sys.stderr)
            mine_n, mine_r = -1, -1
        sent = float(nframes) * F * (R + 1) * 8.0 if gather_mode != "none" else 0.0
        info = torch.tensor([float(mine_n), float(mine_r), sent], device=ctrl, dtype=torch.float64)
        every_info = [torch.zeros_like(info) for _ in range(world)]
        dist.all_gather(every_info, info)
        rccl_seen = [[int(t[0].item()), int(t[1].item())] for t in every_info]
        rccl_nranks = rccl_seen[0][0] if comm is not None else None
        gathered_bytes = [float(t[2].item()) for t in every_info]
    if args.dump and world == 1 and nframes:
        last = outs[(stepno[0] - 1) % nsets]
        pick = sorted({0, min(1, nframes - 1), nframes // 2, nframes - 1})
        extra = {}
        if gather_mode != "none" and last_gather:
            # what the last gather of the step delivered (one GPU: this rank's block), and which frames it was
            drain()
            extra = {"gathered_block": recv[last_gather["k"]][:last_gather["m"]].cpu().numpy(),
                     "gathered_first_frame": last_gather["first"], "gathers_per_step": gcount[0] // max(stepno[0], 1)}
        np.savez(args.dump, frame_index=np.array(pick), seed0=seed0, nframes=nframes, **extra,
                 **{f"ill{i}_frames": o[pick].cpu().numpy() for i, o in enumerate(last)},
                 **{f"ill{i}_sums": o[:nframes].sum(dim=(1, 2)).cpu().numpy() for i, o in enumerate(last)})
    if args.dump and world > 1 and rank == 0 and gather_mode != "none" and nill == 1:
        # N > 1: what ARRIVED on the root.  Strong (cfg4): the assembled stream's maps -- every frame's sum and four picked
        # maps, comparable with a one-rank --dump of the same stream (sharding + halo + gather end to end).  Weak: the last
        # gathered block [rank 0's frames | rank 1's | ...] of the step and rank 0's own frames of that sub-batch.
        drain()
        torch.cuda.synchronize()
        if strong:
            full = recv[((stepno[0] - 1) % nsets) % 2] if args.gather_parts > 1 else recv[last_gather["k"]]
            pick = sorted({0, min(1, total - 1), total // 2, total - 1})
            np.savez(args.dump, frame_index=np.array(pick), seed0=seed0, nframes=total, world=world,
                     ill0_frames=full[pick].cpu().numpy(), ill0_sums=full[:total].sum(dim=(1, 2)).cpu().numpy())
        elif last_gather:
            m, f0 = last_gather["m"], last_gather["first"]
            np.savez(args.dump, world=world, m=m, first=f0, gathered_block=recv[last_gather["k"]][:m * world].cpu().numpy(),
                     own_frames=outs[(stepno[0] - 1) % nsets][0][f0:f0 + m].cpu().numpy())

    # ---- per-kernel timing with HIP events on the launch stream (rank 0) --------------------
    result = None
    if rank == 0 and args.no_kernel_timing:
        result = {"metric": f"CAF frames/sec ({wl}; child pass of a counter / trace collection)", "value": value, "unit": "frames/s",
                  "n_gpus": world, "steps": steps, "warmup": max(args.warmup, 1), "ms_per_step": dt / steps * 1e3,
                  "config": {"workload": wl, "frames_per_gpu_per_step": nframes}}
    elif rank == 0:
        reps = 5
        nb = min(batch, nframes)
        ev = lambda: torch.cuda.Event(enable_timing=True)
        kt = {}
        ref0 = refs[0]
        clean = be.clean(ref0, srv_pad, min(nlocal, nb + 1))
        out = outs[0][0]
        s = _lib.torch_stream_ptr()
        e0, e1, e2 = [], [], []
        for _ in range(reps + 1):                    # the first repetition is a warm-up (dropped); medians below
            a, b, c = ev(), ev(), ev()
            a.record()
            be.caf.execute_segments(ref0[first:], clean[first:], nb, C, n, be.window, s)
            b.record()
            be.caf.execute_doppler(out, nb, s)
            c.record()
            e0.append(a); e1.append(b); e2.append(c)
        torch.cuda.synchronize()
        e0, e1, e2 = e0[1:], e1[1:], e2[1:]
        kt["caf_segments"] = {"ms": float(np.median([a.elapsed_time(b) for a, b in zip(e0, e1)])),
                              "launches_per_step": -(-nframes // batch) * len(refs), "bound": "hbm",
                              "work": nb * (20.0 * n + 8.0 * F * (R + 1))}
        kt["caf_doppler"] = {"ms": float(np.median([b.elapsed_time(c) for b, c in zip(e1, e2)])),
                             "launches_per_step": -(-nframes // batch) * len(refs), "bound": "hbm",
                             "work": nb * 16.0 * F * (R + 1)}
        if clutter == "ls":
            be.ls.set_profiling(True)
            nb_ls = min(nlocal, be.sub)                         # blocks behind one LS launch
            samples = []
            for _ in range(reps + 1):                # first repetition: warm-up, dropped
                be._clean_range(ref0, srv_pad, clean, 0, nb_ls, s)
                ms3, k3 = be.ls.get_profile()
                samples.append(ms3)
            be.ls.set_profiling(False)
            acc = np.median(np.array(samples[1:]), axis=0)     # median per kernel kind: one disturbed repetition does not count
            T = R + 10
            fused = k3[0] == 1 and k3[2] > 1                   # cached-spectrum chain: corr(i+1) inside FIR(i)
            execs = -(-nlocal // nb_ls)                        # LS executes per step
            kt["ls_correlate"] = {"ms": acc[0] / k3[0], "launches_per_step": k3[0] * execs, "bound": "hbm",
                                  "work": nb_ls * 16.0 * C}
            kt["ls_solve"] = {"ms": acc[1] / k3[1], "launches_per_step": k3[1] * execs, "bound": "latency",
                              "work": nb_ls * (T * T * 16.0 + 64 * 2 * T * 8.0)}
            nb_bins = k3[2]
            fir_bytes = 24.0 * C + (16.0 * C * (nb_bins - 1) / nb_bins if fused else 0.0)
            kt["ls_fir_subtract"] = {"ms": acc[2] / k3[2], "launches_per_step": k3[2] * execs, "bound": "hbm",
                                     "work": nb_ls * fir_bytes}
            if fused:
                # what the fused kernel HAS to move per chunk-bin: the cached block spectrum (P complex64
                # per P - E new samples), the stream in and the stream out -- against SURVEY 8d's 24C + 16C (4/5) that the
                # roofline object is priced on (the fusion removes the separate correlation pass, so it moves fewer bytes)
                cache_per_sample = 32768.0 / (4096 - ((T - 1 + 15) & ~15)) if (T >= 250 and C >= 65536) else 8192.0 / (1025 - T)
                kt["ls_fir_subtract"]["compulsory_work"] = nb_ls * (16.0 + cache_per_sample) * C
        elif clutter == "nlms":
            a, b = ev(), ev()
            a.record()
            be.clean(ref0, srv_pad, nlocal)
            b.record()
            torch.cuda.synchronize()
            T = R + 10
            # per step and stream: e = d - w^H u, u^H u, w += mu u conj(e)/(u^H u): 2 T complex MACs = 16 T flops
            # (+ the energy slide); SURVEY 8d counts ~20 T
            kt["nlms"] = {"ms": a.elapsed_time(b), "launches_per_step": 1, "bound": "valu",
                          "work": nlocal * float(C - T) * 20.0 * T,
                          "streams": nlocal, "samples_per_s_per_stream": (C - T) / (a.elapsed_time(b) * 1e-3)}
        dom = max(kt, key=lambda k_: kt[k_]["ms"] * kt[k_]["launches_per_step"])
        d = kt[dom]
        # units (hop chunks for the LS kernels, frames for the CAF kernels) behind one launch of each family in this run
        units_of = {k_: (min(nlocal, be.sub) if k_.startswith("ls_") else (nlocal if k_ == "nlms" else nb)) for k_ in kt}
        # units of each family per frame of the step: an LS chunk per frame, x bins for the per-bin kernels
        per_frame_units = {k_: (kt[k_]["launches_per_step"] * units_of[k_] / max(nframes, 1)) for k_ in kt}
        default_line = world == 1 and not args.no_cpu and nill == 1
        # ---- HBM bytes MOVED, per launch of every family: this run's own counter passes, else the committed per-unit figures
        traffic_by, tsrc, moved_per_frame = {}, None, None
        mode = args.traffic
        if mode == "auto":
            mode = "measure" if default_line else "file"
        if mode == "measure":
            extra = ["--sub-batch", str(args.sub_batch)] + (["--no-clutter"] if args.no_clutter else [])
            got, note = measure_traffic(wl, units_of[dom], extra)
            if got is None:
                tsrc = {"how": f"not measured ({note}); "}
                mode = "file"
            else:
                moved_per_frame = got.pop("_per_frame_total")
                traffic_by, tsrc = got, note
        tpath = os.path.join(REPO, "profiles", "traffic_latest.json")
        if mode == "file" and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                per_unit = tj.get("cfg2" if wl == "cfg4" else wl, {})
                # the file holds bytes per chunk / frame; an LS launch covers one LS sub-batch, a CAF launch nb frames
                traffic_by = {k_: per_unit[k_] * units_of[k_] for k_ in kt if k_ in per_unit}
                if traffic_by:
                    tsrc = {"how": ((tsrc or {}).get("how", "") + tj.get("_source", "profiles/traffic_latest.json (rocprofv3 --pmc "
                                    "FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command; not measured by this run)"))}
                    if all(k_ in per_unit for k_ in kt if kt[k_]["bound"] != "valu"):
                        moved_per_frame = sum(per_unit[k_] * per_frame_units[k_] for k_ in kt if k_ in per_unit)
            except Exception:                                     # noqa: BLE001
                traffic_by = {}
        # ---- the kernels' durations inside the overlapped step (one kernel-trace child pass)
        in_step, in_step_note = None, None
        if args.in_step == "measure" or (args.in_step == "auto" and default_line):
            extra = ["--sub-batch", str(args.sub_batch)] + (["--no-clutter"] if args.no_clutter else [])
            in_step, in_step_note = measure_in_step(wl, min(nframes, 10 * batch), extra)
        if d["bound"] == "valu":
            achieved = d["work"] / (d["ms"] * 1e-3) / 1e12
            roof = {"kernel": dom, "bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / VALU_PEAK_TFLOPS, "traffic": None,
                    "frac_basis": "algorithmic flops (SURVEY 8d: ~20 T per step and stream) over the solo HIP-event launch time",
                    "note": "sample-recursive NLMS: one wavefront per stream, VALU-issue-bound; "
                            f"{d['samples_per_s_per_stream']:.3g} samples/s/stream x {d['streams']} streams"}
        else:
            # `frac` is the kernel's rate on the bytes it MOVED (PMC), not on SURVEY 8d's two-pass accounting, which
            # credits the fused kernel with a separate correlation pass it never makes (kept below, labelled)
            two_pass = d["work"] / (d["ms"] * 1e-3) / 1e9
            traffic = traffic_by.get(dom)
            if traffic:
                achieved, basis = traffic / (d["ms"] * 1e-3) / 1e9, "HBM bytes moved per launch (2 x FETCH_SIZE + WRITE_SIZE) over the solo HIP-event launch time"
            elif "compulsory_work" in d:
                achieved, basis = d["compulsory_work"] / (d["ms"] * 1e-3) / 1e9, "bytes the fused form has to move (analytic; no counters available) over the solo HIP-event launch time"
            else:
                achieved, basis = two_pass, "algorithmic bytes of SURVEY 8d (no counters available) over the solo HIP-event launch time"
            roof = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "frac_basis": basis,
                    "traffic_source": tsrc, "frac_of_copy_ceiling": achieved / 6290.0,
                    "solo_ms_per_launch": d["ms"], "units_per_launch": units_of[dom],
                    # SURVEY 8d's accounting (separate correlation and FIR passes): bytes a fused kernel is credited with but
                    # does not move -- a figure of merit of the FORMULATION, not a rate of the memory system
                    "achieved_two_pass_accounting": two_pass, "frac_two_pass_accounting": two_pass / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": d["work"]}
            if "compulsory_work" in d:
                roof["compulsory_bytes_per_launch"] = d["compulsory_work"]
                roof["frac_compulsory_bytes"] = d["compulsory_work"] / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            if in_step and dom in in_step and traffic:
                roof["in_step_ms_per_launch"] = in_step[dom]["avg_ms"]
                roof["frac_in_step"] = traffic / (in_step[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        caf_bytes = 20.0 * n + 8.0 * F * (R + 1)
        if nill > 1:                                            # srv and the window are shared by the illuminators
            per_frame_bytes = nill * 8.0 * n + 8.0 * n + 4.0 * n + nill * 8.0 * F * (R + 1)
        else:
            per_frame_bytes = caf_bytes + (200.0 * C if clutter == "ls" else (24.0 * C if clutter == "nlms" else 0.0))
        m, dp = be.caf.method, be.caf.doppler
        if strong:
            par = (f"{frames_per_step_total} frames of one stream sharded contiguously x{world} (strong), "
                   f"one gather of the maps per pass ({gather_mode})") if world > 1 else "single GPU, whole stream"
        elif nill > 1:
            par = (f"{nill} illuminators sharded {ill_ways}-way x {groups} frame group(s), surfaces stay on their GPU"
                   if world > 1 else f"single GPU, {nill} illuminators")
        else:
            par = f"frame-sharded x{world}, gather of maps to rank 0 per {gsize} frames ({gather_mode})" if world > 1 \
                else "single GPU"
        result = {
            "metric": "CAF frames/sec (1s CPI @ 2.4 MS/s, 256 range x 512 Doppler); HBM GB/s %peak"
                      if wl in ("cfg2", "cfg4") else f"CAF frames/sec ({wl})",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": steps,
            "warmup": max(args.warmup, 1), "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{wl}: Fs={fs:g} N={n} R={R} F={F} clutter="
                                   f"{clutter or 'none'}{' x5 Doppler bins, T=%d' % (R + 10) if clutter == 'ls' else ''}, "
                                   + (f"{frames_per_step_total}-frame stream (600 s), " if strong else
                                      f"{B} overlapped frames/GPU/step, ")
                                   + f"sub-batches of {batch} (hop N/2), Kaiser(5) window"
                                   + (f", {nill} illuminators / frame" if nill > 1 else ""),
                       "frames_per_gpu_per_step": nframes, "frames_per_step_total": frames_per_step_total,
                       "arithmetic": "complex64 streams, f32 FFT butterflies, f64 Levinson-Durbin / tap solves",
                       "caf_method": {1: "direct", 2: "fft"}.get(m, m),
                       "doppler_method": {1: "rocfft"}.get(dp, dp),
                       "parallelism": par},
            "timed_seconds": dt,
            # a rehearsal of the multi-rank control flow (gloo, ranks sharing a GPU): NOT a scaling number
            "rehearsal": (f"{args.dist_backend} backend, {world} ranks on {torch.cuda.device_count()} GPU(s): control flow only, "
                          f"not a number of record") if rehearsal else None,
            # which code moved the maps (prc = prc_gather_frames: RCCL send/recv group through the C ABI; torch =
            # torch.distributed.gather; none = single GPU or surfaces stay put) and every rank's own clock
            "gather_path": gather_mode, "rccl_version": _lib.rccl_version() if gather_mode == "prc" else None,
            # what RCCL itself reports for the communicator behind prc_gather_frames (ncclCommCount / ncclCommUserRank on
            # every rank, all-gathered) and the bytes of maps every rank handed to the gather per step
            "rccl_nranks": rccl_nranks, "rccl_ranks_seen": rccl_seen, "gathered_bytes_per_rank_per_step": gathered_bytes,
            "rank_ms_per_step": [t / steps * 1e3 for t in rank_seconds],
            # the whole step against the memory system, on the bytes MOVED per frame (counters: this run's passes or the
            # committed per-unit figures); SURVEY 8d's algorithmic accounting (two passes per Doppler bin; more bytes than the
            # fused chain moves, so its "fraction" can exceed what the chip can copy) is kept under its own name
            "hbm_moved_bytes_per_frame": moved_per_frame,
            "hbm_moved_GBps": (moved_per_frame * value / world / 1e9) if moved_per_frame else None,
            "hbm_frac_of_peak": (moved_per_frame * value / world / 1e9 / HBM_PEAK_GBS) if moved_per_frame else None,
            "hbm_frac_of_copy_ceiling": (moved_per_frame * value / world / 1e9 / 6290.0) if moved_per_frame else None,   # MI355X_MICROARCH.md: ~6.3 TB/s achievable
            "hbm_two_pass_accounting": {"bytes_per_frame": per_frame_bytes, "GBps": per_frame_bytes * value / world / 1e9,
                                        "frac_of_peak": per_frame_bytes * value / world / 1e9 / HBM_PEAK_GBS,
                                        "note": "SURVEY 8d algorithmic bytes (20N + 8F(R+1) + 200C): an accounting of the unfused "
                                                "formulation, not bytes that crossed the memory interface"},
            # caf_* timings are of ONE channel's launch; a multi-illuminator step runs all its channels in one launch per stage
            "caf_channels_per_launch": len(refs) if multi else 1,
            # avg_ms_per_launch: HIP events around the kernel ALONE on its stream (the sum over a step of these exceeds
            # ms_per_step: in the step three LS chains and the CAF stream run together); in_step_ms_per_launch: the same kernel's
            # average duration in the overlapped step (kernel trace), where it shares the chip
            "kernels": {k_: {"avg_ms_per_launch": v["ms"], "launches_per_step": v["launches_per_step"], "bound": v["bound"],
                             "units_per_launch": units_of[k_],
                             ("algorithmic_TFLOPs" if v["bound"] == "valu" else "algorithmic_GBps"):
                                 v["work"] / (v["ms"] * 1e-3) / (1e12 if v["bound"] == "valu" else 1e9),
                             "moved_bytes_per_launch": traffic_by.get(k_),
                             "moved_GBps": (traffic_by[k_] / (v["ms"] * 1e-3) / 1e9) if traffic_by.get(k_) else None,
                             "in_step_ms_per_launch": in_step[k_]["avg_ms"] if (in_step and k_ in in_step) else None}
                        for k_, v in kt.items()},
            "roofline": roof,
        }
        solo_sum = sum(v["ms"] * v["launches_per_step"] for v in kt.values())
        result["kernel_time_accounting"] = {
            "solo_sum_ms_per_step": solo_sum, "ms_per_step": dt / steps * 1e3,
            "note": "solo = each kernel alone on the chip (HIP events on its launch stream); the step overlaps streams, so the solo "
                    "sum may exceed ms_per_step, and in the step every kernel runs longer than solo"}
        if in_step:
            child_steps = in_step_note["child_steps"]
            tot = sum(a["total_ms"] for a in in_step.values()) / child_steps
            result["kernel_time_accounting"].update({
                "in_step_sum_ms_per_child_step": tot, "child_ms_per_step": in_step_note["child_ms_per_step"],
                "child_frames_per_step": in_step_note["child_frames"],
                "in_step_mean_concurrency": (tot / in_step_note["child_ms_per_step"]) if in_step_note["child_ms_per_step"] else None,
                "how": in_step_note["how"]})
        elif in_step_note:
            result["kernel_time_accounting"]["in_step"] = f"not measured ({in_step_note})"
        caf_ms = kt["caf_segments"]["ms"] + kt["caf_doppler"]["ms"]
        result["caf_only_frames_per_s_per_gpu"] = nb / (caf_ms * 1e-3)
        result["caf_only_hbm_frac"] = caf_bytes * nb / (caf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(wl, max_workers=None if args.cpu_workers in ("all", "auto") else int(args.cpu_workers), asis=args.cpu_asis,
                              force_all=args.cpu_workers == "all")
            result["cpu_baseline"] = cb
            result["speedup_vs_cpu_all_cores_measured"] = value / cb["value"]
            result["speedup_vs_cpu_1core"] = value / cb["one_core_value"]
            if clutter == "ls":
                result["fm_like_suppression_dB"] = fm_suppression(R if wl == "cfg1" else 64)
            if args.cpu_asis:
                # the reference as it ships: scipy.signal.decimate calls dlti._as_zpk -> np.roots on the (q+1)-tap
                # boxcar once per lag column (no effect on the result; bypassed in every other figure of this line)
                q = n // F
                t0 = time.perf_counter()
                np.roots(np.ones(q + 1))
                t_roots = time.perf_counter() - t0
                per_frame = nill * ((R + 1) * t_roots + cb["one_core_caf_seconds"]) + cb["one_core_clutter_seconds"]
                result["cpu_reference_as_is"] = {
                    "np_roots_seconds_per_lag": t_roots, "lags": R + 1, "seconds_per_frame": per_frame,
                    "frames_per_s": 1.0 / per_frame,
                    "note": f"one np.roots(ones({q + 1})) timed on this box (LAPACK may use several cores), times the "
                            f"{R + 1} lag columns, plus the one-core path above"}
    if result is not None and secondary is not None:
        result["secondary"] = secondary
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    elif comm is not None:
        drain()
        comm.close()
    # The JSON line is the LAST thing this job writes: RCCL prints a version banner through C stdio (fully buffered when
    # stdout is a pipe, so it would otherwise appear at process exit, after the line) -- flush C stdio on every rank first
    _flush_c_stdio()
    if result is not None:
        if world > 1:
            time.sleep(1.0)                         # the other ranks' flushes (same point, no collective left) land first
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
