#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own modules (build container only).

    python oracle/gen_golden.py [--big]

The reference (pure Python) is imported from /root/reference; it never travels to
the GPU box -- only the vectors written here do.  Inputs come from
passiveradar_amd.scene (seeded Philox) and are stored alongside the outputs for
the small cases; the big cases store the seed + scene parameters only.

SciPy-1.15 artefact: ``scipy.signal.decimate`` calls ``dlti._as_zpk()`` -> ``np.roots`` on the
(q+1)-tap boxcar for every lag.  That is O(q^3) per lag and changes nothing in the
output (checked below, bit for bit, on a small case).  For the big cases (q=1024..4882) the
root finder is short-circuited *in this process only* by reporting "no poles" for a
denominator of 1; the reference files are not modified.
"""
import argparse
import json
import os
import sys
import time
import warnings

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)
warnings.filterwarnings("ignore")

import numpy as np
import scipy
import scipy.signal as signal

from passiveRadar import range_doppler_processing as ref_rd      # noqa: E402
from passiveRadar import clutter_removal as ref_cr               # noqa: E402
from passiveRadar import signal_utils as ref_su                  # noqa: E402
from passiveRadar import config as ref_cfg                       # noqa: E402
from passiveradar_amd import scene                               # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
META = {"numpy": np.__version__, "scipy": scipy.__version__,
        "generator": "oracle/gen_golden.py", "reference": "Max-Manning/passiveRadar @ /root/reference"}


class no_root_finding:
    """Context manager: make dlti(num, 1)._as_zpk() report an FIR without calling np.roots."""

    def __enter__(self):
        from scipy.signal._ltisys import TransferFunction, ZerosPolesGain
        self.cls = TransferFunction
        self.orig = TransferFunction.to_zpk
        orig = self.orig

        def to_zpk(tf):
            den = np.atleast_1d(tf.den)
            if den.size == 1 and den[0] == 1:
                return ZerosPolesGain(np.zeros(0), np.zeros(0), 1.0, **tf._dt_dict)
            return orig(tf)
        TransferFunction.to_zpk = to_zpk
        return self

    def __exit__(self, *a):
        self.cls.to_zpk = self.orig


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, meta=json.dumps(META), **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1e3:.1f} kB")


def caf_cases():
    print("fast_xambg small cases")
    cases = [  # name, N_in, R, F, inputLen, window spec, shortFilt, srv dtype
        ("p2", 4096, 7, 64, 4096, "none", True, "c64"),
        ("p2_kaiser_arr", 4096, 7, 64, 4096, "array", True, "c64"),
        ("p2_kaiser_tuple", 4096, 7, 64, 4096, "tuple", True, "c64"),
        ("oddq", 6000, 6, 64, 6000, "array", True, "c64"),
        ("oddq_p1", 6001, 6, 64, 6001, "none", True, "c64"),
        ("nondiv", 4100, 3, 64, 4100, "array", True, "c64"),
        ("small", 2401, 3, 16, 2401, "tuple", True, "c64"),
        ("bigq", 9375, 3, 2, 9375, "none", True, "c64"),
        ("padded", 4000, 7, 64, 4096, "tuple", True, "c64"),
        ("longfilt", 4096, 7, 64, 4096, "array", False, "c64"),
        ("srv128", 4096, 7, 64, 4096, "array", True, "c128"),
        ("lags_gt_q", 4096, 40, 256, 4096, "array", True, "c64"),   # R > q=16
    ]
    for i, (name, n, R, F, ilen, wspec, short, sdt) in enumerate(cases):
        a, s = scene.make_scene(n, 8000.0, R, scene.scene_seed(90, i),
                                targets=((max(R - 2, 1), 31.0, 0.05),))
        if sdt == "c128":
            s = s.astype(np.complex128)
        if wspec == "none":
            w = None
        elif wspec == "array":
            w = signal.get_window(("kaiser", 5.0), ilen)
        else:
            w = ("kaiser", 5.0)
        out = ref_rd.fast_xambg(a, s, R, F, ilen, w, short)
        save("caf_" + name, ref=a, srv=s, R=R, F=F, inputLen=ilen,
             window=(np.zeros(0) if w is None else (w if wspec == "array" else np.array([5.0]))),
             wspec=wspec, shortFilt=short, out=out)


def artefact_check():
    a, s = scene.make_scene(4096, 8000.0, 7, scene.scene_seed(91))
    w = signal.get_window(("kaiser", 5.0), 4096)
    x0 = ref_rd.fast_xambg(a, s, 7, 64, 4096, w)
    with no_root_finding():
        x1 = ref_rd.fast_xambg(a, s, 7, 64, 4096, w)
    assert np.array_equal(x0, x1), "root-finding bypass changed the output"
    print("root-finding bypass is bit-identical on the small case")


def helper_cases():
    print("xcorr / frequency_shift")
    a, s = scene.make_scene(5000, 8000.0, 20, scene.scene_seed(92))
    save("xcorr", s1=a, s2=s,
         z_0_20=ref_su.xcorr(a, s, 0, 20), z_7_0=ref_su.xcorr(a, s, 7, 0),
         z_3_9=ref_su.xcorr(a, s, 3, 9), z_auto=ref_su.xcorr(a, a, 0, 15))
    x, _ = scene.make_scene(200000, 262184.87, 8, scene.scene_seed(93))
    dec = slice(None, None, 16)          # outputs stored decimated; x is regenerated from the seed
    save("freqshift", seed=scene.scene_seed(93), n=200000, fs=262184.87, x_head=x[:64], stride=16,
         y_p1=ref_su.frequency_shift(x, 1, 262184.87)[dec], y_m2=ref_su.frequency_shift(x, -2, 262184.87)[dec],
         y_f=ref_su.frequency_shift(x, 37.5, 262184.87)[dec],
         y_ph=ref_su.frequency_shift(x, 80.0, 262184.87, 0.3)[dec])


def helper_uneven_cases():
    """Round 5: argument forms the reference's expressions accept and its own call sites never use -- xcorr of two
    signals of different lengths (signal_utils.py:29-32: 'valid' correlation of s1 with the padded s2, whichever is
    longer) and frequency_shift with one phase per sample (signal_utils.py:24-27 broadcasts the array)."""
    print("xcorr with unequal lengths / frequency_shift with a phase array")
    a, s = scene.make_scene(3000, 8000.0, 20, scene.scene_seed(95))
    cases = [(3000, 2500, 4, 9), (3000, 2990, 6, 7), (2000, 3000, 5, 11), (3000, 1500, 0, 0), (1200, 3000, 30, 0),
             (3000, 2991, 4, 4), (3000, 2992, 4, 4), (64, 3000, 2, 3), (3000, 64, 2, 3)]
    out = {}
    for i, (n1, n2, nlead, nlag) in enumerate(cases):
        out[f"z{i}"] = ref_su.xcorr(a[:n1], s[:n2], nlead, nlag)
    save("xcorr_uneven", s1=a, s2=s, cases=np.array(cases), **out)
    n, fs = 24000, 262184.87
    x, _ = scene.make_scene(n, fs, 8, scene.scene_seed(96))
    rng = np.random.default_rng(96)
    ph64 = rng.uniform(-40.0, 40.0, n)
    ph32 = ph64.astype(np.float32)
    phi = rng.integers(-9, 10, n)
    dec = slice(None, None, 8)
    save("freqshift_phases", seed=scene.scene_seed(96), n=n, fs=fs, stride=8, ph64=ph64, phi=phi,
         y64=ref_su.frequency_shift(x, 37500.5, fs, ph64)[dec], y32=ref_su.frequency_shift(x, 37500.5, fs, ph32)[dec],
         yi=ref_su.frequency_shift(x, -12.25, fs, phi)[dec],
         y1_32=ref_su.frequency_shift(x, 80.0, fs, np.array([0.3], np.float32))[dec])


def ls_cases():
    print("LS filters")
    n, L = 12000, 30
    a, s = scene.make_scene(n, 262184.0, L, scene.scene_seed(94))
    o, t = ref_cr.LS_Filter_Toeplitz(a, s, L, return_filter=True)
    save("ls_toeplitz_white", ref=a, srv=s, L=L, peek=10, out=o, taps=t)
    o, t = ref_cr.LS_Filter_Toeplitz(a, s, L, peek=0, return_filter=True)
    save("ls_toeplitz_peek0", ref=a, srv=s, L=L, peek=0, out=o, taps=t)
    ac, sc = scene.make_scene(n, 262184.0, L, scene.scene_seed(95), colour=(1.0, 0.5, 0.2))
    o, t = ref_cr.LS_Filter_Toeplitz(ac, sc, L, return_filter=True)
    save("ls_toeplitz_coloured", ref=ac, srv=sc, L=L, peek=10, out=o, taps=t)
    o = ref_cr.LS_Filter_Multiple(a, s, L, 262184.0, [0, 1, -1, 2, -2])
    save("ls_multiple", ref=a, srv=s, L=L, fs=262184.0, bins=np.array([0, 1, -1, 2, -2]), out=o)
    n2 = 8192
    a2, s2 = scene.make_scene(n2, 262184.0, L, scene.scene_seed(96))
    o, t = ref_cr.LS_Filter(a2, s2, L, return_filter=True)
    save("ls_direct", ref=a2, srv=s2, L=L, reg=1.0, peek=10, out=o, taps=t)
    o, t = ref_cr.LS_Filter(a2, s2, L, reg=0.25, peek=3, return_filter=True)
    save("ls_direct_reg", ref=a2, srv=s2, L=L, reg=0.25, peek=3, out=o, taps=t)


def nlms_cases():
    print("NLMS")
    n, L = 4000, 20
    a, s = scene.make_scene(n, 262184.0, L, scene.scene_seed(97))
    o, t = ref_cr.NLMS_filter(a, s, L, 0.05, returnFilter=True)
    save("nlms", ref=a, srv=s, L=L, mu=0.05, peek=10, out=o, taps=t)
    o2, t2 = ref_cr.NLMS_filter(a, s, 999, 0.02, initialTaps=t.copy(), returnFilter=True)
    save("nlms_warm", ref=a, srv=s, mu=0.02, peek=10, initialTaps=t, out=o2, taps=t2)
    o3, t3 = ref_cr.NLMS_filter(a, s, 70, 0.08, peek=4, returnFilter=True)     # T=74 > one wave
    save("nlms_t74", ref=a, srv=s, L=70, mu=0.08, peek=4, out=o3, taps=t3)


def nlms_long_cases():
    """filters longer than the 2048 taps of one wavefront (two / four wavefronts per stream on the device): the reference's own
    NLMS_filter on a short seeded scene; inputs are regenerated from the seed, the full output and the final taps are stored"""
    print("NLMS, 2110 and 4110 taps")
    for L in (2100, 4100):
        n = L + 10 + 1300
        seed = 7000 + L
        a, s = scene.make_scene(n, 1e4, 50, seed)
        o, t = ref_cr.NLMS_filter(a, s, L, 0.05, returnFilter=True)
        save(f"nlms_t{L + 10}", seed=seed, N=n, L=L, fs=1e4, scene_R=50, mu=0.05, peek=10, out=o, taps=t)


def config_cases():
    print("getConfiguration")
    import tempfile
    import yaml
    d = ref_cfg.getConfiguration("/root/reference/PRconfig.yaml")
    with open(os.path.join(GOLD, "config_prconfig.json"), "w") as fh:
        json.dump(d, fh, indent=1, sort_keys=True)
    base = yaml.safe_load(open("/root/reference/PRconfig.yaml"))
    base.update(cpi_seconds_nominal=1.0, max_doppler_nominal=128.0, max_range_nominal=292.7)
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as fh:
        yaml.safe_dump(base, fh)
    d1 = ref_cfg.getConfiguration(fh.name)
    os.unlink(fh.name)
    with open(os.path.join(GOLD, "config_cfg1.json"), "w") as fh:
        json.dump({"yaml_overrides": {"cpi_seconds_nominal": 1.0, "max_doppler_nominal": 128.0,
                                      "max_range_nominal": 292.7}, "derived": d1},
                  fh, indent=1, sort_keys=True)


def stream_case():
    """main.py:169-194 emulated in plain NumPy around the two imported reference functions."""
    print("stream (block pipeline) case")
    C, R, F, nch = 8192, 16, 64, 6
    cpi = 2 * C
    fs = 262184.0
    a, s = scene.make_stream(nch, C, fs, R, scene.scene_seed(98))
    cleaned = np.concatenate([
        ref_cr.LS_Filter_Multiple(a[i * C:(i + 1) * C], s[i * C:(i + 1) * C], R, fs, [0, 1, -1, 2, -2])
        for i in range(nch)])
    depth = cpi // 4
    pad = np.zeros(depth)
    ap = np.concatenate((pad, a, pad))
    sp = np.concatenate((pad, cleaned, pad))
    w = signal.get_window(("kaiser", 5.0), cpi)
    frames = [ref_rd.fast_xambg(ap[i * C:i * C + cpi], sp[i * C:i * C + cpi], R, F, cpi, w)
              for i in range(nch)]
    save("stream", ref=a, srv=s, C=C, R=R, F=F, fs=fs, cleaned=cleaned.astype(np.complex64),
         out=np.concatenate(frames, axis=2))


def frontend_case():
    """the front end of main.py:105-166 emulated around the reference's own
    deinterleave_IQ / frequency_shift (array block phase) / resample."""
    print("front end (deinterleave / block-phase tuning / 13:119 resampler)")
    rng = np.random.Generator(np.random.Philox(key=scene.scene_seed(99)))
    icl, nblk = 36000, 3                       # raw scalars per block (18000 complex samples)
    fs, foff, up, dn = 2400000, 100000, 13, 119
    raw8 = rng.integers(-100, 100, size=icl * nblk, dtype=np.int8)
    mod_period = fs // foff
    per_block = (icl // 2) % mod_period
    phases = 2 * np.pi * np.arange(nblk) * per_block * (foff / fs)
    outs, tuned1, deint1 = [], None, None
    for i in range(nblk):
        blk = ref_su.deinterleave_IQ(raw8[i * icl:(i + 1) * icl])
        tuned = ref_su.frequency_shift(blk, foff, fs, np.array([phases[i]]))
        if i == 1:
            tuned1, deint1 = tuned, blk
        outs.append(ref_su.resample(tuned, up, dn))
    rawf = (rng.standard_normal(20001) * 3).astype(np.float32)          # odd length: last scalar dropped
    save("frontend", raw8=raw8, icl=icl, fs=fs, foff=foff, up=up, dn=dn, phases=phases,
         deint1=deint1, tuned1=tuned1, out=np.concatenate(outs),
         rawf=rawf, deintf=ref_su.deinterleave_IQ(rawf),
         res_c64=ref_su.resample(deint1, 3, 7), res_simple=ref_su.resample(deint1[:5000], 13, 119))


def cfar_case():
    """CFAR_2D (target_detection.py:683-703) on a |fast_xambg| map."""
    print("CFAR_2D")
    np.float = float          # target_detection.py:10-17 uses aliases removed in NumPy >= 1.24
    np.int = int
    from passiveRadar import target_detection as ref_td
    n, R, F = 16384, 40, 64
    a, s = scene.make_scene(n, 1e4, R, scene.scene_seed(89), targets=((13, 37.0, 0.2), (30, -11.0, 0.1)))
    X = np.abs(ref_rd.fast_xambg(a, s, R, F))[:, :, 0]
    cr = ref_td.CFAR_2D(X, 18, 4)
    thr = float(np.sort(cr.ravel())[-12])          # a threshold that keeps a dozen cells
    save("cfar", X=X, cr_18_4=cr, cr_7_2=ref_td.CFAR_2D(X, 7, 2), thr=thr,
         det_18_4=ref_td.CFAR_2D(X, 18, 4, thr))


def offset_case():
    """find_channel_offset (signal_utils.py:73-78) as main.py:54/:83 calls it (nd=1) and
    as signal_preview.py:36 does (nd=4); plus the intermediate signal.decimate output it is built on."""
    print("find_channel_offset")
    rng = np.random.Generator(np.random.Philox(key=scene.scene_seed(77)))

    def cw(n):
        return (rng.standard_normal(2 * n, dtype=np.float32) * np.float32(np.sqrt(0.5))).view(np.complex64)

    out = {}
    cases = (("a", 20000, 20000, 1, 1500, 137), ("b", 30001, 30001, 4, 500, -91), ("c", 16000, 13000, 2, 2000, 420),
             ("d", 4096, 4096, 1, 64, 0), ("e", 9000, 9000, 1, 100, -100))
    for tag, n1, n2, nd, nl, shift in cases:
        base = cw(max(n1, n2) + 1000)
        s1 = base[500:500 + n1].copy()
        s2 = (0.7 * base[500 - shift:500 - shift + n2] + 0.5 * cw(n2)).astype(np.complex64)   # s2[n] = s1[n - shift]
        B1 = signal.decimate(s1, nd)
        B2 = np.pad(signal.decimate(s2, nd), (nl, nl), "constant")
        xc = np.abs(signal.correlate(B1, B2, mode="valid"))
        off = ref_su.find_channel_offset(s1, s2, nd, nl)
        assert off == (np.argmax(xc) - nl) * nd
        print(f"  case {tag}: n=({n1},{n2}) nd={nd} nl={nl} true shift {shift} -> offset {off}")
        out.update({f"{tag}_s1": s1, f"{tag}_s2": s2, f"{tag}_nd": nd, f"{tag}_nl": nl, f"{tag}_offset": int(off),
                    f"{tag}_xc": xc.astype(np.float32), f"{tag}_B1_head": B1[:1500].astype(np.complex64), f"{tag}_B1_tail": B1[-1500:].astype(np.complex64)})
    save("channel_offset", cases=np.array([c[0] for c in cases]), **out)


def big_cases():
    print("big CAF cases (root finder short-circuited)")
    with no_root_finding():
        # config 1: N=262144, R=256, F=256
        n, R, F, fs = 262144, 256, 256, 262184.87
        a, s = scene.make_scene(n, fs, R, scene.scene_seed(1))
        w = signal.get_window(("kaiser", 5.0), n)
        t0 = time.time()
        out = ref_rd.fast_xambg(a, s, R, F, n, w)
        print(f"  cfg1 {time.time() - t0:.1f}s")
        save("caf_cfg1", seed=scene.scene_seed(1), N=n, R=R, F=F, fs=fs, out=out)
        # config 2: N=2.4e6, R=256, F=512
        n, R, F, fs = 2400000, 256, 512, 2.4e6
        a, s = scene.make_scene(n, fs, R, scene.scene_seed(2))
        w = signal.get_window(("kaiser", 5.0), n)
        t0 = time.time()
        out = ref_rd.fast_xambg(a, s, R, F, n, w)
        print(f"  cfg2 {time.time() - t0:.1f}s")
        save("caf_cfg2", seed=scene.scene_seed(2), N=n, R=R, F=F, fs=fs, out=out)
        # config 3: N=5e6, R=1024, F=1024 -- store a decimated digest (8.4 MB is not "small")
        n, R, F, fs = 5000000, 1024, 1024, 1.0e7
        a, s = scene.make_scene(n, fs, R, scene.scene_seed(3))
        w = signal.get_window(("kaiser", 5.0), n)
        t0 = time.time()
        out = ref_rd.fast_xambg(a, s, R, F, n, w)
        print(f"  cfg3 {time.time() - t0:.1f}s")
        sub = out[::8, ::8, 0]
        mag = np.abs(out[:, :, 0])
        top = np.argsort(mag.ravel())[-64:]
        save("caf_cfg3_digest", seed=scene.scene_seed(3), N=n, R=R, F=F, fs=fs,
             sub=sub, top_idx=top, top_val=out[:, :, 0].ravel()[top],
             col_sums=out[:, :, 0].sum(axis=0), row_sums=out[:, :, 0].sum(axis=1),
             peak=mag.max())


def _surface_digest(out2d, step=16):
    """What the full-size digests keep of a surface: every step-th cell, the 64 strongest cells, row and column sums."""
    mag = np.abs(out2d)
    top = np.argsort(mag.ravel())[-64:]
    return dict(sub=out2d[::step, ::step], top_idx=top, top_val=out2d.ravel()[top], col_sums=out2d.sum(axis=0),
                row_sums=out2d.sum(axis=1), peak=mag.max())


def _cfg5_one(i):
    n, R, F, fs = 1 << 23, 2048, 2048, 2.0e7
    seeds = [scene.scene_seed(5, c) for c in range(4)]
    refs, srv = scene.make_multi_scene(n, fs, R, seeds)
    w = signal.get_window(("kaiser", 5.0), n)
    t0 = time.time()
    with no_root_finding():
        out = ref_rd.fast_xambg(refs[i], srv, R, F, n, w)[:, :, 0]
    print(f"  cfg5 illuminator {i}: {time.time() - t0:.1f}s", flush=True)
    return _surface_digest(out)


def caf_cfg5_digest_case():
    """BASELINE config 5 at full size through the reference's own fast_xambg: 20 MS/s, N = 2^23, 2048 x 2048, four
    independent white illuminators against one surveillance channel that carries every illuminator's scene
    (range_doppler_processing.py:12-90 once per pair, as the reference would be called).  One digest per surface."""
    print("config-5 CAF digests (4 surfaces of 2049 x 2048; four worker processes, several minutes each)")
    import multiprocessing as mp
    with mp.get_context("fork").Pool(4) as pool:
        digs = pool.map(_cfg5_one, range(4))
    arrays = {}
    for i, d in enumerate(digs):
        for k, v in d.items():
            arrays[f"ill{i}_{k}"] = v
    save("caf_cfg5_digest", seeds=np.array([scene.scene_seed(5, c) for c in range(4)]), N=1 << 23, R=2048, F=2048,
         fs=2.0e7, **arrays)


def nlms_cfg3_digest_case():
    """BASELINE config 3's clutter stage at full hop length through the reference's own NLMS_filter
    (clutter_removal.py:189-249): 2.5 M samples, filterLen 1024 (T = 1034), mu = 0.02, cold start.  Kept: every
    997th output sample, the first and last 4096, and the final taps."""
    print("config-3 NLMS hop (2.5 M sequential steps of the reference's Python loop, about a minute or two)")
    n, L, fs = 2500000, 1024, 1.0e7
    a, s = scene.make_scene(n, fs, L, scene.scene_seed(3))
    t0 = time.time()
    out, taps = ref_cr.NLMS_filter(a, s, L, 0.02, 10, None, True)
    print(f"  {time.time() - t0:.1f}s")
    save("nlms_cfg3_digest", seed=scene.scene_seed(3), N=n, L=L, fs=fs, mu=0.02, peek=10,
         sub=out[::997].astype(np.complex64), head=out[:4096].astype(np.complex64),
         tail=out[-4096:].astype(np.complex64), taps=np.asarray(taps).astype(np.complex64),
         peak=np.abs(out).max(), energy=np.float64(np.vdot(out, out).real))


def ls_cfg1_case():
    """BASELINE config 1 (the reference's own CPU-runnable case): one 131 072-sample hop chunk of the 262 184.87 Hz
    IF stream through LS_Filter_Multiple with 256 range cells (T = 266) and the five Doppler bins of main.py:169-176."""
    print("config-1 LS chunk (LS_Filter_Multiple, T=266, 5 bins)")
    n, R, fs = 131072, 256, 262184.87
    a, s = scene.make_scene(n, fs, R, scene.scene_seed(11))
    t0 = time.time()
    out = ref_cr.LS_Filter_Multiple(a, s, R, fs, [0, 1, -1, 2, -2])
    print(f"  {time.time() - t0:.1f}s")
    save("ls_cfg1", seed=scene.scene_seed(11), N=n, R=R, fs=fs, out_sub=out[::7].astype(np.complex64),
         head=out[:600].astype(np.complex64), tail=out[-600:].astype(np.complex64))


def pipeline_cfg1_case():
    """BASELINE config 1 end to end (the reference's own CPU-runnable case, PRconfig.yaml with a 1 s CPI): three
    131 072-sample hop chunks through LS_Filter_Multiple, the middle overlapped frame through fast_xambg
    (256 range x 256 Doppler, Kaiser(5) window), main.py:169-194 geometry."""
    print("config-1 pipeline frame (LS x5 + CAF)")
    n, R, F, fs = 262144, 256, 256, 262184.87
    C = n // 2
    a, s = scene.make_stream(3, C, fs, R, scene.scene_seed(12))
    cleaned = np.concatenate([
        ref_cr.LS_Filter_Multiple(a[i * C:(i + 1) * C], s[i * C:(i + 1) * C], R, fs, [0, 1, -1, 2, -2])
        for i in range(3)])
    pad = np.zeros(n // 4)
    ap = np.concatenate((pad, a, pad))
    sp = np.concatenate((pad, cleaned, pad))
    w = signal.get_window(("kaiser", 5.0), n)
    with no_root_finding():
        out = ref_rd.fast_xambg(ap[C:C + n], sp[C:C + n], R, F, n, w)
    save("pipeline_cfg1", seed=scene.scene_seed(12), N=n, R=R, F=F, fs=fs, frame_index=1,
         out=out[:, :, 0].astype(np.complex64))


def pipeline_cfg2_case():
    """BASELINE config 2 end to end, from the reference's own functions: three 1.2 M-sample hop chunks through
    LS_Filter_Multiple (5 Doppler bins, T=266) and the middle overlapped frame (main.py:169-194 geometry)
    through fast_xambg with the Kaiser(5) window.  Inputs are regenerated from the seed by the tests."""
    print("config-2 pipeline frame (LS x5 + CAF at full size)")
    n, R, F, fs = 2400000, 256, 512, 2.4e6
    C = n // 2
    a, s = scene.make_stream(3, C, fs, R, scene.scene_seed(4))
    t0 = time.time()
    cleaned = np.concatenate([
        ref_cr.LS_Filter_Multiple(a[i * C:(i + 1) * C], s[i * C:(i + 1) * C], R, fs, [0, 1, -1, 2, -2])
        for i in range(3)])
    print(f"  LS_Filter_Multiple x3 chunks {time.time() - t0:.1f}s")
    depth = n // 4
    pad = np.zeros(depth)
    ap = np.concatenate((pad, a, pad))
    sp = np.concatenate((pad, cleaned, pad))
    w = signal.get_window(("kaiser", 5.0), n)
    t0 = time.time()
    with no_root_finding():
        out = ref_rd.fast_xambg(ap[C:C + n], sp[C:C + n], R, F, n, w)
    print(f"  fast_xambg {time.time() - t0:.1f}s")
    save("pipeline_cfg2", seed=scene.scene_seed(4), N=n, R=R, F=F, fs=fs, frame_index=1,
         cleaned_sub=cleaned[::101].astype(np.complex64), out=out[:, :, 0].astype(np.complex64))


def pipeline_cfg2_c128_case():
    """BASELINE config 2 end to end with the reference's own functions fed complex128 inputs: its
    scipy.signal.correlate / np.convolve then sum the 1.2 M-term correlations in double (clutter_removal.py:142-155,
    signal_utils.py:29-32), which removes the float32 summation noise that sits on the zero-Doppler ridge of
    pipeline_cfg2.npz.  Same scene, same geometry (three hop chunks, the middle overlapped frame)."""
    print("config-2 pipeline frame, complex128 inputs to the reference (LS x5 + CAF at full size)")
    n, R, F, fs = 2400000, 256, 512, 2.4e6
    C = n // 2
    a, s = scene.make_stream(3, C, fs, R, scene.scene_seed(4))
    a = a.astype(np.complex128)
    s = s.astype(np.complex128)
    t0 = time.time()
    cleaned = np.concatenate([
        ref_cr.LS_Filter_Multiple(a[i * C:(i + 1) * C], s[i * C:(i + 1) * C], R, fs, [0, 1, -1, 2, -2])
        for i in range(3)])
    print(f"  LS_Filter_Multiple x3 chunks {time.time() - t0:.1f}s  dtype {cleaned.dtype}")
    pad = np.zeros(n // 4)
    ap = np.concatenate((pad, a, pad))
    sp = np.concatenate((pad, cleaned, pad))
    w = signal.get_window(("kaiser", 5.0), n)
    t0 = time.time()
    with no_root_finding():
        out = ref_rd.fast_xambg(ap[C:C + n], sp[C:C + n], R, F, n, w)
    print(f"  fast_xambg {time.time() - t0:.1f}s")
    save("pipeline_cfg2_c128", seed=scene.scene_seed(4), N=n, R=R, F=F, fs=fs, frame_index=1,
         cleaned_sub=cleaned[::101].astype(np.complex64), out=out[:, :, 0].astype(np.complex64))


def pipeline_prconfig_raw_case():
    """The reference's published workload, end to end, as shipped (PRconfig.yaml unmodified: 2 s CPI, 1024 Doppler x 176
    range cells, LS x 5 bins, T = 185): three raw int8 blocks of 4 799 250 scalars per channel through the reference's own
    deinterleave_IQ -> frequency_shift(block phase) -> resample(13, 119) -> LS_Filter_Multiple -> fast_xambg in the
    geometry of main.py:105-194 (50 % overlapped CPIs, zero boundary, Kaiser(5) window); the middle frame, a digest of
    the IF streams and of the cleaned stream."""
    print("PRconfig.yaml raw -> frame (front end + LS x5 + CAF at the shipped sizes)")
    cfg = ref_cfg.getConfiguration("/root/reference/PRconfig.yaml")
    icl, fs_in, foff = cfg["input_chunk_length"], cfg["input_sample_rate"], cfg["offset_freq"]
    up, dn, C, n = cfg["resamp_up"], cfg["resamp_dn"], cfg["output_chunk_length"], cfg["cpi_samples"]
    R, F, fs_if = cfg["num_range_cells"], cfg["num_doppler_cells"], cfg["IF_sample_rate"]
    nblk = 3
    seed = scene.scene_seed(14)
    raw_ref, raw_srv = scene.make_raw_stream(nblk, icl, fs_in, foff, seed)
    t0 = time.time()
    mod_period = fs_in // foff
    per_block = (icl // 2) % mod_period
    phases = 2 * np.pi * np.arange(nblk) * per_block * (foff / fs_in)          # main.py:125-131

    def front(raw):
        out = []
        for i in range(nblk):
            blk = ref_su.deinterleave_IQ(raw[i * icl:(i + 1) * icl])
            tuned = ref_su.frequency_shift(blk, foff, fs_in, np.array([phases[i]]))
            out.append(ref_su.resample(tuned, up, dn))
        return np.concatenate(out)
    a, s = front(raw_ref), front(raw_srv)
    assert a.shape[0] == nblk * C, (a.shape, C)
    print(f"  front end {time.time() - t0:.1f}s")
    # main.py:169-176 declares dtype=complex64 for these arrays; the functions receive what resample returned (complex128)
    t0 = time.time()
    cleaned = np.concatenate([
        ref_cr.LS_Filter_Multiple(a[i * C:(i + 1) * C], s[i * C:(i + 1) * C], R, fs_if, [0, 1, -1, 2, -2])
        for i in range(nblk)])
    print(f"  LS_Filter_Multiple x{nblk} {time.time() - t0:.1f}s")
    depth = cfg["window_overlap"]
    pad = np.zeros(depth)
    ap = np.concatenate((pad, a, pad))
    sp = np.concatenate((pad, cleaned, pad))
    w = signal.get_window(("kaiser", 5.0), n)
    t0 = time.time()
    with no_root_finding():
        out = ref_rd.fast_xambg(ap[C:C + n], sp[C:C + n], R, F, n, w)
    print(f"  fast_xambg {time.time() - t0:.1f}s")
    keys = ("input_chunk_length", "input_sample_rate", "offset_freq", "resamp_up", "resamp_dn", "output_chunk_length",
            "cpi_samples", "num_range_cells", "num_doppler_cells", "IF_sample_rate", "window_overlap")
    save("pipeline_prconfig_raw", seed=seed, nblk=nblk, frame_index=1,
         **{"cfg_" + k: cfg[k] for k in keys},
         raw_ref_checksum=np.array(scene.raw_checksum(raw_ref), dtype=np.int64),
         raw_srv_checksum=np.array(scene.raw_checksum(raw_srv), dtype=np.int64),
         if_ref_sub=a[::61].astype(np.complex64), if_srv_sub=s[::61].astype(np.complex64),
         if_ref_rms=float(np.sqrt(np.mean(np.abs(a) ** 2))), if_srv_rms=float(np.sqrt(np.mean(np.abs(s) ** 2))),
         cleaned_sub=cleaned[::61].astype(np.complex64), cleaned_rms=float(np.sqrt(np.mean(np.abs(cleaned) ** 2))),
         out=out[:, :, 0].astype(np.complex64))


def caf_longfilt_cfg1_case():
    """fast_xambg(..., shortFilt=False) at a BASELINE size (config 1: N = 262 144, 257 lags, 256 Doppler bins): the
    decimation filter is firwin(10 q + 1, 1/q, 'flattop') with q = 1024, 10 241 taps per output sample
    (range_doppler_processing.py:73-78)."""
    print("config-1 CAF with the long decimation FIR (shortFilt=False)")
    n, R, F, fs = 262144, 256, 256, 262184.87
    a, s = scene.make_scene(n, fs, R, scene.scene_seed(1))
    w = signal.get_window(("kaiser", 5.0), n)
    t0 = time.time()
    with no_root_finding():
        out = ref_rd.fast_xambg(a, s, R, F, n, w, shortFilt=False)
    print(f"  {time.time() - t0:.1f}s")
    save("caf_longfilt_cfg1", seed=scene.scene_seed(1), N=n, R=R, F=F, fs=fs, out=out[:, :, 0].astype(np.complex64))


def ls_wide_cases():
    """LS_Filter_Multiple with Doppler bins far from zero (|2 pi f/Fs| * peek up to ~0.4 rad: the reference accepts
    any bin list, clutter_removal.py:178-187) and the LS filters at the config-3 tap count T = 1034."""
    print("LS filters: large Doppler bins, T = 1034")
    n, L = 12000, 30
    a, s = scene.make_scene(n, 1.0e4, L, scene.scene_seed(94))
    save("ls_multiple_bin50", ref=a, srv=s, L=L, fs=1.0e4, bins=np.array([0.0, 50.0]),
         out=ref_cr.LS_Filter_Multiple(a, s, L, 1.0e4, [0, 50]))
    save("ls_multiple_kHz", ref=a, srv=s, L=L, fs=262184.87, bins=np.array([0.0, 1500.0, -1700.0, 40.0]),
         out=ref_cr.LS_Filter_Multiple(a, s, L, 262184.87, [0, 1500, -1700, 40]))
    n, L = 40000, 1024
    a, s = scene.make_scene(n, 1.0e7, L, scene.scene_seed(31))
    t0 = time.time()
    o, t = ref_cr.LS_Filter_Toeplitz(a, s, L, return_filter=True)
    om = ref_cr.LS_Filter_Multiple(a, s, L, 1.0e7, [0, 1, -1])
    print(f"  T=1034 Toeplitz + Multiple {time.time() - t0:.1f}s")
    save("ls_t1034", seed=scene.scene_seed(31), N=n, L=L, fs=1.0e7, bins=np.array([0.0, 1.0, -1.0]),
         out=o.astype(np.complex64), taps=t, out_multiple=om.astype(np.complex64))
    n2 = 12288
    a2, s2 = scene.make_scene(n2, 1.0e7, L, scene.scene_seed(32))
    t0 = time.time()
    o2, t2 = ref_cr.LS_Filter(a2, s2, L, return_filter=True)
    print(f"  T=1034 LS_Filter (N x T matrix of {n2 * 1034 * 8 / 1e6:.0f} MB) {time.time() - t0:.1f}s")
    save("ls_direct_t1034", seed=scene.scene_seed(32), N=n2, L=L, fs=1.0e7, reg=1.0, out=o2, taps=t2)


def direct_xambg_case():
    """Round 6: range_doppler_processing.py:93-124 (never on the reference's processing path; the drop-in exists so that an
    import swap does not raise).  Two shapes: an echo at a known (Doppler, delay) cell on a power-of-two CPI, and an odd
    length with a non-integer Doppler step."""
    print("direct_xambg")
    out = {}
    for tag, n, R, F, fs, i in (("a", 4096, 20, 64, 4096.0, 0), ("b", 3001, 9, 16, 8000.0, 1)):
        a, s = scene.make_scene(n, fs, R, scene.scene_seed(96, i), targets=((7, 5.0 * fs / n, 0.5),))
        out.update({f"ref_{tag}": a, f"srv_{tag}": s, f"R_{tag}": R, f"F_{tag}": F, f"fs_{tag}": fs,
                    f"out_{tag}": ref_rd.direct_xambg(a, s, R, F, fs)})
    save("direct_xambg", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also generate the cfg1/2/3 CAF goldens (minutes)")
    ap.add_argument("--only-big", action="store_true")
    ap.add_argument("--only", default=None, help="run a single case function by name, e.g. offset_case")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if args.only:
        globals()[args.only]()
        sys.exit(0)
    if not args.only_big:
        artefact_check()
        caf_cases()
        helper_cases()
        helper_uneven_cases()
        ls_cases()
        nlms_cases()
        nlms_long_cases()
        config_cases()
        stream_case()
        frontend_case()
        cfar_case()
        offset_case()
        ls_cfg1_case()
        ls_wide_cases()
        direct_xambg_case()
    if args.big or args.only_big:
        big_cases()
        pipeline_cfg1_case()
        pipeline_cfg2_case()
        pipeline_cfg2_c128_case()
        nlms_cfg3_digest_case()
        caf_cfg5_digest_case()
        pipeline_prconfig_raw_case()
        caf_longfilt_cfg1_case()
