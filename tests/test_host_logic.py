"""Host-side logic that needs no GPU: config derivation vs the reference's dict, scene generator,
frame sharding arithmetic, argument validation that happens before any device call."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REPO
from passiveradar_amd import scene
from passiveradar_amd.config import derive, getConfiguration, nearestpow2, nextpow2
from passiveradar_amd.stream import plan_shard

REF_YAML = """
input_file: 'PassiveRadar_20191102_1011.hdf5'
interleaved_input_channels: False
input_ref_path: '/data/ref'
input_srv_path: '/data/srv'
interleaved_data_path: '/data'
range_doppler_map_ftype: 'zarr'
output_fname: 'XAMBG_1011'
num_frames: 1200
input_sample_rate: 2400000
input_center_freq: 102000000
channel_freq: 101900000
channel_bandwidth: 200000
cpi_seconds_nominal: 2.0
max_doppler_nominal: 256.0
max_range_nominal: 200.0
overlap_cpi: True
"""


def _same(a, b):
    if isinstance(a, float) or isinstance(b, float):
        return a == pytest.approx(b, rel=1e-15, abs=0)
    return a == b


def test_config_matches_reference_dict(tmp_path):
    gold = json.load(open(os.path.join(GOLDEN, "config_prconfig.json")))
    p = tmp_path / "PRconfig.yaml"
    p.write_text(REF_YAML)
    cfg = getConfiguration(str(p))
    assert set(cfg) == set(gold)
    for k in gold:
        assert _same(cfg[k], gold[k]), k
    assert cfg["cpi_samples"] == 524288 and cfg["num_range_cells"] == 175 and cfg["num_doppler_cells"] == 1024


def test_config_cfg1_variant(tmp_path):
    import yaml
    g = json.load(open(os.path.join(GOLDEN, "config_cfg1.json")))
    base = yaml.safe_load(REF_YAML)
    base.update(g["yaml_overrides"])
    cfg = derive(base)
    for k, v in g["derived"].items():
        assert _same(cfg[k], v), k
    assert (cfg["cpi_samples"], cfg["num_range_cells"], cfg["num_doppler_cells"]) == (262144, 256, 256)


def test_config_non_overlap_wart_is_kept(tmp_path):
    import yaml
    base = yaml.safe_load(REF_YAML)
    base["overlap_cpi"] = False
    with pytest.raises(KeyError):          # config.py:77 reads config['cpi']
        derive(base)


def test_pow2_helpers():
    assert [nextpow2(i) for i in (1, 2, 3, 400000)] == [1, 2, 4, 524288]
    assert [nearestpow2(i) for i in (3, 5, 6, 7, 1023.8)] == [2, 4, 4, 8, 1024]


def test_scene_is_deterministic_and_peaks_where_expected():
    a1, s1 = scene.make_scene(4096, 8000.0, 20, 123)
    a2, s2 = scene.make_scene(4096, 8000.0, 20, 123)
    assert np.array_equal(a1, a2) and np.array_equal(s1, s2) and a1.dtype == np.complex64
    assert abs(np.mean(np.abs(a1) ** 2) - 1.0) < 0.05
    assert scene.expected_peak_cell(7, 5.0, 4096, 4096, 20, 64) == (27, 13)


@pytest.mark.parametrize("nchunks,world", [(1199, 8), (6, 2), (5, 4), (3, 8), (16, 1)])
def test_plan_shard_covers_every_frame_once(nchunks, world):
    seen = []
    for r in range(world):
        sh = plan_shard(nchunks, r, world)
        seen += list(range(sh.frame_lo, sh.frame_hi))
        if sh.nframes:
            # every chunk a frame touches (i-1, i, i+1, clipped) is resident on the rank
            assert sh.chunk_lo == max(sh.frame_lo - 1, 0) and sh.chunk_hi == min(sh.frame_hi + 1, nchunks)
            assert sh.frame_offset(sh.frame_lo, 8) == (sh.frame_lo - sh.chunk_lo) * 8
    assert seen == list(range(nchunks))


def test_shard_sizes_of_the_600_s_stream():
    """config 4: 1199 frames over 8 GPUs = 7 x 150 + 149; blocks sum to the stream whatever the world size"""
    from passiveradar_amd.stream import shard_sizes
    assert shard_sizes(plan_shard(1199, 0, 8)) == [150] * 7 + [149]
    for world in (1, 2, 3, 5, 8, 16):
        for nchunks in (1, 7, 1199):
            sz = shard_sizes(plan_shard(nchunks, 0, world))
            assert sum(sz) == nchunks and len(sz) == world
            assert sz == [plan_shard(nchunks, r, world).nframes for r in range(world)]


def test_shape_errors_are_raised_before_any_device_call():
    from passiveradar_amd.clutter_removal import LS_Filter, LS_Filter_Multiple, LS_Filter_Toeplitz
    from passiveradar_amd.range_doppler_processing import fast_xambg
    a, b = np.zeros(64, np.complex64), np.zeros(63, np.complex64)
    for fn in (lambda: fast_xambg(a, b, 3, 8), lambda: LS_Filter_Toeplitz(a, b, 4),
               lambda: LS_Filter_Multiple(a, b, 4, 1e3), lambda: LS_Filter(a, b, 4)):
        with pytest.raises(ValueError):
            fn()


def test_output_store_roundtrip(tmp_path):
    """main.py:200-224 geometry: (F, R+1, nframes) complex64, one uncompressed zarr-v2 chunk per frame"""
    from passiveradar_amd import output
    rng = np.random.default_rng(3)
    frames = (rng.standard_normal((5, 8, 4)) + 1j * rng.standard_normal((5, 8, 4))).astype(np.complex64)
    p = output.save_range_doppler_zarr(str(tmp_path / "XAMBG.zarr"), frames)
    meta = json.load(open(os.path.join(p, ".zarray")))
    assert meta["shape"] == [8, 4, 5] and meta["chunks"] == [8, 4, 1] and meta["order"] == "C"
    assert os.path.getsize(os.path.join(p, "0.0.3")) == 8 * 4 * 8
    back = output.load_range_doppler_zarr(p)
    assert np.array_equal(back, np.moveaxis(frames, 0, 2))
    cfg = dict(num_doppler_cells=8, num_range_cells=3, frame_interval=0.5, range_cell_width=1.1,
               doppler_cell_width=0.5, meta_fname=str(tmp_path / "m.npz"))
    m = np.load(output.save_metadata(cfg, 5))
    assert m["frame_timestamps"].shape == (5,) and m["range_bins"].shape == (4,) and m["doppler_bins"].shape == (16,)


def test_zarr_store_follows_the_v2_specification(tmp_path):
    """zarr is not installed here, so the store is held against the v2 storage specification: required .zarray keys
    and types, chunk grid / file names / in-chunk order through a reader written from the spec (which also reads
    layouts the writer never produces), the format dispatch of main.py:208-227."""
    from passiveradar_amd import output
    rng = np.random.default_rng(8)
    frames = (rng.standard_normal((3, 6, 5)) + 1j * rng.standard_normal((3, 6, 5))).astype(np.complex64)
    cfg = dict(range_doppler_map_ftype="zarr", range_doppler_map_fname=str(tmp_path / "X.zarr"))
    p = output.save_range_doppler(cfg, frames)
    meta = json.load(open(os.path.join(p, ".zarray")))
    assert output.validate_zarr_v2_metadata(meta) == np.dtype("<c8")
    assert sorted(os.listdir(p)) == [".zarray", ".zattrs", "0.0.0", "0.0.1", "0.0.2"]
    assert np.array_equal(output.read_zarr_v2(p), np.moveaxis(frames, 0, 2))
    # the reader is generic: a hand-made store with a ragged 2 x 3 chunk grid, F order, '/' separator, a missing chunk
    q = tmp_path / "other.zarr"
    os.makedirs(q / "0")
    os.makedirs(q / "1")
    full = np.arange(5 * 7, dtype="<i4").reshape(5, 7)
    json.dump({"zarr_format": 2, "shape": [5, 7], "chunks": [3, 3], "dtype": "<i4", "compressor": None,
               "fill_value": -1, "order": "F", "filters": None, "dimension_separator": "/"}, open(q / ".zarray", "w"))
    for i in range(2):
        for j in range(3):
            if (i, j) == (1, 1):
                continue
            blk = np.full((3, 3), 99, dtype="<i4")
            part = full[3 * i:3 * i + 3, 3 * j:3 * j + 3]
            blk[:part.shape[0], :part.shape[1]] = part
            blk.ravel(order="F").tofile(q / str(i) / str(j))
    want = full.copy()
    want[3:5, 3:6] = -1
    assert np.array_equal(output.read_zarr_v2(str(q)), want)
    # spec violations are caught
    for bad in ({**meta, "zarr_format": 3}, {k: v for k, v in meta.items() if k != "filters"}, {**meta, "chunks": [6, 5]},
                {**meta, "dtype": "c8"}, {**meta, "order": "K"}, {**meta, "fill_value": 0.0}):
        with pytest.raises(ValueError):
            output.validate_zarr_v2_metadata(bad)
    with pytest.raises(ValueError):
        output.save_range_doppler({**cfg, "range_doppler_map_ftype": "npy"}, frames)


def test_hdf5_output_through_libhdf5(tmp_path):
    """main.py:208-214: dataset '/xambg' (F, R+1, nframes) complex64, written by the HDF5 C library itself (ctypes; h5py is
    not installed) in h5py's complex convention, read back by the library (whole-dataset read, a different path from the
    per-frame hyperslab writes) and inspected byte-wise (HDF5 signature, the data really contiguous in the file)."""
    from passiveradar_amd import output
    if not output.hdf5_available():
        pytest.skip("no libhdf5 on this machine")
    rng = np.random.default_rng(9)
    frames = (rng.standard_normal((4, 6, 5)) + 1j * rng.standard_normal((4, 6, 5))).astype(np.complex64)
    cfg = dict(range_doppler_map_ftype="hdf5", range_doppler_map_fname=str(tmp_path / "XAMBG.hdf5"))
    p = output.save_range_doppler(cfg, frames)
    want = np.ascontiguousarray(np.moveaxis(frames, 0, 2))
    back = output.load_range_doppler_hdf5(p)
    assert back.shape == (6, 5, 4) and back.dtype == np.complex64 and np.array_equal(back, want)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n"
    assert raw.find(want.tobytes()) > 0                       # contiguous layout: the C-order array sits in the file as is
    with pytest.raises(KeyError):
        output.load_range_doppler_hdf5(p, "/nothing")
    # an outside reader: the HDF5 project's own h5dump (what range_doppler_plot.py:43-47 would meet through h5py):
    # dataset path, shape (F, R+1, nframes), the compound {r, i} of little-endian float32 h5py stores complex64 as,
    # and the numbers of one frame
    import re
    import shutil
    import subprocess
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump is None:
        pytest.skip("no h5dump on this machine (the library round trip above passed)")
    hdr = subprocess.run([h5dump, "-H", p], capture_output=True, text=True, check=True).stdout
    assert re.search(r'DATASET "xambg"', hdr)
    assert re.search(r'H5T_COMPOUND\s*\{\s*H5T_IEEE_F32LE "r";\s*H5T_IEEE_F32LE "i";\s*\}', hdr)
    assert re.search(r"DATASPACE\s+SIMPLE\s*\{\s*\(\s*6,\s*5,\s*4\s*\)", hdr)
    txt = subprocess.run([h5dump, "-d", "/xambg", "-s", "0,0,2", "-c", "6,5,1", "-m", "%.9g", p],
                         capture_output=True, text=True, check=True).stdout
    cells = re.findall(r"\((\d+),(\d+),(\d+)\):\s*\{\s*([-+0-9.eE]+),\s*([-+0-9.eE]+)\s*\}", txt)
    assert len(cells) == 30
    for f_, k_, i_, re_, im_ in cells:
        v = frames[int(i_), int(f_), int(k_)]
        assert int(i_) == 2 and np.float32(re_) == v.real and np.float32(im_) == v.imag


def test_iir_decimator_design_matches_scipy():
    """host-side filter design handed to prc_channel_offset / prc_decimate_iir (no device call)"""
    import scipy.signal as sg
    from passiveradar_amd import engine
    for q in (1, 2, 4, 7, 10):
        z, p, k = engine.cheby1_lowpass_zpk(8, 0.05, 0.8 / q)
        z0, p0, k0 = sg.cheby1(8, 0.05, 0.8 / q, output="zpk")
        assert np.allclose(np.sort_complex(p), np.sort_complex(p0), rtol=0, atol=1e-13)
        assert np.allclose(z, z0) and abs(k - k0) <= 1e-12 * abs(k0)
        d = engine.IirDecimator(q)
        assert d.padlen == 27 and np.abs(p).max() ** d.settle < 1e-9 and d.settle % 64 == 0
        assert d.out_len(1001) == -(-1001 // q) and d.n_lags(1001, 1001, 50) == 101
    with pytest.raises(ValueError):
        engine.IirDecimator(0)
    with pytest.raises(ValueError):
        engine.IirDecimator(2.5)


def test_plan_cache_evicts_with_close():
    """every object the per-thread plan cache can hold must be closable (eviction calls close())"""
    from passiveradar_amd import engine
    for cls in (engine.CafPlan, engine.LsPlan, engine.FrontendPlan, engine.IirDecimator):
        assert callable(getattr(cls, "close", None)), cls
    closed = []

    class Fake:
        def __init__(self, k):
            self.k = k

        def close(self):
            closed.append(self.k)

    import threading
    def work():
        for k in range(5):
            engine.cached_plan(("fake", k), lambda k=k: Fake(k), limit=3)
    t = threading.Thread(target=work)
    t.start(); t.join()
    assert closed == [0, 1]


def _committed_bench_tags():
    import glob
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    return sorted(os.path.basename(p)[:3] for p in glob.glob(os.path.join(here, "r??_bench_default.json")))


@pytest.mark.parametrize("tag", _committed_bench_tags())          # every round's committed line, r01 .. the latest
def test_committed_bench_line_follows_the_contract(tag):
    """profiles/rNN_bench_default.json is bench.py's own output on the GPU box; the driver's contract fields must be there"""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_bench_default.json")
    d = json.load(open(path))
    base = json.load(open(os.path.join(os.path.dirname(path), "..", "BASELINE.json")))
    assert d["metric"].replace("x", "×") == base["metric"] or d["metric"] == base["metric"].replace("×", "x")
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    frames = d["config"]["frames_per_gpu_per_step"]
    assert frames == (256 if tag == "r01" else 5632)
    assert abs(d["value"] - frames * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) / d["value"] < 1e-6
    if tag != "r01":
        assert d["timed_seconds"] >= 5.0                                       # long enough for the driver's sampler
        assert c["cores"] == c["workers"] >= 1 and c["one_core_value"] > 0     # all-cores figure is measured, not scaled
        src = r["traffic_source"]["how"] if isinstance(r.get("traffic_source"), dict) else r.get("traffic_source")
        assert r["traffic"] is None or "profiles/" in src or src.startswith("measured by this run")
    if tag >= "r06":
        # `frac` is the rate on the bytes MOVED, nothing in the line exceeds the copy ceiling unlabelled, the
        # two-pass accounting has its own name, the kernels carry solo AND in-step durations, configs 3 and 5 ride along
        assert "moved" in r["frac_basis"] and r["frac"] < r["frac_two_pass_accounting"] and r["frac_of_copy_ceiling"] < 1.0
        assert abs(r["frac"] - r["traffic"] / (r["solo_ms_per_launch"] * 1e-3) / 8e12) < 1e-9
        assert d["hbm_frac_of_copy_ceiling"] < 1.0 and d["hbm_two_pass_accounting"]["bytes_per_frame"] > d["hbm_moved_bytes_per_frame"]
        for fam in ("caf_segments", "caf_doppler", "ls_correlate", "ls_fir_subtract"):
            k = d["kernels"][fam]
            assert k["in_step_ms_per_launch"] >= 0.9 * k["avg_ms_per_launch"] and k["moved_bytes_per_launch"] > 0, fam
        acc = d["kernel_time_accounting"]
        assert acc["in_step_mean_concurrency"] > 1.0 and acc["solo_sum_ms_per_step"] > 0
        assert c["cores"] <= c["usable_cores"] <= c["host_cores"] and len(c["legs"]) >= 1
        assert c["value"] == max(l["value"] for l in c["legs"] if "value" in l)
        for leg in ("cfg3", "cfg5"):
            sec = d["secondary"][leg]
            assert sec["value"] > 0 and sec["unit"] == "frames/s" and "roofline" in sec and leg in sec["workload"]


@pytest.mark.parametrize("F", [256, 512, 1024, 2048, 4096])
def test_doppler_column_kernel_phases_on_the_cpu(F):
    """the Doppler column-FFT kernel (passiveradar_amd/csrc/doppler_col.h) run phase by phase, thread by thread on
    the host with its own index algebra, LDS slots and twiddle table (tests/csrc/doppler_emul.cpp): equals
    fftshift(fft(y, axis=0), axes=0) of range_doppler_processing.py:89 for ragged column counts and several frames"""
    import ctypes
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    lib = os.path.join(here, "libdopemul.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", here, "libdopemul.so"])
    h = ctypes.CDLL(lib)
    h.dop_emul.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(F)
    for cols, nf in ((1, 1), (37, 2), (65, 1)):
        y = (rng.standard_normal((nf, F, cols)) + 1j * rng.standard_normal((nf, F, cols))).astype(np.complex64)
        out = np.full_like(y, np.nan)
        assert h.dop_emul(F, y.ctypes.data, out.ctypes.data, cols, nf) == 0
        exp = np.fft.fftshift(np.fft.fft(y.astype(np.complex128), axis=1), axes=1)
        assert np.abs(out - exp).max() / np.abs(exp).max() < 1e-6


@pytest.mark.parametrize("up,dn,n", [(13, 119, 20000), (3, 7, 5000), (1, 4, 4001), (5, 4, 3000), (16, 15, 2000), (7, 1, 900)])
def test_front_end_group_tables_reproduce_the_resampler(up, dn, n):
    """tools/frontend_group_model.py builds the tap rows, LDS offsets (with the pad sample of an even decimation) and
    windows of frontend_group_kernel exactly as the plan and the kernel do; replayed in NumPy they must give
    resample_poly's polyphase sum (the oracle's restatement, itself pinned to the reference's output)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import np_oracle as O
    from tools import frontend_group_model as M
    rng = np.random.default_rng(up * 1000 + dn)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    hp, npr, u, d = O.resample_design(up, dn)
    ref = O.resample(x, up, dn)
    slope = (x[-1] - x[0]) / (n - 1)

    def xe(i):
        i = np.asarray(i)
        out = np.empty(i.shape, dtype=np.complex128)
        lo, hi = i < 0, i >= n
        mid = ~(lo | hi)
        out[lo] = x[0] + slope * i[lo]
        out[hi] = x[-1] + slope * (i[hi] - (n - 1))
        out[mid] = x[i[mid]]
        return out
    for balance in (True, False):        # round 5's banded, cost-balanced runs of rows / rounds 3-4's equal runs at full width
        y = M.run(xe, ref.shape[0], hp, u, d, npr, dtype=np.complex128, balance=balance)
        assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-7      # the taps are float32 in the table
    T, roff, (segs, nq), r_first, lane_stride, pad, span = M.tables(hp, u, d, npr)
    assert pad == (1 if d % 2 == 0 else 0) and lane_stride == d + pad
    # the segments are consecutive runs of trips that cover every row once, their column windows stay inside the
    # accumulators, and no tap lies outside a window
    flat = [sg for w in segs for sg in w]
    assert all(len(w) <= M.SEGS for w in segs) and flat[0][0] == 0
    assert all(flat[i + 1][0] == flat[i][0] + 2 * flat[i][1] for i in range(len(flat) - 1))
    assert flat[-1][0] + 2 * flat[-1][1] == T.shape[0] - 1
    inside = 0.0
    for row0, ntrips, code in flat:
        qa, wd = M.window_of(code, nq)
        assert 0 <= qa and qa + wd <= nq and wd >= 1
        inside += float(np.abs(T[row0:row0 + 2 * ntrips, qa:qa + wd]).sum())
    assert np.isclose(inside, float(np.abs(T).sum()), rtol=1e-6)
    # every tap of every phase sits in exactly one row
    assert np.isclose(T[:, :u].sum(), np.asarray(hp, dtype=np.float32).sum(), rtol=1e-5)
    if (u, d) == (13, 119):              # what the split buys at the published workload's ratio: multiply-add slots per lane
        per_wave = [sum(2 * nt * (c & 0xff) for _, nt, c in w) for w in segs]
        old, _, _, _ = M.split_rows(len(hp), u, d, npr, balance=False)
        per_wave_old = [sum(2 * nt * (c & 0xff) for _, nt, c in w) for w in old]
        print("13:119 segments", segs, "multiply-add slots per wavefront", per_wave, "against", per_wave_old)
        assert max(per_wave) < 0.8 * max(per_wave_old)


def test_zarr_frame_writer_stores_blocks_in_any_order(tmp_path):
    """ZarrFrameWriter (the store bench.py --workload prconfig fills batch by batch from a host thread) writes the same
    store as save_range_doppler_zarr whatever the order of the blocks, and refuses a block that does not fit."""
    from passiveradar_amd import output
    rng = np.random.default_rng(3)
    frames = (rng.standard_normal((7, 6, 5)) + 1j * rng.standard_normal((7, 6, 5))).astype(np.complex64)
    a, b = str(tmp_path / "a.zarr"), str(tmp_path / "b.zarr")
    output.save_range_doppler_zarr(a, frames)
    w = output.ZarrFrameWriter(b, 6, 5, 7)
    for lo, hi in ((4, 7), (0, 1), (1, 4)):
        w.write(lo, frames[lo:hi])
    got, exp = output.read_zarr_v2(b), output.read_zarr_v2(a)
    assert got.shape == (6, 5, 7) and np.array_equal(got, exp) and np.array_equal(got, np.moveaxis(frames, 0, 2))
    for bad_first, blk in ((5, frames[:3]), (-1, frames[:1]), (0, frames[:1].astype(np.complex128)), (0, frames[:1, :5])):
        with pytest.raises(ValueError):
            w.write(bad_first, blk)


def test_cfar_tile_wraparound_indices_advance_like_the_modulo():
    """cfar_sep_kernel (passiveradar_amd/csrc/cfar.hip) loads its LDS tile with wrap-around row / column indices that are
    ADVANCED (ii += 4 % H, one conditional subtraction) instead of divided per element; replayed here for maps smaller and
    larger than the box, tiles on every edge: the advanced index is (start + step k) mod size for every element loaded."""
    TX, TY = 64, 16
    for H, W, fw in ((1024, 177, 18), (5, 7, 18), (3, 200, 5), (100, 3, 6), (16, 65, 8), (70, 90, 31), (1, 1, 3), (64, 64, 2)):
        c = (fw - 1) // 2
        th, tw = TY + fw - 1, TX + fw - 1
        for by in range(-(-H // TY)):
            for bx in range(-(-W // TX)):
                i0, j0 = by * TY, bx * TX
                for tq in range(4):
                    ii = (i0 + c - (fw - 1) + tq) % H            # C's % then "if (ii < 0) ii += H" == Python's %
                    rstep = 4 % H
                    for r in range(tq, th, 4):
                        assert ii == (i0 + c - (fw - 1) + r) % H
                        ii += rstep
                        if ii >= H:
                            ii -= H
                for tx in (0, 1, 31, 63):
                    jj = (j0 + c - (fw - 1) + tx) % W
                    sstep = 64 % W
                    for s_ in range(tx, tw, 64):
                        assert jj == (j0 + c - (fw - 1) + s_) % W
                        jj += sstep
                        if jj >= W:
                            jj -= W


def test_xcorr_of_unequal_lengths_reduces_to_the_equal_length_sum():
    """signal_utils.py:29-32 with len(s1) != len(s2): the drop-in zero-extends both signals so that the equal-length
    entry point (prc_xcorr) computes the reference's 'valid' correlation -- the mapping itself is host logic and is held
    here against the oracle's restatement of SciPy's rule (the oracle against the reference: tests/golden/xcorr_uneven)"""
    from oracle import np_oracle as O
    from passiveradar_amd.signal_utils import _xcorr_equal_lengths
    rng = np.random.default_rng(5)
    for _ in range(300):
        n1, n2 = (int(v) for v in rng.integers(1, 150, 2))
        nlead, nlag = (int(v) for v in rng.integers(0, 40, 2))
        if n1 == n2:
            continue
        s1 = (rng.standard_normal(n1) + 1j * rng.standard_normal(n1)).astype(np.complex64)
        s2 = (rng.standard_normal(n2) + 1j * rng.standard_normal(n2)).astype(np.complex64)
        e1, e2, lead, lag = _xcorr_equal_lengths(s1, s2, nlead, nlag)
        assert e1.shape == e2.shape and lead >= 0 and lag >= 0
        want = O.xcorr(s1, s2, nlead, nlag)
        assert want.shape == (abs(n2 + nlag + nlead - n1) + 1,)
        assert np.array_equal(O.xcorr(e1, e2, lead, lag), want)


def test_zarr_reader_and_writer_against_the_specifications_own_example(tmp_path):
    """VERDICT r4 (f4 by a consumer): no zarr / h5py is installed here, so the nearest thing to a consumer is the zarr v2
    specification's own worked example -- a (20, 20) int32 array in (10, 10) zlib chunks, fill value 42 -- as a committed
    fixture written from the specification's rules with the standard library alone (oracle/gen_zarr_fixture.py).
    read_zarr_v2 must read it, and write_zarr_v2 must emit the same store again: the same file names, the same .zarray
    bytes, the same chunk payloads (the same compressed bytes too when the zlib library is the one that made the
    fixture).  The store of main.py:216-224 goes through the same writer code (zarray_json, chunk naming)."""
    import zlib
    from passiveradar_amd import output
    fx = os.path.join(GOLDEN, "zarr_v2_spec_example")
    want = np.arange(400, dtype="<i4").reshape(20, 20)
    got = output.read_zarr_v2(fx)
    assert got.dtype == np.dtype("<i4") and np.array_equal(got, want)
    meta = json.load(open(os.path.join(fx, ".zarray")))
    out = output.write_zarr_v2(str(tmp_path / "again.zarr"), got, meta)
    names = sorted(n for n in os.listdir(fx) if n != "zlib_version.txt")
    assert sorted(os.listdir(out)) == names == [".zarray", "0.0", "0.1", "1.0", "1.1"]
    assert open(os.path.join(out, ".zarray"), "rb").read() == open(os.path.join(fx, ".zarray"), "rb").read()
    same_zlib = open(os.path.join(fx, "zlib_version.txt")).read().strip() == zlib.ZLIB_RUNTIME_VERSION
    for n in names[1:]:
        a, b = open(os.path.join(out, n), "rb").read(), open(os.path.join(fx, n), "rb").read()
        assert zlib.decompress(a) == zlib.decompress(b)
        if same_zlib:
            assert a == b, n
    # a missing chunk reads as the fill value; an edge chunk is stored at full size
    os.remove(os.path.join(out, "1.1"))
    miss = output.read_zarr_v2(out)
    assert np.all(miss[10:, 10:] == 42) and np.array_equal(miss[:10], want[:10])
    edge = output.write_zarr_v2(str(tmp_path / "edge.zarr"), want[:15, :7], {**meta, "shape": [15, 7], "compressor": None})
    assert os.path.getsize(os.path.join(edge, "1.0")) == 400 and np.array_equal(output.read_zarr_v2(edge), want[:15, :7])
    # the range-Doppler store itself: its .zarray is the same canonical document
    frames = (np.arange(2 * 4 * 3).reshape(2, 4, 3) * (1 + 0.5j)).astype(np.complex64)
    p = output.save_range_doppler_zarr(str(tmp_path / "X.zarr"), frames)
    doc = open(os.path.join(p, ".zarray"), "rb").read()
    assert doc == output.zarray_json(json.loads(doc)) and np.array_equal(output.read_zarr_v2(p), np.moveaxis(frames, 0, 2))


def test_bench_gpus_n_refuses_to_run_on_fewer_gpus():
    """`python bench.py --gpus N` with no launcher around it must start N ranks itself or fail loudly --
    never benchmark one GPU and print `n_gpus: 1`.  On a box with fewer than N GPUs (this container has none): a non-zero
    exit, the reason on stderr, no JSON line.  A launcher whose WORLD_SIZE disagrees with --gpus is refused the same way."""
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except ImportError:
        have = 0
    want = have + 1 if have else 2
    r = subprocess.run([sys.executable, bench, "--gpus", str(max(want, 2))], capture_output=True, text=True, timeout=300, env=clean)
    assert r.returncode != 0
    assert f"--gpus {max(want, 2)} asked for, this host exposes {have} GPU(s)" in r.stderr, r.stderr[-500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(clean, RANK="0", WORLD_SIZE="1"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr, r.stderr[-500:]
