"""The oracle (oracle/np_oracle.py) against every golden vector generated from the reference.
CPU only.  This is what pins the oracle; the GPU parity tests then compare HIP vs oracle."""
import json
import os

import numpy as np
import pytest

from conftest import CAF_SMALL, GOLDEN, golden_window, load_golden, rel_err
from oracle import np_oracle as O
from passiveradar_amd import scene


@pytest.mark.parametrize("name", CAF_SMALL)
def test_caf_small(name):
    g = load_golden("caf_" + name)
    out = O.fast_xambg(g["ref"], g["srv"], int(g["R"]), int(g["F"]), int(g["inputLen"]),
                       golden_window(g), bool(g["shortFilt"]))
    assert out.shape == g["out"].shape and out.dtype == np.complex64
    assert rel_err(out, g["out"]) < 2e-6


def test_caf_libcalls_form_matches():
    g = load_golden("caf_oddq")
    out = O.fast_xambg_libcalls(g["ref"], g["srv"], int(g["R"]), int(g["F"]), g["window"])
    assert rel_err(out, g["out"]) < 1e-6


def test_caf_cfg1_full_size():
    g = load_golden("caf_cfg1")
    from scipy.signal import get_window
    n, R, F = int(g["N"]), int(g["R"]), int(g["F"])
    ref, srv = scene.make_scene(n, float(g["fs"]), R, int(g["seed"]))
    out = O.fast_xambg(ref, srv, R, F, n, get_window(("kaiser", 5.0), n))
    assert rel_err(out, g["out"]) < 2e-6


def test_caf_axis_conventions():
    # echo srv[n] = ref[n-7] e^{+j 2 pi 5 n / Fs}, 1 s CPI -> row F/2-5, column R-7
    n, R, F = 4096, 20, 64
    ref = scene.white_reference(n, 11)
    srv = (np.roll(ref, 7) * np.exp(2j * np.pi * 5 * np.arange(n) / n)).astype(np.complex64)
    X = np.abs(O.fast_xambg(ref, srv, R, F, n, None)[:, :, 0])
    assert np.unravel_index(X.argmax(), X.shape) == scene.expected_peak_cell(7, 5.0, n, n, R, F)
    D = np.abs(O.direct_xambg(ref, srv, R, F, float(n))[:, :, 0])      # not mirrored
    assert np.unravel_index(D.argmax(), D.shape) == (F // 2 + 5, R - 7)


def test_direct_xambg_against_the_reference():
    """range_doppler_processing.py:93-124, golden made by the reference (oracle/gen_golden.py direct_xambg_case)"""
    g = load_golden("direct_xambg")
    for t in "ab":
        out = O.direct_xambg(g[f"ref_{t}"], g[f"srv_{t}"], int(g[f"R_{t}"]), int(g[f"F_{t}"]), float(g[f"fs_{t}"]))
        assert out.shape == g[f"out_{t}"].shape and out.dtype == np.complex64
        assert rel_err(out, g[f"out_{t}"]) < 2e-6
    with pytest.raises(ValueError):
        O.direct_xambg(np.zeros(8, np.complex64), np.zeros(9, np.complex64), 1, 2, 1.0)


def test_caf_shape_error():
    with pytest.raises(ValueError):
        O.fast_xambg(np.zeros(8, np.complex64), np.zeros(9, np.complex64), 1, 2)


def test_xcorr():
    g = load_golden("xcorr")
    for key, (nl, ng) in {"z_0_20": (0, 20), "z_7_0": (7, 0), "z_3_9": (3, 9)}.items():
        z = O.xcorr(g["s1"], g["s2"], nl, ng)
        assert z.dtype == np.complex64 and rel_err(z, g[key]) < 2e-6
    assert rel_err(O.xcorr(g["s1"], g["s1"], 0, 15), g["z_auto"]) < 2e-6


def test_xcorr_of_unequal_lengths():
    """signal_utils.py:29-32 takes any two lengths ('valid' correlation of s1 with the padded s2, whichever is longer):
    |n1 - (n2 + nlag + nlead)| + 1 values, reference-made golden"""
    g = load_golden("xcorr_uneven")
    for i, (n1, n2, nlead, nlag) in enumerate(g["cases"]):
        z = O.xcorr(g["s1"][:n1], g["s2"][:n2], int(nlead), int(nlag))
        want = g[f"z{i}"]
        assert z.shape == want.shape == (abs(n2 + nlag + nlead - n1) + 1,) and z.dtype == np.complex64
        assert np.abs(z - want).max() < 2e-6 * np.abs(g[f"z{0}"]).max(), i


def test_frequency_shift_with_a_phase_per_sample():
    """signal_utils.py:24-27 with an array phase_offset of the signal's length: complex128 for float64 / integer
    phases, complex64 for float32 ones (NumPy's promotion), reference-made golden"""
    g = load_golden("freqshift_phases")
    n, fs, st = int(g["n"]), float(g["fs"]), int(g["stride"])
    x, _ = scene.make_scene(n, fs, 8, int(g["seed"]))
    for key, fc, ph, dt in (("y64", 37500.5, g["ph64"], np.complex128), ("y32", 37500.5, g["ph64"].astype(np.float32), np.complex64),
                            ("yi", -12.25, g["phi"], np.complex128), ("y1_32", 80.0, np.array([0.3], np.float32), np.complex64)):
        y = O.frequency_shift(x, fc, fs, ph)
        assert y.dtype == dt and y.dtype == g[key].dtype, key
        assert np.abs(y[::st] - g[key]).max() < 2e-6, key


def test_frequency_shift():
    g = load_golden("freqshift")
    x, _ = scene.make_scene(int(g["n"]), float(g["fs"]), 8, int(g["seed"]))
    assert np.array_equal(x[:64], g["x_head"])          # generator is deterministic
    st = int(g["stride"])
    fs = float(g["fs"])
    for key, fc, ph in (("y_p1", 1, 0), ("y_m2", -2, 0), ("y_f", 37.5, 0), ("y_ph", 80.0, 0.3)):
        y = O.frequency_shift(x, fc, fs, ph)
        assert y.dtype == np.complex64
        assert np.abs(y[::st] - g[key]).max() < 1e-6


@pytest.mark.parametrize("name", ["ls_toeplitz_white", "ls_toeplitz_peek0", "ls_toeplitz_coloured"])
def test_ls_toeplitz(name):
    g = load_golden(name)
    out, taps = O.LS_Filter_Toeplitz(g["ref"], g["srv"], int(g["L"]), int(g["peek"]), True)
    assert out.dtype == np.complex128 and taps.dtype == np.complex128
    assert rel_err(taps, g["taps"]) < 5e-6
    assert rel_err(out, g["out"]) < 5e-6


def test_levinson_against_scipy():
    from scipy.linalg import solve_toeplitz
    rng = np.random.default_rng(5)
    c = rng.standard_normal(40) + 1j * rng.standard_normal(40)
    c[0] = 12.0
    b = rng.standard_normal(40) + 1j * rng.standard_normal(40)
    assert rel_err(O.levinson_hermitian(c, b), solve_toeplitz(c, b)) < 1e-12


def test_ls_multiple():
    g = load_golden("ls_multiple")
    out = O.LS_Filter_Multiple(g["ref"], g["srv"], int(g["L"]), float(g["fs"]), list(g["bins"]))
    assert rel_err(out, g["out"]) < 5e-6


@pytest.mark.parametrize("name", ["ls_direct", "ls_direct_reg"])
def test_ls_direct(name):
    g = load_golden(name)
    out, taps = O.LS_Filter(g["ref"], g["srv"], int(g["L"]), float(g["reg"]), int(g["peek"]), True)
    assert out.dtype == np.complex64
    assert rel_err(taps, g["taps"]) < 2e-5          # the reference solves in complex64 (LAPACK)
    assert rel_err(out, g["out"]) < 5e-5


def test_nlms():
    g = load_golden("nlms")
    out, taps = O.NLMS_filter(g["ref"], g["srv"], int(g["L"]), float(g["mu"]), int(g["peek"]), None, True)
    assert rel_err(out, g["out"]) < 1e-5 and rel_err(taps, g["taps"]) < 1e-5
    gw = load_golden("nlms_warm")
    out, taps = O.NLMS_filter(gw["ref"], gw["srv"], 999, float(gw["mu"]), 10, gw["initialTaps"], True)
    assert rel_err(out, gw["out"]) < 1e-5 and rel_err(taps, gw["taps"]) < 1e-5
    g7 = load_golden("nlms_t74")
    out, taps = O.NLMS_filter(g7["ref"], g7["srv"], int(g7["L"]), float(g7["mu"]), int(g7["peek"]), None, True)
    assert rel_err(out, g7["out"]) < 1e-5 and rel_err(taps, g7["taps"]) < 1e-5
    # support of the output (:231): zero before L and from n-peek on
    L, pk = int(g["L"]), int(g["peek"])
    assert not out[:0].any() and not g["out"][:L].any() and not g["out"][-pk:].any()


@pytest.mark.parametrize("T", [2110, 4110])
def test_nlms_long_filters_vs_reference(T):
    """the reference's own NLMS_filter at filter lengths beyond one wavefront's 2048 taps (inputs from the seed): both
    restatements -- they are what the GPU kernels with two / four wavefronts per stream are checked against"""
    from oracle import c_oracle
    from passiveradar_amd import scene
    g = load_golden(f"nlms_t{T}")
    a, s = scene.make_scene(int(g["N"]), float(g["fs"]), int(g["scene_R"]), int(g["seed"]))
    L, mu, pk = int(g["L"]), float(g["mu"]), int(g["peek"])
    out, taps = O.NLMS_filter(a, s, L, mu, pk, None, True)
    assert rel_err(out, g["out"]) < 1e-5 and rel_err(taps, g["taps"]) < 1e-5
    out, taps = c_oracle.nlms(a, s, L, mu, pk)
    assert rel_err(out, g["out"]) < 1e-5 and rel_err(taps, g["taps"]) < 1e-5


def test_stream_pipeline():
    g = load_golden("stream")
    C, R, F = int(g["C"]), int(g["R"]), int(g["F"])
    out = O.process_stream(g["ref"], g["srv"], 2 * C, R, F, float(g["fs"]))
    assert out.shape == g["out"].shape == (F, R + 1, 6)
    assert rel_err(out, g["out"]) < 1e-5


def test_c_twin_nlms_and_caf():
    from oracle import c_oracle
    g = load_golden("nlms")
    out, taps = c_oracle.nlms(g["ref"], g["srv"], int(g["L"]), float(g["mu"]), int(g["peek"]))
    assert rel_err(out, g["out"]) < 1e-5 and rel_err(taps, g["taps"]) < 1e-5
    gw = load_golden("nlms_warm")
    out, taps = c_oracle.nlms(gw["ref"], gw["srv"], 999, float(gw["mu"]), 10, gw["initialTaps"])
    assert rel_err(out, gw["out"]) < 1e-5 and rel_err(taps, gw["taps"]) < 1e-5
    for name in ("p2", "oddq", "nondiv", "lags_gt_q", "bigq"):
        g = load_golden("caf_" + name)
        w = golden_window(g)
        X = c_oracle.fast_xambg(g["ref"], g["srv"], int(g["R"]), int(g["F"]),
                                None if w is None else w)
        assert rel_err(X, g["out"]) < 2e-6, name


def test_ls_libcalls_form_matches():
    g = load_golden("ls_multiple")
    out = O.LS_Filter_Multiple_libcalls(g["ref"], g["srv"], int(g["L"]), float(g["fs"]), list(g["bins"]))
    assert rel_err(out, g["out"]) < 1e-7


def test_front_end_restatement():
    """deinterleave / block-phase tuning / polyphase 'line' resampler."""
    g = load_golden("frontend")
    assert np.array_equal(O.deinterleave_IQ(g["rawf"]), g["deintf"])
    icl = int(g["icl"])
    assert np.array_equal(O.deinterleave_IQ(g["raw8"][icl:2 * icl]), g["deint1"])
    tuned = O.frequency_shift(g["deint1"], int(g["foff"]), int(g["fs"]), np.array([g["phases"][1]]))
    assert tuned.dtype == np.complex128 and rel_err(tuned, g["tuned1"]) < 1e-12
    out = O.front_end(g["raw8"], icl, int(g["foff"]), int(g["fs"]), int(g["up"]), int(g["dn"]))
    assert out.dtype == np.complex128 and out.shape == g["out"].shape
    assert rel_err(out, g["out"]) < 1e-12
    assert rel_err(O.resample(g["deint1"], 3, 7), g["res_c64"]) < 2e-6
    assert rel_err(O.resample(g["deint1"][:5000], 13, 119), g["res_simple"]) < 2e-6


def test_cfar_restatement():
    """wrap-around box filter with the reference's asymmetric guard block."""
    g = load_golden("cfar")
    assert rel_err(O.CFAR_2D(g["X"], 18, 4), g["cr_18_4"]) < 1e-12
    assert rel_err(O.CFAR_2D(g["X"], 7, 2), g["cr_7_2"]) < 1e-12
    det = O.CFAR_2D(g["X"], 18, 4, float(g["thr"]))
    assert det.dtype == bool and np.array_equal(det, g["det_18_4"]) and 5 < det.sum() < 30


def test_channel_offset_restatement():
    """signal.decimate's zero-phase IIR restated (design, steady state, odd extension,
    recursion) and the 'valid' correlation's index convention, against the reference's outputs."""
    g = load_golden("channel_offset")
    for tag in g["cases"]:
        s1, s2, nd, nl = g[f"{tag}_s1"], g[f"{tag}_s2"], int(g[f"{tag}_nd"]), int(g[f"{tag}_nl"])
        B1 = O.decimate_iir(s1, nd)
        # the reference's recursion runs in complex64: a few 1e-6 of rounding noise against the double oracle
        assert rel_err(B1[:1500], g[f"{tag}_B1_head"]) < 2e-5 and rel_err(B1[-1500:], g[f"{tag}_B1_tail"]) < 2e-5
        off, xc = O.find_channel_offset(s1, s2, nd, nl, return_xc=True)
        assert off == int(g[f"{tag}_offset"]), tag
        assert xc.shape == g[f"{tag}_xc"].shape and rel_err(xc, g[f"{tag}_xc"]) < 2e-5
        assert O.find_channel_offset_libcalls(s1, s2, nd, nl) == off
    with pytest.raises(ValueError):
        O.decimate_iir(np.zeros(27, np.complex64), 1)


def test_python_sosfilt_loop_matches_c_twin():
    from oracle import c_oracle
    sos = O.cheby1_lowpass_sos(8, 0.05, 0.2)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(300) + 1j * rng.standard_normal(300)
    zi = O.sosfilt_zi(sos) * x[0]
    yc, zc = c_oracle.sosfilt(sos, x, zi)
    import scipy.signal as sg
    ys, zs = sg.sosfilt(sos, x, zi=zi)
    assert np.abs(yc - ys).max() < 1e-12 and np.abs(zc - zs).max() < 1e-12


def test_cfg2_pipeline_full_size():
    """BASELINE config 2 at full size against the reference's own output (one of the three 1.2 M-sample hop chunks,
    to keep the CPU suite short): the library-call form of the oracle (float32 SciPy correlate, like the reference)
    reproduces the cleaned stream; the closed form (float64 sums) shows how much of it is the reference's own
    float32 noise -- 2e-5 of the stream, which becomes 2.3e-4 of the map's peak on the cancelled direct-path ridge
    (tests/test_gpu_stream.py::test_cfg2_pipeline_against_reference_output holds the map numbers)."""
    g = load_golden("pipeline_cfg2")
    n, R, fs = int(g["N"]), int(g["R"]), float(g["fs"])
    C = n // 2
    a, s = scene.make_stream(3, C, fs, R, int(g["seed"]))
    bins = [0, 1, -1, 2, -2]
    i = 1
    idx = np.arange(0, 3 * C, 101)
    sel = idx[(idx >= i * C) & (idx < (i + 1) * C)]
    want = g["cleaned_sub"][sel // 101]
    scale = np.abs(g["cleaned_sub"]).max()                            # same normalisation as the whole-stream figure
    lib = O.LS_Filter_Multiple_libcalls(a[i * C:(i + 1) * C], s[i * C:(i + 1) * C], R, fs, bins)
    assert np.abs(lib[sel - i * C] - want).max() / scale < 1e-6
    exact = O.LS_Filter_Multiple(a[i * C:(i + 1) * C], s[i * C:(i + 1) * C], R, fs, bins)
    assert 2e-6 < np.abs(exact[sel - i * C] - want).max() / scale < 1e-4   # the reference's float32 accumulation
    # ... and that noise is the reference's own: fed complex128 inputs (its correlations then sum in double),
    # the reference lands on the closed form to 1e-7 (tests/golden/pipeline_cfg2_c128.npz)
    g2 = load_golden("pipeline_cfg2_c128")
    assert np.abs(exact[sel - i * C] - g2["cleaned_sub"][sel // 101]).max() / scale < 5e-7
    d = np.abs(g["out"] - g2["out"]) / np.abs(g2["out"]).max()        # the two reference maps differ on the ridge only
    off = np.ones(d.shape[0], bool)
    off[d.shape[0] // 2 - 1:d.shape[0] // 2 + 2] = False
    assert d[off].max() < 1e-4 < d.max() < 5e-4


def test_ls_cfg1_chunk():
    """config 1's LS stage at full chunk size (131 072 samples, T = 266, five bins) against the reference's output"""
    g = load_golden("ls_cfg1")
    n, R, fs = int(g["N"]), int(g["R"]), float(g["fs"])
    a, s = scene.make_scene(n, fs, R, int(g["seed"]))
    out = O.LS_Filter_Multiple(a, s, R, fs, [0, 1, -1, 2, -2])
    scale = np.abs(g["head"]).max()
    assert np.abs(out[::7] - g["out_sub"]).max() / scale < 1e-5
    assert np.abs(out[:600] - g["head"]).max() / scale < 1e-5 and np.abs(out[-600:] - g["tail"]).max() / scale < 1e-5


def test_cfg1_pipeline_restatement():
    """config 1 end to end: the oracle's main.py:169-194 restatement against the reference's own frame"""
    g = load_golden("pipeline_cfg1")
    n, R, F, fs = int(g["N"]), int(g["R"]), int(g["F"]), float(g["fs"])
    a, s = scene.make_stream(3, n // 2, fs, R, int(g["seed"]))
    frames = O.process_stream(a, s, n, R, F, fs)
    assert rel_err(frames[:, :, int(g["frame_index"])], g["out"]) < 1e-4


@pytest.mark.parametrize("name", ["ls_multiple_bin50", "ls_multiple_kHz"])
def test_ls_multiple_far_doppler_bins(name):
    """Doppler bins far from zero (the reference takes any list, clutter_removal.py:178-187)"""
    g = load_golden(name)
    out = O.LS_Filter_Multiple(g["ref"], g["srv"], int(g["L"]), float(g["fs"]), [float(b) for b in g["bins"]])
    assert rel_err(out, g["out"]) < 2e-5


def test_ls_at_the_config3_tap_count():
    """T = 1034 (config 3's filter length): Toeplitz, Multiple and the circular direct form against the reference"""
    g = load_golden("ls_t1034")
    n, L, fs = int(g["N"]), int(g["L"]), float(g["fs"])
    a, s = scene.make_scene(n, fs, L, int(g["seed"]))
    out, taps = O.LS_Filter_Toeplitz(a, s, L, return_filter=True)
    assert rel_err(taps, g["taps"]) < 2e-5 and rel_err(out, g["out"]) < 2e-5
    assert rel_err(O.LS_Filter_Multiple(a, s, L, fs, [float(b) for b in g["bins"]]), g["out_multiple"]) < 2e-5
    g = load_golden("ls_direct_t1034")
    a, s = scene.make_scene(int(g["N"]), float(g["fs"]), L, int(g["seed"]))
    out, taps = O.LS_Filter(a, s, L, return_filter=True)
    # the reference solves and applies this one in complex64 (1034-term float32 dot products, no block edges: the
    # residual that normalises the error is only ~0.05 of the input): taps to 1e-6, output to 1e-3 of its own peak
    assert rel_err(taps, g["taps"]) < 5e-6 and rel_err(out, g["out"]) < 1e-3


def test_nlms_cfg3_hop_c_twin_vs_reference_digest():
    """config 3's NLMS stage at FULL hop length (2.5 M samples, T = 1034, mu = 0.02): the C twin of the oracle against
    the digest the reference's own NLMS_filter produced (clutter_removal.py:189-249, oracle/gen_golden.py
    nlms_cfg3_digest_case) -- pins the checker the GPU test leans on at the size it is used"""
    from oracle import c_oracle
    g = load_golden("nlms_cfg3_digest")
    n, L = int(g["N"]), int(g["L"])
    ref, srv = scene.make_scene(n, float(g["fs"]), L, int(g["seed"]))
    out, taps = c_oracle.nlms(ref, srv, L, float(g["mu"]), int(g["peek"]))
    peak = float(g["peak"])
    assert np.abs(out[::997] - g["sub"]).max() / peak < 1e-4
    assert np.abs(out[:4096] - g["head"]).max() / peak < 1e-4
    assert np.abs(out[-4096:] - g["tail"]).max() / peak < 1e-4
    assert rel_err(taps, g["taps"]) < 1e-4
    assert abs(float(np.vdot(out, out).real) / float(g["energy"]) - 1) < 1e-4


def _prconfig_raw(g):
    """the golden's raw int8 recordings, regenerated from its seed and checked against its checksums"""
    icl, fs_in, foff = int(g["cfg_input_chunk_length"]), int(g["cfg_input_sample_rate"]), int(g["cfg_offset_freq"])
    raw_ref, raw_srv = scene.make_raw_stream(int(g["nblk"]), icl, fs_in, foff, int(g["seed"]))
    if (list(scene.raw_checksum(raw_ref)) != [int(v) for v in g["raw_ref_checksum"]] or
            list(scene.raw_checksum(raw_srv)) != [int(v) for v in g["raw_srv_checksum"]]):
        pytest.skip("the regenerated raw stream differs from the golden's (another libm / NumPy build)")
    return raw_ref, raw_srv


def test_prconfig_raw_to_frame_restatement():
    """The reference's published workload as shipped (PRconfig.yaml: raw int8 -> tune -> 13:119 -> LS x5, T = 185 ->
    1024 x 176 CAF, main.py:105-194): the oracle's restatement against the reference's own IF streams, cleaned stream
    and middle frame, at full size"""
    g = load_golden("pipeline_prconfig_raw")
    raw_ref, raw_srv = _prconfig_raw(g)
    icl, fs_in, foff = int(g["cfg_input_chunk_length"]), int(g["cfg_input_sample_rate"]), int(g["cfg_offset_freq"])
    up, dn, n = int(g["cfg_resamp_up"]), int(g["cfg_resamp_dn"]), int(g["cfg_cpi_samples"])
    R, F, fs = int(g["cfg_num_range_cells"]), int(g["cfg_num_doppler_cells"]), float(g["cfg_IF_sample_rate"])
    a = O.front_end(raw_ref, icl, foff, fs_in, up, dn)
    s = O.front_end(raw_srv, icl, foff, fs_in, up, dn)
    assert a.shape[0] == int(g["nblk"]) * int(g["cfg_output_chunk_length"])
    assert rel_err(a[::61], g["if_ref_sub"]) < 1e-5 and rel_err(s[::61], g["if_srv_sub"]) < 1e-5
    frames, cleaned = O.process_stream(a, s, n, R, F, fs, return_cleaned=True)
    assert np.abs(cleaned[::61] - g["cleaned_sub"]).max() / float(g["if_srv_rms"]) < 1e-4
    assert rel_err(frames[:, :, int(g["frame_index"])], g["out"]) < 1e-4


def test_caf_long_filter_at_config1_size():
    """shortFilt=False at config-1 size (10 241-tap flat-top decimation FIR): the oracle against the reference's surface"""
    from scipy.signal import get_window
    g = load_golden("caf_longfilt_cfg1")
    n, R, F = int(g["N"]), int(g["R"]), int(g["F"])
    ref, srv = scene.make_scene(n, float(g["fs"]), R, int(g["seed"]))
    out = O.fast_xambg(ref, srv, R, F, n, get_window(("kaiser", 5.0), n), shortFilt=False)[:, :, 0]
    assert rel_err(out, g["out"]) < 1e-5
