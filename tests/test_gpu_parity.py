"""Parity of the HIP path (through the C ABI, via the drop-in Python functions) against the
golden vectors generated from the reference and against the pinned oracle.  Needs an MI355X.

Tolerance: north_star states float32 parity as max|X_gpu - X_ref| / max|X_ref| < 1e-4 per frame
on white-reference scenes; TOL below is that bar.  The tighter TIGHT bar documents what the
kernels actually achieve (fp32 accumulation of <= ~5000-term sums).
"""
import os
import numpy as np
import pytest

from conftest import CAF_SMALL, golden_window, load_golden, rel_err
from oracle import np_oracle as O
from passiveradar_amd import scene

pytestmark = pytest.mark.gpu

TOL = 1e-4
TIGHT = 2e-5


@pytest.fixture(autouse=True)
def _gpu(gpu_ready):
    yield


@pytest.fixture(params=["direct", "fft"])
def caf_method(request):
    """run every CAF case through both segment kernels (time-domain LDS tiles / wavefront FFT)"""
    from passiveradar_amd import range_doppler_processing as rdp
    rdp.set_default_methods(caf={"direct": 1, "fft": 2}[request.param])
    yield request.param
    rdp.set_default_methods(caf=0)


@pytest.mark.parametrize("name", CAF_SMALL)
def test_caf_golden(name, caf_method):
    from passiveradar_amd.range_doppler_processing import fast_xambg
    g = load_golden("caf_" + name)
    out = fast_xambg(g["ref"], g["srv"], int(g["R"]), int(g["F"]), int(g["inputLen"]),
                     golden_window(g), bool(g["shortFilt"]))
    assert out.shape == g["out"].shape and out.dtype == np.complex64
    e = rel_err(out, g["out"])
    assert e < TIGHT, e


@pytest.mark.parametrize("name,cfg", [("caf_cfg1", 1), ("caf_cfg2", 2)])
def test_caf_full_size_golden(name, cfg, caf_method):
    from scipy.signal import get_window
    from passiveradar_amd.range_doppler_processing import fast_xambg
    g = load_golden(name)
    n, R, F = int(g["N"]), int(g["R"]), int(g["F"])
    ref, srv = scene.make_scene(n, float(g["fs"]), R, int(g["seed"]))
    out = fast_xambg(ref, srv, R, F, n, get_window(("kaiser", 5.0), n))
    e = rel_err(out, g["out"])
    assert e < TOL, e
    # injected targets land in the same cells as in the reference surface
    ref_mag, mag = np.abs(g["out"][:, :, 0]), np.abs(out[:, :, 0])
    for d, fd, _ in scene.default_targets(R):
        r, c = scene.expected_peak_cell(d, fd, n, float(g["fs"]), R, F)
        win = (slice(max(r - 2, 0), r + 3), slice(max(c - 2, 0), c + 3))
        assert np.unravel_index(ref_mag[win].argmax(), ref_mag[win].shape) == \
            np.unravel_index(mag[win].argmax(), mag[win].shape)


def test_caf_cfg3_digest():
    from scipy.signal import get_window
    from passiveradar_amd.range_doppler_processing import fast_xambg
    g = load_golden("caf_cfg3_digest")
    n, R, F = int(g["N"]), int(g["R"]), int(g["F"])
    ref, srv = scene.make_scene(n, float(g["fs"]), R, int(g["seed"]))
    out = fast_xambg(ref, srv, R, F, n, get_window(("kaiser", 5.0), n))[:, :, 0]
    peak = float(g["peak"])
    assert np.abs(out[::8, ::8] - g["sub"]).max() / peak < TOL
    assert np.abs(out.ravel()[g["top_idx"]] - g["top_val"]).max() / peak < TOL
    assert np.abs(out.sum(axis=0) - g["col_sums"]).max() / (peak * np.sqrt(F)) < TOL
    assert np.abs(out.sum(axis=1) - g["row_sums"]).max() / (peak * np.sqrt(R + 1)) < TOL


@pytest.fixture(params=["direct", "fft", "fft_cached", "fft4096_cached"])
def ls_method(request):
    """run every LS case through every kernel family (time-domain LDS tiles / wavefront FFT with the
    reference spectra recomputed per bin / kept in an HBM cache / the cached chain on 4096-point team transforms)"""
    from passiveradar_amd import clutter_removal as cr
    cr.set_default_ls_method({"direct": 1, "fft": 2, "fft_cached": 3, "fft4096_cached": 4}[request.param])
    yield request.param
    cr.set_default_ls_method(0)


@pytest.mark.parametrize("n,R,F,win", [(8192, 70, 128, True), (5000, 4, 51, False), (3000, 129, 8, True),
                                       (65536, 300, 64, True)])
def test_caf_vs_oracle_shapes(n, R, F, win, caf_method):
    """lag spans that are not multiples of 64, non-power-of-two Doppler bins, R > q."""
    from passiveradar_amd.range_doppler_processing import fast_xambg
    ref, srv = scene.make_scene(n, 1e4, R, 4242 + n)
    w = np.kaiser(n, 5.0) if win else None
    exp = O.fast_xambg(ref, srv, R, F, n, w)
    out = fast_xambg(ref, srv, R, F, n, w)
    assert rel_err(out, exp) < TIGHT


def test_caf_linearity_and_errors(caf_method):
    from passiveradar_amd.range_doppler_processing import fast_xambg
    n, R, F = 16384, 20, 64
    ref, s1 = scene.make_scene(n, 1e4, R, 77)
    _, s2 = scene.make_scene(n, 1e4, R, 78)
    a = fast_xambg(ref, s1, R, F)
    b = fast_xambg(ref, s2, R, F)
    ab = fast_xambg(ref, (s1 + 2 * s2).astype(np.complex64), R, F)
    assert rel_err(ab, a + 2 * b) < TIGHT                       # linear in srv
    with pytest.raises(ValueError):
        fast_xambg(ref, s1[:-1], R, F)
    with pytest.raises(ValueError):
        fast_xambg(ref, s1, R, F, inputLen=n - 1)


def test_xcorr_and_freqshift():
    from passiveradar_amd.signal_utils import frequency_shift, xcorr
    g = load_golden("xcorr")
    for key, (nl, ng) in {"z_0_20": (0, 20), "z_7_0": (7, 0), "z_3_9": (3, 9)}.items():
        z = xcorr(g["s1"], g["s2"], nl, ng)
        assert z.dtype == np.complex64 and rel_err(z, g[key]) < TIGHT
    assert rel_err(xcorr(g["s1"], g["s1"], 0, 15), g["z_auto"]) < TIGHT
    gf = load_golden("freqshift")
    x, _ = scene.make_scene(int(gf["n"]), float(gf["fs"]), 8, int(gf["seed"]))
    st, fs = int(gf["stride"]), float(gf["fs"])
    for key, fc, ph in (("y_p1", 1, 0), ("y_m2", -2, 0), ("y_f", 37.5, 0), ("y_ph", 80.0, 0.3)):
        y = frequency_shift(x, fc, fs, ph)
        assert np.abs(y[::st] - gf[key]).max() < 2e-6


def test_direct_xambg_drop_in():
    """a user who swaps the import line finds direct_xambg (range_doppler_processing.py:93-124) too -- a
    composition of prc_frequency_shift + prc_xcorr on device buffers, against the golden made by the reference; same
    shape, dtype, axis convention (rows NOT mirrored) and ValueError as the reference."""
    from passiveradar_amd.range_doppler_processing import direct_xambg
    g = load_golden("direct_xambg")
    for t in "ab":
        out = direct_xambg(g[f"ref_{t}"], g[f"srv_{t}"], int(g[f"R_{t}"]), int(g[f"F_{t}"]), float(g[f"fs_{t}"]))
        want = g[f"out_{t}"]
        assert out.shape == want.shape and out.dtype == np.complex64
        assert rel_err(out, want) < TIGHT, t
        assert np.unravel_index(np.abs(out).argmax(), out.shape) == np.unravel_index(np.abs(want).argmax(), want.shape)
    # an echo at delay 7, Doppler +5 bins peaks at row F/2 + 5 (fast_xambg: F/2 - 5), column R - 7
    n, R, F = 4096, 20, 64
    ref = scene.white_reference(n, 11)
    srv = (np.roll(ref, 7) * np.exp(2j * np.pi * 5 * np.arange(n) / n)).astype(np.complex64)
    D = np.abs(direct_xambg(ref, srv, R, F, float(n))[:, :, 0])
    assert np.unravel_index(D.argmax(), D.shape) == (F // 2 + 5, R - 7)
    with pytest.raises(ValueError, match="same length"):
        direct_xambg(ref, srv[:-1], R, F, float(n))


def test_xcorr_of_unequal_lengths_and_phase_arrays():
    """argument forms the reference's expressions accept -- xcorr of two signals of different lengths
    (signal_utils.py:29-32) and frequency_shift with one phase per sample (signal_utils.py:24-27), against goldens made
    by the reference; the error conventions of NumPy for what cannot broadcast"""
    from passiveradar_amd.signal_utils import frequency_shift, xcorr
    g = load_golden("xcorr_uneven")
    scale = np.abs(g["z0"]).max()
    for i, (n1, n2, nlead, nlag) in enumerate(g["cases"]):
        z = xcorr(g["s1"][:n1], g["s2"][:n2], int(nlead), int(nlag))
        want = g[f"z{i}"]
        assert z.shape == want.shape and z.dtype == np.complex64
        assert np.abs(z - want).max() < 2e-6 * scale, i
    gf = load_golden("freqshift_phases")
    n, fs, st = int(gf["n"]), float(gf["fs"]), int(gf["stride"])
    x, _ = scene.make_scene(n, fs, 8, int(gf["seed"]))
    for key, fc, ph in (("y64", 37500.5, gf["ph64"]), ("y32", 37500.5, gf["ph64"].astype(np.float32)), ("yi", -12.25, gf["phi"]),
                        ("y1_32", 80.0, np.array([0.3], np.float32))):
        y = frequency_shift(x, fc, fs, ph)
        assert y.dtype == gf[key].dtype, key
        assert np.abs(y[::st] - gf[key]).max() < 2e-6, key
    with pytest.raises(ValueError):
        frequency_shift(x, 1.0, fs, np.zeros(7))
    with pytest.raises(ValueError):
        xcorr(x[:100], x[:90], -1, 3)


@pytest.mark.parametrize("name", ["ls_toeplitz_white", "ls_toeplitz_peek0", "ls_toeplitz_coloured"])
def test_ls_toeplitz_golden(name, ls_method):
    from passiveradar_amd.clutter_removal import LS_Filter_Toeplitz
    g = load_golden(name)
    out, taps = LS_Filter_Toeplitz(g["ref"], g["srv"], int(g["L"]), int(g["peek"]), True)
    assert out.dtype == np.complex128 and taps.dtype == np.complex128
    assert rel_err(taps, g["taps"]) < TIGHT
    assert rel_err(out, g["out"]) < TIGHT


def test_ls_multiple_golden(ls_method):
    from passiveradar_amd.clutter_removal import LS_Filter_Multiple
    g = load_golden("ls_multiple")
    out = LS_Filter_Multiple(g["ref"], g["srv"], int(g["L"]), float(g["fs"]), list(g["bins"]))
    assert rel_err(out, g["out"]) < TIGHT


@pytest.mark.parametrize("name", ["ls_direct", "ls_direct_reg"])
def test_ls_direct_golden(name, ls_method):
    from passiveradar_amd.clutter_removal import LS_Filter
    g = load_golden(name)
    out, taps = LS_Filter(g["ref"], g["srv"], int(g["L"]), float(g["reg"]), int(g["peek"]), True)
    assert out.dtype == np.complex64 and taps.dtype == np.complex64
    assert rel_err(taps, g["taps"]) < 5e-5
    assert rel_err(out, g["out"]) < TOL


def test_ls_large_vs_oracle(ls_method):
    """multi-tile blocks, T=266 (config-2 taps), five Doppler bins."""
    from passiveradar_amd.clutter_removal import LS_Filter_Multiple, LS_Filter_Toeplitz
    n, L = 150000, 256
    ref, srv = scene.make_scene(n, 2.4e6, L, 31337)
    exp, etaps = O.LS_Filter_Toeplitz(ref, srv, L, 10, True)
    out, taps = LS_Filter_Toeplitz(ref, srv, L, 10, True)
    assert rel_err(taps, etaps) < TIGHT and rel_err(out, exp) < TIGHT
    expm = O.LS_Filter_Multiple(ref, srv, L, 2.4e6, [0, 1, -1, 2, -2])
    outm = LS_Filter_Multiple(ref, srv, L, 2.4e6, [0, 1, -1, 2, -2])
    assert rel_err(outm, expm) < TOL
    # the canceller actually cancels: direct path + clutter are >= 30 dB down
    assert np.mean(np.abs(outm[2000:-2000]) ** 2) < 1e-3 * np.mean(np.abs(srv) ** 2)


def test_nlms_golden():
    from passiveradar_amd.clutter_removal import NLMS_filter
    g = load_golden("nlms")
    out, taps = NLMS_filter(g["ref"], g["srv"], int(g["L"]), float(g["mu"]), int(g["peek"]), None, True)
    assert out.dtype == np.complex64
    assert rel_err(out, g["out"]) < TIGHT and rel_err(taps, g["taps"]) < TIGHT
    L, pk = int(g["L"]), int(g["peek"])
    assert not out[:L].any() and not out[-pk:].any()
    gw = load_golden("nlms_warm")
    out, taps = NLMS_filter(gw["ref"], gw["srv"], 999, float(gw["mu"]), 10, gw["initialTaps"], True)
    assert rel_err(out, gw["out"]) < TIGHT and rel_err(taps, gw["taps"]) < TIGHT
    g7 = load_golden("nlms_t74")
    out, taps = NLMS_filter(g7["ref"], g7["srv"], int(g7["L"]), float(g7["mu"]), int(g7["peek"]), None, True)
    assert rel_err(out, g7["out"]) < TIGHT and rel_err(taps, g7["taps"]) < TIGHT


def test_nlms_long_filter_vs_c_oracle():
    """T = 1034 (config-3 taps) over several staged windows, checked against the C twin of the oracle."""
    from oracle import c_oracle
    from passiveradar_amd.clutter_removal import NLMS_filter
    n, L = 12000, 1024
    ref, srv = scene.make_scene(n, 1e7, L, 999)
    exp, etaps = c_oracle.nlms(ref, srv, L, 0.02, 10)
    out, taps = NLMS_filter(ref, srv, L, 0.02, 10, None, True)
    assert rel_err(out, exp) < TOL and rel_err(taps, etaps) < TOL


@pytest.mark.parametrize("n,R,F", [(600, 0, 4), (1000, 5, 1), (2047, 9, 7), (2048, 1, 2), (5003, 1000, 3)])
def test_caf_edge_shapes(n, R, F):
    """degenerate spans: one lag column, one Doppler bin, CPI shorter than one FFT piece,
    range span close to the CPI length (several lag blocks, wrap inside every piece)"""
    from passiveradar_amd.range_doppler_processing import fast_xambg
    ref, srv = scene.make_scene(n, 1e4, max(R, 1), 5150 + n)
    exp = O.fast_xambg(ref, srv, R, F, n, None)
    out = fast_xambg(ref, srv, R, F, n, None)
    assert out.shape == (F, R + 1, 1)
    assert rel_err(out, exp) < TIGHT


def test_ls_and_nlms_edge_shapes():
    from passiveradar_amd.clutter_removal import LS_Filter_Multiple, LS_Filter_Toeplitz, NLMS_filter
    ref, srv = scene.make_scene(700, 1e4, 8, 616)
    out, taps = LS_Filter_Toeplitz(ref, srv, 3, 0, True)               # no non-causal taps, tiny filter
    exp, etaps = O.LS_Filter_Toeplitz(ref, srv, 3, 0, True)
    assert rel_err(out, exp) < TIGHT and rel_err(taps, etaps) < TIGHT
    assert rel_err(LS_Filter_Multiple(ref, srv, 5, 1e4, [3, 0]), O.LS_Filter_Multiple(ref, srv, 5, 1e4, [3, 0])) < TOL
    assert LS_Filter_Multiple(ref, srv, 5, 1e4, []) is srv               # empty bin list: input returned (:178)
    y = NLMS_filter(ref[:30], srv[:30], 25, 0.1)                         # n <= T: no step runs, all zeros
    assert y.shape == (30,) and not y.any()
    with pytest.raises(ValueError):
        LS_Filter_Toeplitz(ref[:20], srv[:20], 30)                       # more taps than samples


def test_concurrent_callers():
    """dask-style: worker threads call the drop-in functions concurrently (one plan cache per thread)"""
    import threading
    from passiveradar_amd.clutter_removal import LS_Filter_Toeplitz
    from passiveradar_amd.range_doppler_processing import fast_xambg
    n, R, F = 8192, 30, 64
    cases = [scene.make_scene(n, 1e4, R, 900 + i) for i in range(6)]
    exp = [(O.fast_xambg(a, s, R, F), O.LS_Filter_Toeplitz(a, s, R)) for a, s in cases]
    got = [None] * len(cases)
    errs = []

    def work(i):
        try:
            for _ in range(3):
                got[i] = (fast_xambg(cases[i][0], cases[i][1], R, F), LS_Filter_Toeplitz(cases[i][0], cases[i][1], R))
        except Exception as e:      # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for (gx, gl), (ex, el) in zip(got, exp):
        assert rel_err(gx, ex) < TIGHT and rel_err(gl, el) < TIGHT


def test_front_end_golden():
    """SURVEY 8f next #1 on the GPU: fused deinterleave -> block-phase tuning -> 13:119 resampler."""
    from passiveradar_amd.signal_utils import deinterleave_IQ, frequency_shift, front_end, resample
    g = load_golden("frontend")
    icl, fs, foff, up, dn = int(g["icl"]), int(g["fs"]), int(g["foff"]), int(g["up"]), int(g["dn"])
    assert np.array_equal(deinterleave_IQ(g["rawf"]), g["deintf"])
    assert np.array_equal(deinterleave_IQ(g["raw8"][icl:2 * icl]), g["deint1"])
    tuned = frequency_shift(g["deint1"], foff, fs, np.array([g["phases"][1]]))
    assert tuned.dtype == np.complex128 and rel_err(tuned, g["tuned1"]) < 1e-6
    out = front_end(g["raw8"], icl, foff, fs, up, dn)
    assert out.dtype == np.complex64 and out.shape == g["out"].shape
    assert rel_err(out, g["out"]) < TIGHT
    out2 = front_end(g["raw8"], icl, foff, fs, up, dn, max_blocks=2)       # batches of 2 + 1 blocks
    assert np.array_equal(out, out2)
    assert rel_err(resample(g["deint1"], 3, 7), g["res_c64"]) < TIGHT
    r = resample(g["tuned1"], up, dn)
    assert r.dtype == np.complex128 and rel_err(r, g["out"][len(g["out"]) // 3:2 * len(g["out"]) // 3]) < TIGHT


def test_cfar_golden():
    """SURVEY 8f next #3 on the GPU (single map, batch of maps, device tensors)."""
    import torch
    from passiveradar_amd.target_detection import CFAR_2D
    g = load_golden("cfar")
    cr = CFAR_2D(g["X"], 18, 4)
    assert cr.dtype == np.float64 and rel_err(cr, g["cr_18_4"]) < TIGHT
    assert rel_err(CFAR_2D(g["X"], 7, 2), g["cr_7_2"]) < TIGHT
    det = CFAR_2D(g["X"], 18, 4, float(g["thr"]) * 0.999)     # float32 ratio: test just inside the threshold
    assert det.dtype == bool and (det & ~g["det_18_4"]).sum() <= 1 and (g["det_18_4"] & ~det).sum() == 0
    stack = np.stack([g["X"], 2 * g["X"], g["X"][::-1]])
    out = CFAR_2D(torch.from_numpy(stack).cuda(), 18, 4).cpu().numpy()
    assert rel_err(out[0], g["cr_18_4"]) < TIGHT and rel_err(2 * out[1], g["cr_18_4"]) < TIGHT   # numerator is normalised, the box sum is not
    assert rel_err(out[2], O.CFAR_2D(g["X"][::-1], 18, 4)) < TIGHT


def test_channel_offset_golden():
    """SURVEY 8f next #2 on the GPU: spectral zero-phase IIR decimator + rocFFT correlation + argmax against the
    reference's find_channel_offset outputs (integer offsets identical, |xc| profile to float32 accuracy)."""
    from passiveradar_amd.signal_utils import decimate_iir, find_channel_offset
    g = load_golden("channel_offset")
    for tag in g["cases"]:
        s1, s2, nd, nl = g[f"{tag}_s1"], g[f"{tag}_s2"], int(g[f"{tag}_nd"]), int(g[f"{tag}_nl"])
        B1 = decimate_iir(s1, nd)
        assert B1.dtype == np.complex64 and B1.shape[0] == -(-s1.shape[0] // nd)
        assert rel_err(B1[:1500], g[f"{tag}_B1_head"]) < 2e-5 and rel_err(B1[-1500:], g[f"{tag}_B1_tail"]) < 2e-5
        assert rel_err(B1, O.decimate_iir(s1, nd)) < 5e-6
        off, xc = find_channel_offset(s1, s2, nd, nl, return_xc=True)
        assert off == int(g[f"{tag}_offset"]), tag
        assert xc.dtype == np.float32 and xc.shape == g[f"{tag}_xc"].shape
        assert rel_err(xc, g[f"{tag}_xc"]) < 2e-5
        assert find_channel_offset(s1, s2, nd, nl) == off


def test_channel_offset_edges():
    from passiveradar_amd.signal_utils import decimate_iir, find_channel_offset
    with pytest.raises(ValueError):
        decimate_iir(np.zeros(27, np.complex64), 1)                  # sosfiltfilt: n must exceed padlen
    with pytest.raises(ValueError):
        find_channel_offset(np.ones(4000, np.complex64), np.ones(100, np.complex64), 1, 10)   # 'valid' needs the padded s2 longer
    with pytest.raises(ValueError):
        find_channel_offset(np.ones(4000, np.complex64), np.ones(4000, np.complex64), 0, 10)
    # shortest legal input, a real-valued input, nl = 0 (a single lag)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(28).astype(np.float32)
    assert rel_err(decimate_iir(x, 1), O.decimate_iir(x, 1)) < 5e-6
    s = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
    off, xc = find_channel_offset(s, s, 1, 0, return_xc=True)
    assert off == 0 and xc.shape == (1,)
    # a delay well past the decimated correlation's noise floor, odd decimation factor
    s2 = np.roll(s, 333)
    assert find_channel_offset(s, s2, 3, 200) == O.find_channel_offset(s, s2, 3, 200) == -333


def test_ls_chain_first_bin_rotated():
    """cached multi-bin chain whose FIRST Doppler bin is non-zero: the caller's raw stream has to be taken into
    that bin's rotated frame on the way in (every later bin reads a stream already stored in its own frame)"""
    from passiveradar_amd.clutter_removal import LS_Filter_Multiple
    n, L, fs = 40000, 24, 1.0e4
    ref, srv = scene.make_scene(n, fs, L, 4711, targets=((11, 3.0, 0.05),))
    for bins in ([2, 0, -1], [-3, 3], [1, 1, 0]):
        exp = O.LS_Filter_Multiple(ref, srv, L, fs, bins)
        got = LS_Filter_Multiple(ref, srv, L, fs, bins)
        assert rel_err(got, exp) < TOL, bins


@pytest.mark.parametrize("L", [54, 55, 118, 375, 390, 438, 439, 600, 1014, 1015, 1200, 1700, 1800, 1960, 2038])
def test_nlms_every_tap_group_count(L):
    """one kernel instantiation per 64-tap group count: T = L + 10 on both sides of group boundaries and in the
    middle of what used to be coarse buckets (7, 10, 19, 27 groups), where whole groups beyond T must stay zero"""
    from oracle import c_oracle
    from passiveradar_amd.clutter_removal import NLMS_filter
    n = L + 10 + 1500
    ref, srv = scene.make_scene(n, 1e4, 50, 9000 + L)
    out, taps = NLMS_filter(ref, srv, L, 0.05, 10, None, True)
    exp, etaps = c_oracle.nlms(ref, srv, L, 0.05, 10)
    assert rel_err(out, exp) < TOL and rel_err(taps, etaps) < TOL


@pytest.mark.parametrize("L", [2039, 2040, 3001, 4086, 4087, 6000, 8182])
def test_nlms_beyond_one_wavefront(L):
    """filters longer than the 2048 taps one wavefront holds (the reference takes any length, clutter_removal.py:189-249):
    a stream becomes a workgroup of two (T <= 4096) or four (T <= 8192) wavefronts with consecutive tap ranges -- T = 2049
    (uneven split 1025 + 1024), 2050, 3011, 4096, 4097 (four wavefronts, 1025 + 3 x 1024), 6010, 8192; cold and warm start"""
    from oracle import c_oracle
    from passiveradar_amd.clutter_removal import NLMS_filter
    n = L + 10 + 1300
    ref, srv = scene.make_scene(n, 1e4, 50, 7000 + L)
    out, taps = NLMS_filter(ref, srv, L, 0.05, 10, None, True)
    exp, etaps = c_oracle.nlms(ref, srv, L, 0.05, 10)
    assert rel_err(out, exp) < TOL and rel_err(taps, etaps) < TOL
    out2, taps2 = NLMS_filter(ref, srv, L, 0.02, 10, taps.astype(np.complex64), True)
    exp2, etaps2 = c_oracle.nlms(ref, srv, L, 0.02, 10, taps.astype(np.complex64))
    assert rel_err(out2, exp2) < TOL and rel_err(taps2, etaps2) < TOL


@pytest.mark.parametrize("T", [2110, 4110])
def test_nlms_long_filters_vs_reference_golden(T):
    """two / four wavefronts per stream against the REFERENCE's own output (oracle/gen_golden.py::nlms_long_cases)"""
    from passiveradar_amd.clutter_removal import NLMS_filter
    g = load_golden(f"nlms_t{T}")
    a, s = scene.make_scene(int(g["N"]), float(g["fs"]), int(g["scene_R"]), int(g["seed"]))
    out, taps = NLMS_filter(a, s, int(g["L"]), float(g["mu"]), int(g["peek"]), None, True)
    assert rel_err(out, g["out"]) < TOL and rel_err(taps, g["taps"]) < TOL


@pytest.mark.parametrize("L,warm", [(8190, False), (12000, True), (20011, False)])
def test_nlms_of_any_length(L, warm):
    """beyond the 8192 taps of the register-resident kernels (round 5; the reference takes any length,
    clutter_removal.py:189-249): one workgroup per stream, taps in a global workspace, block-wide reductions in double
    -- a slow fallback held to the same bar against the C twin, cold and warm start, two streams in one launch"""
    from oracle import c_oracle
    from passiveradar_amd.clutter_removal import NLMS_filter
    n = L + 10 + 900
    ref, srv = scene.make_scene(n, 1e4, 50, 9000 + L)
    t0 = None
    if warm:
        rng = np.random.default_rng(L)
        t0 = ((rng.standard_normal(L + 10) + 1j * rng.standard_normal(L + 10)) * 1e-3).astype(np.complex64)
    out, taps = NLMS_filter(ref, srv, L, 0.05, 10, t0, True)
    exp, etaps = c_oracle.nlms(ref, srv, L, 0.05, 10, t0)
    assert taps.shape == (L + 10,) and rel_err(out, exp) < TOL and rel_err(taps, etaps) < TOL


@pytest.mark.parametrize("tail", [1, 5, 9, 10])
def test_ls_chain_last_piece_shorter_than_peek(tail):
    """cached chain, block length = 30 overlap-save pieces + `tail` samples: with tail < peek the run of `peek`
    wrapped reference samples straddles the last two pieces (they were once counted twice in the autocorrelation)"""
    from passiveradar_amd.clutter_removal import LS_Filter_Multiple
    L, fs = 48, 1.0e4
    n = 30 * (1025 - (L + 10)) + tail
    ref, srv = scene.make_scene(n, fs, 50, 12345)
    for bins in ([0, 0], [0, 2], [2, 0, -1]):
        assert rel_err(LS_Filter_Multiple(ref, srv, L, fs, bins), O.LS_Filter_Multiple(ref, srv, L, fs, bins)) < 5e-6, bins


@pytest.mark.parametrize("tail", [1, 5, 10, 2000])
def test_ls_team_chain_last_piece_shorter_than_peek(tail):
    """the same edge on the 4096-point chain (method 4): six pieces of 4097 - T samples + `tail`; also peek = 0 and a
    single-bin call (per-bin team kernels) on the same plan family"""
    from passiveradar_amd import clutter_removal as cr
    L, fs = 48, 1.0e4
    n = 6 * (4097 - (L + 10)) + tail
    ref, srv = scene.make_scene(n, fs, 50, 54321)
    cr.set_default_ls_method(4)
    try:
        for bins in ([0, 0], [0, 2], [2, 0, -1], [0]):
            assert rel_err(cr.LS_Filter_Multiple(ref, srv, L, fs, bins), O.LS_Filter_Multiple(ref, srv, L, fs, bins)) < 5e-6, bins
        got, taps = cr.LS_Filter_Toeplitz(ref, srv, L, 0, True)
        exp, etaps = O.LS_Filter_Toeplitz(ref, srv, L, 0, True)
        assert rel_err(got, exp) < TIGHT and rel_err(taps, etaps) < TIGHT
    finally:
        cr.set_default_ls_method(0)


def test_library_first_then_torch_in_a_fresh_process():
    """import order must not matter: libprcore used before torch is imported (fresh interpreter), then torch must
    still see the GPU and the device-tensor path must agree with the NumPy path (one HIP runtime per process)"""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, '.')\n"
        "from passiveradar_amd import scene\n"
        "from passiveradar_amd.range_doppler_processing import fast_xambg\n"
        "assert 'torch' not in sys.modules\n"
        "ref, srv = scene.make_scene(8192, 1e4, 20, 1)\n"
        "X = fast_xambg(ref, srv, 20, 32)\n"
        "import torch\n"
        "assert torch.cuda.is_available() and torch.cuda.device_count() >= 1\n"
        "Xt = fast_xambg(torch.from_numpy(ref).cuda(), torch.from_numpy(srv).cuda(), 20, 32)\n"
        "assert float(np.abs(Xt.cpu().numpy() - X).max()) == 0.0\n"
        "print('ok')\n")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=repo, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_library_and_torch_loaded_by_two_threads_at_once():
    """found by the round-5 fuzz run with two caller threads: one thread loads libprcore while another is half way through
    `import torch` (the module is in sys.modules before its libraries are loaded) -- the library then took /opt/rocm's HIP
    runtime first and torch reported "No HIP GPUs are available".  Fresh interpreters, both orders of the race."""
    import subprocess
    import sys
    code = (
        "import sys, threading, time, numpy as np\n"
        "sys.path.insert(0, '.')\n"
        "delay = float(sys.argv[1])\n"
        "out = {}\n"
        "def use_torch():\n"
        "    import torch\n"
        "    out['torch'] = float(torch.ones(4, device='cuda').sum().item())\n"
        "def use_lib():\n"
        "    time.sleep(delay)\n"
        "    from passiveradar_amd import scene\n"
        "    from passiveradar_amd.range_doppler_processing import fast_xambg\n"
        "    ref, srv = scene.make_scene(8192, 1e4, 20, 1)\n"
        "    out['lib'] = fast_xambg(ref, srv, 20, 32).shape\n"
        "ts = [threading.Thread(target=use_torch), threading.Thread(target=use_lib)]\n"
        "[t.start() for t in ts]; [t.join() for t in ts]\n"
        "assert out.get('torch') == 4.0 and out.get('lib') == (32, 21, 1), out\n"
        "print('ok')\n")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for delay in ("0.0", "0.05", "0.3", "1.0"):
        r = subprocess.run([sys.executable, "-c", code, delay], cwd=repo, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (delay, r.stderr[-2000:])


@pytest.mark.parametrize("nstreams,L", [(5, 24), (9, 90), (1030, 24), (2100, 24), (3100, 24)])
def test_nlms_many_streams(nstreams, L):
    """batched NLMS through the C ABI: independent streams at a stride, one wavefront each, 4 / 8 / 12 wavefronts
    per workgroup depending on the stream count (1030 -> two per SIMD, 3100 -> three), taps in and out per stream"""
    import torch
    from oracle import c_oracle
    from passiveradar_amd import engine
    n, stride = 500, 512
    base_ref, base_srv = scene.make_scene(nstreams * 7 + n, 1e4, 50, 31 + nstreams)
    ref = np.zeros((nstreams, stride), np.complex64)
    srv = np.zeros((nstreams, stride), np.complex64)
    tin = np.zeros((nstreams, L + 10), np.complex64)
    rng = np.random.default_rng(nstreams)
    for s in range(nstreams):
        ref[s, :n] = base_ref[7 * s:7 * s + n]
        srv[s, :n] = base_srv[7 * s:7 * s + n]
    warm = nstreams <= 9
    if warm:
        tin[:] = (rng.standard_normal(tin.shape) + 1j * rng.standard_normal(tin.shape)).astype(np.complex64) * 0.05
    dr, ds = torch.from_numpy(ref).cuda(), torch.from_numpy(srv).cuda()
    do = torch.full((nstreams, stride), 7.0 + 0j, dtype=torch.complex64, device="cuda")
    dti = torch.from_numpy(tin).cuda()
    dto = torch.empty_like(dti)
    engine.nlms_execute(dr, ds, do, n, L, 0.05, 10, dti if warm else None, dto, nstreams, stride, stride)
    torch.cuda.synchronize()
    out, taps = do.cpu().numpy(), dto.cpu().numpy()
    assert np.all(out[:, n:] == 7.0)                                   # nothing written past a stream's n samples
    check = range(nstreams) if nstreams <= 9 else list(range(0, nstreams, 97)) + [nstreams - 1, nstreams - 2, 1023 % nstreams, 1024 % nstreams]
    for s in check:
        e, et = c_oracle.nlms(ref[s, :n], srv[s, :n], L, 0.05, 10, tin[s] if warm else None)
        assert rel_err(out[s, :n], e) < TOL and rel_err(taps[s], et) < TOL, s


def test_argument_ranges_the_reference_accepts():
    """found by tests/fuzz_parity.py: lags beyond the signal length (zero sums), equal up/down factors (a copy),
    and a decimation ratio whose staged span needs fewer outputs per workgroup to fit LDS"""
    from passiveradar_amd.signal_utils import resample, xcorr
    a, b = scene.make_scene(150, 1e4, 20, 8)
    assert rel_err(xcorr(a, b, 30, 260), O.xcorr(a, b, 30, 260)) < TIGHT
    z = xcorr(a, b, 0, 400)
    assert z.shape == (401,) and not z[151:].any()
    x = scene.white_reference(4000, 5)
    y = resample(x, 7, 7)
    assert y is not x and np.array_equal(y, x)
    for up, dn in ((1, 101), (3, 128), (19, 2)):
        assert rel_err(resample(x, up, dn), O.resample(x, up, dn)) < TIGHT, (up, dn)


def test_plain_c_host_of_the_abi(tmp_path):
    """examples/c_host.c: a C99 program that drives libprcore.so directly (no Python, no torch in that process) must
    see the same cross-ambiguity surface as the Python drop-in on the same synthetic echo"""
    import shutil
    import subprocess
    from passiveradar_amd.range_doppler_processing import fast_xambg
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_host")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(repo, "include"), os.path.join(repo, "examples", "c_host.c"),
                           "-L", os.path.join(repo, "passiveradar_amd"), "-lprcore", "-lm",
                           "-Wl,-rpath," + os.path.join(repo, "passiveradar_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    tok = r.stdout.split()
    row, col, power, checksum = int(tok[2]), int(tok[4]), float(tok[6]), float(tok[8])
    n, R, F = 8192, 20, 32
    s = np.uint32(12345)
    v = np.empty(2 * n, np.float32)
    state = 12345
    for i in range(2 * n):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        v[i] = np.float32((state >> 8) / 8388608.0 - 1.0)
    ref = v.view(np.complex64)
    i = np.arange(n)
    srv = (ref[(i - 7) % n].astype(np.complex128) * np.exp(2j * np.pi * 5.0 * i / n)).astype(np.complex64)
    X = fast_xambg(ref, srv, R, F)[:, :, 0]
    P = np.abs(X.astype(np.complex128)) ** 2
    assert (row, col) == tuple(int(t) for t in np.unravel_index(P.argmax(), P.shape)) == (F // 2 - 5, R - 7)
    assert abs(power - P.max()) / P.max() < 1e-4 and abs(checksum - P.sum()) / P.sum() < 1e-4
    # the C host also pushed its map through prc_comm_* / prc_gather_frames (a world of one)
    assert "difference 0.0e+00" in r.stdout or "gather skipped" in r.stdout, r.stdout


def test_ls_cfg1_chunk_golden():
    """config 1's LS stage at full chunk size (cached five-bin chain, T = 266) against the reference's own output"""
    from passiveradar_amd.clutter_removal import LS_Filter_Multiple
    g = load_golden("ls_cfg1")
    n, R, fs = int(g["N"]), int(g["R"]), float(g["fs"])
    a, s = scene.make_scene(n, fs, R, int(g["seed"]))
    out = LS_Filter_Multiple(a, s, R, fs, [0, 1, -1, 2, -2])
    scale = np.abs(g["head"]).max()
    assert np.abs(out[::7] - g["out_sub"]).max() / scale < 1e-5
    assert np.abs(out[:600] - g["head"]).max() / scale < 1e-5 and np.abs(out[-600:] - g["tail"]).max() / scale < 1e-5


@pytest.mark.parametrize("name", ["ls_multiple_bin50", "ls_multiple_kHz"])
def test_ls_multiple_far_doppler_bins(name, ls_method):
    """Doppler bins far from zero (the reference takes any list, clutter_removal.py:178-187): 50 Hz at 10 kHz and
    +-1.5 kHz at 262 kHz put |2 pi f/Fs| * peek at 0.3-0.4 rad, beyond the Taylor range of the wrapped-sample phase --
    every kernel family must take them (round 1 refused them in auto mode)."""
    from passiveradar_amd.clutter_removal import LS_Filter_Multiple
    g = load_golden(name)
    out = LS_Filter_Multiple(g["ref"], g["srv"], int(g["L"]), float(g["fs"]), [float(b) for b in g["bins"]])
    assert rel_err(out, g["out"]) < TOL


def test_ls_far_bins_long_block_chain():
    """the cached-spectrum chain (long blocks) with bins of hundreds of Hz, against the float64 oracle"""
    from passiveradar_amd import clutter_removal as cr
    n, L, fs = 200000, 40, 262184.87
    ref, srv = scene.make_scene(n, fs, L, 2718)
    bins = [0.0, 300.0, -450.0, 2.0]
    exp = O.LS_Filter_Multiple(ref, srv, L, fs, bins)
    for m in (0, 1, 2, 3, 4):
        cr.set_default_ls_method(m)
        try:
            out = cr.LS_Filter_Multiple(ref, srv, L, fs, bins)
        finally:
            cr.set_default_ls_method(0)
        assert rel_err(out, exp) < TOL, m


def test_ls_at_the_config3_tap_count(ls_method):
    """T = 1034 (config 3's filter length, > the 769 taps of the 1024-point kernels) against the reference's own
    outputs: Toeplitz (+ taps), the three-bin chain, and the circular direct form"""
    from passiveradar_amd.clutter_removal import LS_Filter, LS_Filter_Multiple, LS_Filter_Toeplitz
    g = load_golden("ls_t1034")
    n, L, fs = int(g["N"]), int(g["L"]), float(g["fs"])
    a, s = scene.make_scene(n, fs, L, int(g["seed"]))
    out, taps = LS_Filter_Toeplitz(a, s, L, return_filter=True)
    assert rel_err(taps, g["taps"]) < TIGHT and rel_err(out, g["out"]) < TIGHT
    assert rel_err(LS_Filter_Multiple(a, s, L, fs, [float(b) for b in g["bins"]]), g["out_multiple"]) < TOL
    g = load_golden("ls_direct_t1034")
    a, s = scene.make_scene(int(g["N"]), float(g["fs"]), L, int(g["seed"]))
    out, taps = LS_Filter(a, s, L, return_filter=True)
    # the reference solves the 1034 x 1034 system and applies the 12288 x 1034 data matrix in complex64 (LAPACK cgesv,
    # clutter_removal.py:45,:51): its own output carries ~1e-3 of float32 noise at this size ...
    assert rel_err(taps, g["taps"]) < 5e-5 and rel_err(out, g["out"]) < 1e-3
    # ... so the north-star bar is held against a float64 evaluation of the same normal equations (the oracle's
    # LS_Filter: Gram matrix, right-hand side, solve and circular FIR in complex128), which the reference's own output
    # also matches to its 1e-3
    o64, t64 = O.LS_Filter(a, s, L, return_filter=True)
    assert rel_err(taps, t64) < TIGHT
    # The canceller removes 99 % of the surveillance signal here (|out| peaks at ~1e-2 of |srv|), so an error is stated
    # against both scales: against the INPUT every kernel family is at the float32 floor (< 2e-5); against the cleaned
    # stream's own peak every kernel family holds the 1e-4 bar -- the time-domain kernel too since round 4 (its 1034
    # products per sample are summed 16 at a time in float32 and the blocks in double; a single float32 accumulator
    # left 1e-3, the arithmetic of the reference's own complex64 matrix product, whose output is 3.6e-4 from the
    # float64 evaluation)
    e_in = float(np.abs(out - o64).max() / np.abs(s).max())
    e_out = rel_err(out, o64)
    print(f"LS_Filter T=1034 [{ls_method}]: vs float64 evaluation {e_in:.1e} of the input peak, {e_out:.1e} of the output peak")
    assert e_in < 2e-5
    assert e_out < TOL
    assert rel_err(g["out"], o64) < 1e-3


@pytest.mark.parametrize("L", [2100, 3063])
def test_ls_up_to_the_3073_taps_of_the_team_kernels(L):
    """filters longer than round 2's 2047-tap ceiling (the Levinson recursion kept five T-vectors in LDS; it keeps three
    now, updated in place): T = 2110 and T = 3073 = the most the 4096-point FFT kernels carry; the reference accepts
    any length (clutter_removal.py:109-160)"""
    from passiveradar_amd.clutter_removal import LS_Filter_Toeplitz
    n = 24000
    ref, srv = scene.make_scene(n, 1.0e7, 200, 3000 + L)
    exp, etaps = O.LS_Filter_Toeplitz(ref, srv, L, return_filter=True)
    out, taps = LS_Filter_Toeplitz(ref, srv, L, return_filter=True)
    assert rel_err(taps, etaps) < TIGHT and rel_err(out, exp) < TIGHT


def test_ls_beyond_the_lds_resident_solver():
    """3414 .. 5120 taps (round 4): the Levinson recursion keeps the two vectors it rewrites in LDS and reads the
    autocorrelation from a global workspace; correlations and FIR on the time-domain kernels (double accumulation).  The
    reference accepts any length (clutter_removal.py:109-160)"""
    from passiveradar_amd.clutter_removal import LS_Filter_Toeplitz
    n, L = 30000, 4000
    ref, srv = scene.make_scene(n, 1.0e7, 200, 3000 + L)
    exp, etaps = O.LS_Filter_Toeplitz(ref, srv, L, return_filter=True)
    out, taps = LS_Filter_Toeplitz(ref, srv, L, return_filter=True)
    assert rel_err(taps, etaps) < TIGHT and rel_err(out, exp) < TOL


@pytest.mark.parametrize("L,circular", [(5200, False), (9500, False), (5500, True)])
def test_ls_of_any_length(L, circular):
    """beyond 5120 taps (round 5; the reference accepts any length, clutter_removal.py:6-56, :109-160): the Levinson
    recursion with all three of its vectors in a global workspace (one wavefront per block, a device-scope fence per step)
    and, beyond ~9000 taps, the time-domain FIR in tap tiles -- slow fallbacks, held to the same bar"""
    from passiveradar_amd.clutter_removal import LS_Filter, LS_Filter_Toeplitz
    n = 3 * L + 5000
    ref, srv = scene.make_scene(n, 1.0e7, 200, 7000 + L)
    if circular:
        exp, etaps = O.LS_Filter(ref, srv, L, return_filter=True)
        out, taps = LS_Filter(ref, srv, L, return_filter=True)
    else:
        exp, etaps = O.LS_Filter_Toeplitz(ref, srv, L, return_filter=True)
        out, taps = LS_Filter_Toeplitz(ref, srv, L, return_filter=True)
    assert taps.shape == (L + 10,) and rel_err(taps, etaps) < 1e-5 and rel_err(out, exp) < TOL


def test_ls_t1034_long_block_vs_oracle(ls_method):
    """config-3-shaped LS_Filter_Multiple (T = 1034) on a block long enough for the cached chain"""
    from passiveradar_amd.clutter_removal import LS_Filter_Multiple
    n, L, fs = 300000, 1024, 1.0e7
    ref, srv = scene.make_scene(n, fs, L, 1034)
    exp = O.LS_Filter_Multiple(ref, srv, L, fs, [0, 1, -1, 2, -2])
    out = LS_Filter_Multiple(ref, srv, L, fs, [0, 1, -1, 2, -2])
    assert rel_err(out, exp) < TOL


def test_nlms_full_cfg3_hop_vs_c_oracle():
    """config 3's NLMS stage at FULL hop length: 2.5 M samples, T = 1034, mu = 0.02, one stream, against the C twin
    of the oracle (2.4 M sequential steps: drift between two float32 summation orders would show here)"""
    import time
    from oracle import c_oracle
    from passiveradar_amd.clutter_removal import NLMS_filter
    n, L = 2500000, 1024
    ref, srv = scene.make_scene(n, 1.0e7, L, scene.scene_seed(3))
    t0 = time.time()
    exp, etaps = c_oracle.nlms(ref, srv, L, 0.02, 10)
    t1 = time.time()
    out, taps = NLMS_filter(ref, srv, L, 0.02, 10, None, True)
    # the far end against the SAME scale (the converged residual is ~100x below the stream's peak: its own peak
    # would turn the float32 floor of two summation orders, 2e-5 of the stream, into 1.5e-3)
    e_out, e_taps = rel_err(out, exp), rel_err(taps, etaps)
    e_end = float(np.abs(out[-50000:] - exp[-50000:]).max() / np.abs(exp).max())
    print(f"C twin {t1 - t0:.1f}s, device {time.time() - t1:.1f}s; errors out {e_out:.2e} taps {e_taps:.2e} last-50k {e_end:.2e}")
    assert e_out < TOL and e_taps < TOL, (e_out, e_taps, e_end)
    assert e_end < TOL                                        # no drift at the far end either


def test_nlms_full_cfg3_hop_vs_reference_digest():
    """the same hop against the REFERENCE's own NLMS_filter output (clutter_removal.py:189-249 run for 2.5 M steps
    by oracle/gen_golden.py nlms_cfg3_digest_case): strided samples, both ends, final taps, output energy"""
    from passiveradar_amd.clutter_removal import NLMS_filter
    g = load_golden("nlms_cfg3_digest")
    n, L = int(g["N"]), int(g["L"])
    ref, srv = scene.make_scene(n, float(g["fs"]), L, int(g["seed"]))
    out, taps = NLMS_filter(ref, srv, L, float(g["mu"]), int(g["peek"]), None, True)
    peak = float(g["peak"])
    assert np.abs(out[::997] - g["sub"]).max() / peak < TOL
    assert np.abs(out[:4096] - g["head"]).max() / peak < TOL
    assert np.abs(out[-4096:] - g["tail"]).max() / peak < TOL
    assert rel_err(taps, g["taps"]) < TOL
    assert abs(float(np.vdot(out, out).real) / float(g["energy"]) - 1) < TOL


def test_nlms_reference_level_step():
    """the reference channel drops by 50 dB (and comes back) inside the tap window: u^H u must follow exactly,
    as the reference's per-step re-sum does (:213) -- a float32 sliding sum alone loses it to cancellation"""
    from oracle import c_oracle
    from passiveradar_amd.clutter_removal import NLMS_filter
    for L, n in ((24, 6000), (200, 9000)):
        ref, srv = scene.make_scene(n, 1e5, L, 5050 + L)
        g = np.ones(n, np.float32)
        g[n // 3:n // 2] = 10 ** (-50 / 20)
        ref2 = (ref * g).astype(np.complex64)
        srv2 = (srv * g + 0.001 * srv).astype(np.complex64)
        exp, etaps = c_oracle.nlms(ref2, srv2, L, 0.05, 10)
        out, taps = NLMS_filter(ref2, srv2, L, 0.05, 10, None, True)
        assert np.isfinite(out).all()
        assert rel_err(out, exp) < TOL and rel_err(taps, etaps) < TOL, L


# ---- 4096-point team transforms (caf_fft_team.hip): forced through the plan, every branch of the kernel ----
@pytest.fixture(params=[0, 1], ids=["four_waves_16_points", "eight_waves_8_points"])
def caf_team(request):
    """both forms of the 4096-point segment kernel: teams of four wavefronts (fft_team.h) and of eight (fft_team8.h,
    PRC_OPT_CAF_TEAM8, read per launch)"""
    from passiveradar_amd import _lib, range_doppler_processing as rdp
    rdp.set_default_methods(caf=3)
    old = _lib.set_option(_lib.OPT_CAF_TEAM8, request.param)
    yield
    _lib.set_option(_lib.OPT_CAF_TEAM8, old)
    rdp.set_default_methods(caf=0)


@pytest.mark.parametrize("n,R,F,win,n_in", [
    (8192, 70, 2, True, None),        # one piece per segment, 71 lags
    (65536, 300, 16, True, None),     # q+1 = 4097 -> one piece of 3796 + remainder piece
    (65536, 1024, 8, True, None),     # config-3 lag span: pieces of 3072
    (131072, 2048, 32, True, None),   # config-5 segment shape: 4097 = 2 x 2048 + 1 -> the direct tail sample
    (1 << 17, 2040, 32, False, None), # tail of 9 samples (B = 2056), no window
    (100000, 3500, 4, True, None),    # two lag blocks of 1751, pieces of 2346, odd q
    (40000, 9000, 2, False, None),    # five lag blocks, wrap of srv inside pieces
    (20000, 700, 3, True, 17000),     # zero-pad branch: n_valid < n
    (8192, 4000, 1, False, None),     # range span close to n/2, everything wraps
])
def test_caf_team_vs_oracle_shapes(n, R, F, win, n_in, caf_team):
    from passiveradar_amd.range_doppler_processing import fast_xambg
    m = n if n_in is None else n_in
    ref, srv = scene.make_scene(m, 1e4, min(R, 200), 9090 + n + R)
    w = np.kaiser(n, 5.0) if win else None
    exp = O.fast_xambg(ref, srv, R, F, n, w)
    out = fast_xambg(ref, srv, R, F, n, w)
    assert out.shape == (F, R + 1, 1)
    e = rel_err(out, exp)
    assert e < TIGHT, e


@pytest.mark.parametrize("name", ["caf_cfg1", "caf_cfg2"])
def test_caf_team_full_size_golden(name, caf_team):
    """the reference's own full-size config-1 / config-2 surfaces through the 4096-point kernel (AUTO keeps the
    1024-point one for these 257-lag spans; forced here)"""
    from scipy.signal import get_window
    from passiveradar_amd.range_doppler_processing import fast_xambg
    g = load_golden(name)
    n, R, F = int(g["N"]), int(g["R"]), int(g["F"])
    ref, srv = scene.make_scene(n, float(g["fs"]), R, int(g["seed"]))
    out = fast_xambg(ref, srv, R, F, n, get_window(("kaiser", 5.0), n))
    assert rel_err(out, g["out"]) < TOL


def test_caf_auto_picks_the_team_kernel_for_wide_spans():
    from passiveradar_amd import _lib
    from passiveradar_amd.range_doppler_processing import caf_plan_for
    assert caf_plan_for(2400000, 256, 512).method == _lib.CAF_FFT              # config 2
    assert caf_plan_for(5000000, 1024, 1024).method == _lib.CAF_FFT4096       # config 3
    assert caf_plan_for(1 << 23, 2048, 2048).method == _lib.CAF_FFT4096       # config 5
    assert caf_plan_for(4096, 7, 64).method == _lib.CAF_FFT


# ---- Doppler stage: one column-FFT kernel (doppler_col.h) vs the rocFFT path vs the oracle -----------------------
@pytest.fixture(params=["column", "rocfft"])
def doppler_method(request):
    from passiveradar_amd import range_doppler_processing as rdp
    rdp.set_default_methods(doppler={"rocfft": 1, "column": 2}[request.param])
    yield request.param
    rdp.set_default_methods(doppler=0)


@pytest.mark.parametrize("n,R,F,caf", [(65536, 40, 256, 2), (1 << 16, 300, 512, 2), (1 << 17, 33, 1024, 1),
                                      (1 << 18, 129, 2048, 2), (1 << 18, 7, 4096, 1), (1 << 17, 1500, 512, 3)])
def test_doppler_methods_vs_oracle(n, R, F, caf, doppler_method):
    """every column-FFT size (256 ... 4096) behind every segment kernel, ragged column tiles, against the oracle"""
    from passiveradar_amd import range_doppler_processing as rdp
    ref, srv = scene.make_scene(n, 1e5, R, 515 + F + R)
    w = np.kaiser(n, 5.0)
    exp = O.fast_xambg(ref, srv, R, F, n, w)
    rdp.set_default_methods(caf=caf)
    try:
        out = rdp.fast_xambg(ref, srv, R, F, n, w)
    finally:
        rdp.set_default_methods(caf=0)
    assert rel_err(out, exp) < TIGHT


def test_auto_picks_the_column_doppler_kernel():
    from passiveradar_amd import _lib
    from passiveradar_amd.range_doppler_processing import caf_plan_for
    assert caf_plan_for(2400000, 256, 512).doppler == _lib.DOPPLER_COLUMN
    assert caf_plan_for(1 << 23, 2048, 2048).doppler == _lib.DOPPLER_COLUMN
    assert caf_plan_for(6000, 6, 64).doppler == _lib.DOPPLER_ROCFFT
    assert caf_plan_for(5000, 4, 51).doppler == _lib.DOPPLER_ROCFFT


def test_batched_frames_in_cache_sized_groups():
    """prc_caf_execute alternates segment sums and Doppler transforms over groups of surfaces; a budget of a fraction
    of a surface forces one-frame groups, and the maps must not depend on the grouping"""
    import torch
    from passiveradar_amd import engine
    n, R, F, nf = 1 << 15, 600, 256, 7               # a surface is 1.2 MB: a 1 MiB budget forces one-frame groups
    ref, srv = scene.make_scene(n // 2 * (nf + 1), 1e5, R, 808)
    a, s = torch.from_numpy(ref).cuda(), torch.from_numpy(srv).cuda()
    outs = []
    from passiveradar_amd import _lib
    for mb in (1, 0, 3):             # groups of one frame, the whole batch, groups of two frames
        old = _lib.set_option(_lib.OPT_CAF_GROUP_MB, mb)      # read by the plan at creation
        try:
            plan = engine.CafPlan(n, R, F, nf)
        finally:
            _lib.set_option(_lib.OPT_CAF_GROUP_MB, old)
        out = torch.zeros((nf, F, R + 1), dtype=torch.complex64, device="cuda")
        plan.execute(a, s, out, nf, n // 2, n, None)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
        plan.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    exp = O.fast_xambg(ref[n // 2 * 3:n // 2 * 3 + n], srv[n // 2 * 3:n // 2 * 3 + n], R, F, n, None)[:, :, 0]
    assert rel_err(outs[0][3], exp) < TIGHT


# ---- several illuminators against one surveillance channel: prc_caf_execute_multi --------------------------------
@pytest.mark.parametrize("n,R,F,nref,caf,win,n_in", [
    (131072, 2048, 32, 4, 0, True, None),     # config-5 segment shape (two pieces of 2048 + the direct tail sample)
    (65536, 1024, 8, 3, 0, True, None),       # config-3 lag span: pieces of 3072, two per segment
    (65536, 300, 16, 2, 3, False, None),      # forced 4096-point method, one full piece + a remainder piece
    (1 << 18, 300, 16, 3, 3, True, None),     # five pieces per segment: the illuminators take turns (nothing shared)
    (40000, 9000, 2, 2, 3, False, None),      # several lag blocks, wrap of srv inside pieces
    (20000, 700, 3, 4, 3, True, 17000),       # zero-pad branch: n_valid < n
    (32768, 64, 256, 5, 0, True, None),       # 1024-point method: the fallback of the multi call, column Doppler
    (6000, 6, 64, 2, 0, True, None),          # rocFFT Doppler path
])
@pytest.mark.parametrize("mode", ["turns", "shared", "pairs"])
def test_caf_multi_equals_single_calls_and_oracle(n, R, F, nref, caf, win, n_in, mode):
    """prc_caf_execute_multi (every reference channel against ONE surveillance channel in one call) returns what one
    fast_xambg per pair returns (range_doppler_processing.py:81-89) -- and what the oracle computes"""
    from passiveradar_amd import range_doppler_processing as rdp
    # every way prc_caf_execute_multi can run (prc_caf_desc.multi): one pass per illuminator, the surveillance
    # transforms shared by all illuminators, or by pairs of them
    m = n if n_in is None else n_in
    refs, srv = scene.make_multi_scene(m, 1e5, min(R, 200), [7000 + 13 * i + n + R for i in range(nref)])
    w = np.kaiser(n, 5.0) if win else None
    rdp.set_default_methods(caf=caf)
    try:
        outs = rdp.fast_xambg_multi(refs, srv, R, F, n, w, mode=mode)
        singles = [rdp.fast_xambg(r, srv, R, F, n, w) for r in refs]
    finally:
        rdp.set_default_methods(caf=0)
    assert len(outs) == nref
    for i in range(nref):
        assert outs[i].shape == (F, R + 1, 1) and outs[i].dtype == np.complex64
        assert rel_err(outs[i], singles[i]) < 2e-6, i            # same arithmetic up to the order of two sums
        assert rel_err(outs[i], O.fast_xambg(refs[i], srv, R, F, n, w)) < TIGHT, i


def test_caf_multi_batched_overlapped_frames_and_errors():
    import torch
    from passiveradar_amd import engine
    from passiveradar_amd.range_doppler_processing import fast_xambg_multi
    n, R, F, nf, nref = 1 << 16, 1024, 8, 3, 2
    refs, srv = scene.make_multi_scene(n // 2 * (nf + 1), 1e5, 100, [91, 92])
    dr = [torch.from_numpy(r).cuda() for r in refs]
    ds = torch.from_numpy(srv).cuda()
    plan = engine.CafPlan(n, R, F, nf * nref)
    outs = [torch.zeros((nf, F, R + 1), dtype=torch.complex64, device="cuda") for _ in range(nref)]
    plan.execute_multi(dr, ds, outs, nf, n // 2, n, None)
    torch.cuda.synchronize()
    for i in range(nref):
        for b in range(nf):
            exp = O.fast_xambg(refs[i][b * n // 2:b * n // 2 + n], srv[b * n // 2:b * n // 2 + n], R, F, n, None)
            assert rel_err(outs[i][b].cpu().numpy(), exp[:, :, 0]) < TIGHT, (i, b)
    with pytest.raises(ValueError):
        plan.execute_multi(dr, ds, outs, nf + 1, n // 2, n, None)          # surfaces beyond max_frames
    with pytest.raises(ValueError):
        fast_xambg_multi([refs[0][:100]], srv, R, F)                       # length mismatch (:46-49)
    # device tensors in, device tensors out
    dd = fast_xambg_multi([r[:n] for r in dr], ds[:n], R, F)
    assert rel_err(dd[1].cpu().numpy(), O.fast_xambg(refs[1][:n], srv[:n], R, F)) < TIGHT


@pytest.mark.parametrize("mode", ["auto", "turns", "shared", "pairs"])
def test_caf_cfg5_digest_multi(mode):
    """BASELINE config 5 at full size (N = 2^23, 2048 x 2048, four illuminators against one surveillance channel)
    against digests of the REFERENCE's own fast_xambg, one per illuminator (oracle/gen_golden.py
    caf_cfg5_digest_case); the multi call, which shares the surveillance transforms"""
    from scipy.signal import get_window
    from passiveradar_amd.range_doppler_processing import fast_xambg_multi
    g = load_golden("caf_cfg5_digest")
    n, R, F = int(g["N"]), int(g["R"]), int(g["F"])
    refs, srv = scene.make_multi_scene(n, float(g["fs"]), R, [int(sd) for sd in g["seeds"]])
    outs = fast_xambg_multi(refs, srv, R, F, n, get_window(("kaiser", 5.0), n), mode=mode)
    for i, out in enumerate(outs):
        out = out[:, :, 0]
        peak = float(g[f"ill{i}_peak"])
        assert np.abs(out[::16, ::16] - g[f"ill{i}_sub"]).max() / peak < TOL, i
        assert np.abs(out.ravel()[g[f"ill{i}_top_idx"]] - g[f"ill{i}_top_val"]).max() / peak < TOL, i
        assert np.abs(out.sum(axis=0) - g[f"ill{i}_col_sums"]).max() / (peak * np.sqrt(F)) < TOL, i
        assert np.abs(out.sum(axis=1) - g[f"ill{i}_row_sums"]).max() / (peak * np.sqrt(R + 1)) < TOL, i


@pytest.mark.parametrize("method", ["fft", "direct"])
def test_caf_long_filter_at_config1_size(method):
    """fast_xambg(..., shortFilt=False) at a BASELINE size (config 1: N = 262 144, 257 lags, 256 Doppler bins, q = 1024:
    the flat-top decimation FIR of 10 241 taps, range_doppler_processing.py:73-78) against the reference's own surface
    (golden caf_longfilt_cfg1), on the FFT segment kernel's long-FIR form and on the time-domain kernel"""
    from scipy.signal import get_window
    from passiveradar_amd import range_doppler_processing as rdp
    g = load_golden("caf_longfilt_cfg1")
    n, R, F = int(g["N"]), int(g["R"]), int(g["F"])
    ref, srv = scene.make_scene(n, float(g["fs"]), R, int(g["seed"]))
    rdp.set_default_methods(caf={"direct": 1, "fft": 2}[method])
    try:
        plan = rdp.caf_plan_for(n, R, F, shortFilt=False)
        assert plan.method == {"direct": 1, "fft": 2}[method]
        out = rdp.fast_xambg(ref, srv, R, F, n, get_window(("kaiser", 5.0), n), shortFilt=False)[:, :, 0]
    finally:
        rdp.set_default_methods(caf=0)
    assert rel_err(out, g["out"]) < TOL
    # AUTO takes the FFT form for a long FIR up to 769 lags, the time-domain kernel beyond
    assert rdp.caf_plan_for(n, R, F, shortFilt=False).method == 2
    assert rdp.caf_plan_for(n, 1024, F, shortFilt=False).method == 1


def test_caf_multi_auto_choice():
    """prc_caf_desc.multi = AUTO resolves per shape to what was measured fastest on MI355X: the shared-surveillance
    kernel where a segment's first piece is longer than 2048 samples (config-3 span), turns otherwise (config 5), and
    turns wherever the shared kernel does not apply; an explicit request is kept"""
    from passiveradar_amd import _lib, engine
    assert engine.CafPlan(5000000, 1024, 1024, 2).multi == _lib.CAF_MULTI_SHARED
    assert engine.CafPlan(1 << 23, 2048, 2048, 2).multi == _lib.CAF_MULTI_TURNS
    assert engine.CafPlan(2400000, 256, 512, 2).multi == _lib.CAF_MULTI_TURNS          # 1024-point method
    assert engine.CafPlan(1 << 23, 2048, 2048, 2, multi="pairs").multi == _lib.CAF_MULTI_PAIRS
    old = _lib.set_option(_lib.OPT_CAF_MULTI_MODE, _lib.CAF_MULTI_SHARED)              # what AUTO means, process-wide
    try:
        assert engine.CafPlan(1 << 23, 2048, 2048, 2).multi == _lib.CAF_MULTI_SHARED
    finally:
        _lib.set_option(_lib.OPT_CAF_MULTI_MODE, old)


@pytest.mark.parametrize("L", [256, 700])
def test_ls_plan_without_a_spectrum_cache(L):
    """when the spectrum cache cannot be had (allocation failure, or PRC_OPT_LS_CACHE_LIMIT_MB as here) a plan
    that asked for the 4096-point chain falls back BEFORE it sizes its workspaces -- to the 1024-point chain, which
    recomputes the block spectra per bin when its own cache is refused too -- with the same results as the cached chain"""
    import torch
    from passiveradar_amd import _lib, engine
    n, fs, bins = 131072, 2.4e6, (0.0, 1.0, -1.0)
    ref, srv = scene.make_scene(n, fs, min(L, 200), 4242 + L)
    a, s = torch.from_numpy(ref).cuda(), torch.from_numpy(srv).cuda()
    outs = []
    for limit in (0, 1):                                   # 0: no limit; 1 MiB: no cache of this size fits
        old = _lib.set_option(_lib.OPT_LS_CACHE_LIMIT_MB, limit)
        try:
            plan = engine.LsPlan(n, L, 10, False, 2, 4)
        finally:
            _lib.set_option(_lib.OPT_LS_CACHE_LIMIT_MB, old)
        out = torch.empty_like(s)
        plan.execute(a, s, out, 1, n, n, fs, bins, 0.0, None, _lib.torch_stream_ptr())
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
        plan.close()
    exp = O.LS_Filter_Multiple(ref, srv, L, fs, list(bins))
    assert rel_err(outs[0], exp) < TOL and rel_err(outs[1], exp) < TOL
    assert rel_err(outs[1], outs[0]) < 1e-5


@pytest.mark.parametrize("nref", [1, 3])
def test_caf_team_workgroup_orders_give_the_same_maps(nref):
    """the 4096-point segment kernel's workgroup -> (frame, segment, channel) map has three forms (launch order, an XCD-
    contiguous run of segments per XCD, the two frames that cover the same samples in consecutive slots): batched
    50 %-overlapped frames, odd frame count, one and three channels -- bit-identical maps, and the oracle's for one frame"""
    import torch
    from passiveradar_amd import _lib, engine
    n, R, F, nf = 1 << 17, 1024, 64, 5
    refs, srv = scene.make_multi_scene(n // 2 * (nf + 1), 1e5, 100, [501 + i for i in range(nref)])
    dr = [torch.from_numpy(r).cuda() for r in refs]
    ds = torch.from_numpy(srv).cuda()
    win = torch.from_numpy(np.kaiser(n, 5.0).astype(np.float32)).cuda()
    results = []
    for contig, pair in ((0, 0), (1, 0), (0, 1)):
        o1 = _lib.set_option(_lib.OPT_CAF_XCD_CONTIG, contig)
        o2 = _lib.set_option(_lib.OPT_CAF_PAIR_FRAMES, pair)
        try:
            plan = engine.CafPlan(n, R, F, nf * nref, _lib.CAF_FFT4096, multi="turns")
            outs = [torch.zeros((nf, F, R + 1), dtype=torch.complex64, device="cuda") for _ in range(nref)]
            if nref == 1:
                plan.execute(dr[0], ds, outs[0], nf, n // 2, n, win)
            else:
                plan.execute_multi(dr, ds, outs, nf, n // 2, n, win)
            torch.cuda.synchronize()
            results.append([o.cpu().numpy() for o in outs])
            plan.close()
        finally:
            _lib.set_option(_lib.OPT_CAF_XCD_CONTIG, o1)
            _lib.set_option(_lib.OPT_CAF_PAIR_FRAMES, o2)
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert np.array_equal(a, b)
    b = nf - 2
    exp = O.fast_xambg(refs[-1][b * n // 2:b * n // 2 + n], srv[b * n // 2:b * n // 2 + n], R, F, n, np.kaiser(n, 5.0))
    assert rel_err(results[0][-1][b], exp[:, :, 0]) < TIGHT


@pytest.mark.parametrize("up,dn", [(13, 119), (3, 7), (1, 4), (5, 4), (16, 15), (2, 9), (17, 40), (3, 170)])
def test_front_end_kernel_forms_agree(up, dn):
    """frontend_group_kernel (`up` outputs per thread, taps through the scalar unit, rows split over four wavefronts)
    against frontend_kernel (one output per thread) and the oracle: every raw type, tuning on and off, blocks shorter
    than one window, an even decimation (padded LDS layout), and two ratios the group form does not take (up = 17; a
    window of 64 x 170 samples, more than two workgroups per CU can hold): AUTO falls back, an explicit request is refused."""
    from passiveradar_amd import _lib, engine
    from passiveradar_amd.signal_utils import front_end, resample
    rng = np.random.default_rng(up * 100 + dn)
    fs, foff = 2_400_000, 100_000
    old = _lib.get_option(_lib.OPT_FE_METHOD)
    try:
        for dt, n_in, nblk in (("int8", 9001, 3), ("int16", 700, 2), ("float32", 64 * dn + 5, 1), ("uint8", 2 * dn + 3, 4)):
            if dt == "float32":
                raw = rng.standard_normal(2 * n_in * nblk).astype(np.float32)
            else:
                info = np.iinfo(dt)
                raw = rng.integers(info.min, info.max, 2 * n_in * nblk, endpoint=True).astype(dt)
            got = {}
            for method in (1, 2, 0):
                if method == 2 and (up > 16 or dn >= 150):
                    _lib.set_option(_lib.OPT_FE_METHOD, 2)
                    with pytest.raises(RuntimeError):
                        front_end(raw, 2 * n_in, foff, fs, up, dn, max_blocks=2)
                    continue
                _lib.set_option(_lib.OPT_FE_METHOD, method)
                got[method] = front_end(raw, 2 * n_in, foff, fs, up, dn, max_blocks=2)
            exp = O.front_end(raw, 2 * n_in, foff, fs, up, dn)
            for method, y in got.items():
                assert y.shape == exp.shape and rel_err(y, exp) < TIGHT, (dt, method)
            if 2 in got:
                assert rel_err(got[2], got[1]) < 2e-6, dt               # same samples, same taps; the sum is split in four
                assert np.array_equal(got[0], got[2])                   # the default IS the group form where it applies
        x = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
        for method in (1, 0):
            _lib.set_option(_lib.OPT_FE_METHOD, method)
            assert rel_err(resample(x, up, dn), O.resample(x, up, dn)) < TIGHT
    finally:
        _lib.set_option(_lib.OPT_FE_METHOD, old)


@pytest.mark.parametrize("H,W,fw,gw", [(1024, 177, 18, 4), (64, 257, 18, 4), (5, 7, 18, 4), (33, 64, 7, 2), (40, 130, 9, 0),
                                       (16, 65, 8, 3), (3, 200, 5, 1), (100, 3, 6, 2), (70, 90, 31, 11)])
def test_cfar_kernel_forms_agree(H, W, fw, gw):
    """cfar_sep_kernel (row sums, then column sums: 2 fw LDS reads per cell) against cfar_kernel (every tap of the box
    per cell) and the oracle's convolve2d restatement: maps smaller than the box (wrap-around more than once), odd and
    even widths (the reference's asymmetric centring), no guard cells, several frames, ratio and threshold outputs."""
    import torch
    from passiveradar_amd import _lib
    from passiveradar_amd.target_detection import CFAR_2D
    rng = np.random.default_rng(H * 1000 + W)
    X = np.abs(rng.standard_normal((3, H, W)) + 1j * rng.standard_normal((3, H, W))).astype(np.float32)
    X[1, H // 2, W // 3] += 1e3                                   # a strong cell: the box next to it must not lose the noise
    old = _lib.get_option(_lib.OPT_CFAR_METHOD)
    try:
        got = {}
        for method in (1, 0):
            _lib.set_option(_lib.OPT_CFAR_METHOD, method)
            got[method] = CFAR_2D(torch.from_numpy(X).cuda(), fw, gw).cpu().numpy()
        for k in range(3):
            exp = O.CFAR_2D(X[k], fw, gw)
            for method in (0, 1):
                assert rel_err(got[method][k], exp) < TIGHT, (method, k)
        assert rel_err(got[0], got[1]) < 2e-6
        thr = float(np.median(got[1]))
        det = {}
        for method in (1, 0):
            _lib.set_option(_lib.OPT_CFAR_METHOD, method)
            det[method] = CFAR_2D(torch.from_numpy(X).cuda(), fw, gw, thr).cpu().numpy()
        assert (det[0] != det[1]).mean() < 1e-3                   # cells within a rounding of the threshold may flip
    finally:
        _lib.set_option(_lib.OPT_CFAR_METHOD, old)


@pytest.mark.parametrize("up,dn", [(13, 119), (3, 7), (5, 4), (16, 15), (17, 40), (3, 170)])
def test_front_end_two_channels_in_one_launch(up, dn):
    """prc_frontend_execute2 / HipBackend.front_end2 (main.py:133-149: both recordings are tuned with one frequency and
    the same block phases): both channels of a block in one workgroup, one rotation factor per input sample for the two.
    Bit-identical to two one-channel calls of the same kernel form -- every raw type, blocks shorter than a window
    (both ends 'line'-extended inside one workgroup), an even decimation (padded LDS layout), pieces of a recording with
    block0 -- and for the ratios the group form does not carry (up = 17, a 64 x 170 window) the fallback runs the two
    channels one after the other."""
    import torch
    from passiveradar_amd import _lib
    from passiveradar_amd.stream import HipBackend
    rng = np.random.default_rng(up * 1000 + dn)
    fs, foff = 2_400_000, 100_000
    be = HipBackend(4096, 16, 32, 2.6e5, batch=4, clutter=None)
    for dt, n_in, nblk in (("int8", 9001, 5), ("int16", 700, 2), ("float32", 32 * dn + 5, 1), ("uint8", 2 * dn + 3, 4)):
        if dt == "float32":
            ra, rb = (rng.standard_normal(2 * n_in * nblk).astype(np.float32) for _ in range(2))
        else:
            info = np.iinfo(dt)
            ra, rb = (rng.integers(info.min, info.max, 2 * n_in * nblk, endpoint=True).astype(dt) for _ in range(2))
        args = (2 * n_in, foff, fs, up, dn)
        one_a, one_b = be.front_end(ra, *args, max_blocks=2), be.front_end(rb, *args, max_blocks=2)
        two_a, two_b = be.front_end2(ra, rb, *args, max_blocks=2)
        torch.cuda.synchronize()
        assert torch.equal(two_a, one_a) and torch.equal(two_b, one_b), dt
        assert rel_err(two_b.cpu().numpy(), O.front_end(rb, 2 * n_in, foff, fs, up, dn)) < TIGHT, dt
        if nblk > 2:                       # a recording converted in pieces: the block phases continue (block0)
            n_out = one_a.shape[0] // nblk
            pa, pb = torch.zeros_like(one_a), torch.zeros_like(one_b)
            for b0, m in ((0, 1), (1, nblk - 1)):
                sl = slice(b0 * 2 * n_in, (b0 + m) * 2 * n_in)
                be.front_end2(ra[sl], rb[sl], *args, max_blocks=2, block0=b0,
                              out_ref=pa[b0 * n_out:(b0 + m) * n_out], out_srv=pb[b0 * n_out:(b0 + m) * n_out])
            torch.cuda.synchronize()
            assert torch.equal(pa, one_a) and torch.equal(pb, one_b), dt
    with pytest.raises(ValueError):
        be.front_end2(ra, rb[:-2], *args)
    with pytest.raises(ValueError):
        be.front_end(ra, *args, out=torch.zeros(3, dtype=torch.complex64, device="cuda"))      # too short
    with pytest.raises(ValueError):
        be.front_end(ra, *args, out=torch.zeros(1 << 20, dtype=torch.float32, device="cuda"))  # wrong dtype


@pytest.mark.parametrize("H,W,fw,gw", [(1024, 177, 18, 4), (512, 257, 18, 4), (5, 7, 18, 4), (33, 64, 7, 2), (70, 90, 31, 11)])
def test_cfar_of_the_complex_map_in_one_kernel(H, W, fw, gw):
    """CFAR_2D_abs(X) = CFAR_2D(np.abs(X)) as range_doppler_plot.py:56-57 calls it, |X| taken on the tile load
    (prc_cfar2d_c64): against the oracle on np.abs(X), against the two-step device path, both kernel forms, host arrays
    and device tensors, ratio and threshold outputs"""
    import torch
    from passiveradar_amd import _lib
    from passiveradar_amd.target_detection import CFAR_2D, CFAR_2D_abs
    rng = np.random.default_rng(H * 1000 + W + 1)
    X = (rng.standard_normal((3, H, W)) + 1j * rng.standard_normal((3, H, W))).astype(np.complex64)
    X[1, H // 2, W // 3] += 1e3
    old = _lib.get_option(_lib.OPT_CFAR_METHOD)
    try:
        for method in (0, 1):
            _lib.set_option(_lib.OPT_CFAR_METHOD, method)
            got = CFAR_2D_abs(torch.from_numpy(X).cuda(), fw, gw).cpu().numpy()
            two = CFAR_2D(torch.from_numpy(np.abs(X)).cuda(), fw, gw).cpu().numpy()
            assert rel_err(got, two) < 2e-6, method           # hypotf on the device against NumPy's: an ulp of the input
            for k in range(3):
                assert rel_err(got[k], O.CFAR_2D(np.abs(X[k]), fw, gw)) < TIGHT, (method, k)
        one = CFAR_2D_abs(X[0], fw, gw)
        assert one.dtype == np.float64 and one.shape == (H, W) and rel_err(one, O.CFAR_2D(np.abs(X[0]), fw, gw)) < TIGHT
        thr = float(np.median(one))
        det = CFAR_2D_abs(X[0], fw, gw, thr)
        assert det.dtype == bool and (det != (one > thr)).mean() < 1e-3
    finally:
        _lib.set_option(_lib.OPT_CFAR_METHOD, old)
    with pytest.raises(ValueError):
        CFAR_2D_abs(X[0, 0], fw, gw)


@pytest.mark.parametrize("up,dn", [(13, 119), (3, 7), (16, 15), (7, 45), (2, 9), (5, 4), (1, 5), (12, 7)])
def test_front_end_folded_tap_rows_give_the_same_samples(up, dn):
    """PRC_OPT_FE_FOLD (round 6, default 1, read per launch): rows rho and rho + M of the banded tap table never meet the same
    output of a group, so they share one row of full width (`FegArgs.T2`: 193 rows instead of 304 at 13:119 -- no
    multiply-adds on the zeros of the table's corners); the columns q < q0(rho) multiply the input of row rho + M, the
    others the input of row rho.  Odd decimations only (the even ones, which pad the LDS window, keep the unfolded rows:
    2:9 ... 12:7 here run folded, 5:4 does not).  Same samples as the unfolded rows to float32 rounding (the row sums run
    in another order), the same as the oracle (signal_utils.py:15-27, main.py:105-166), one and two channels, every raw
    type, blocks shorter than one window."""
    import torch
    from passiveradar_amd import _lib
    from passiveradar_amd.stream import HipBackend
    rng = np.random.default_rng(up * 131 + dn)
    old = _lib.get_option(_lib.OPT_FE_FOLD)
    try:
        for dt, n_in, nblk in (("int8", 64 * dn + 37, 3), ("int16", 9001, 2), ("float32", 2 * dn + 3, 4)):
            if dt == "float32":
                ra, rb = (rng.standard_normal(2 * n_in * nblk).astype(np.float32) for _ in range(2))
            else:
                info = np.iinfo(dt)
                ra, rb = (rng.integers(info.min, info.max, 2 * n_in * nblk, endpoint=True).astype(dt) for _ in range(2))
            args = (2 * n_in, 100_000, 2_400_000, up, dn)
            exp = O.front_end(rb, *args)
            be = HipBackend(4096, 16, 32, 2.6e5, batch=4, clutter=None)
            got = {}
            for fold in (0, 1):
                _lib.set_option(_lib.OPT_FE_FOLD, fold)
                one = be.front_end(rb, *args, max_blocks=2)
                two = be.front_end2(ra, rb, *args, max_blocks=2)[1]
                torch.cuda.synchronize()
                assert torch.equal(one, two), (dt, fold)
                got[fold] = one.cpu().numpy()
                assert got[fold].shape == exp.shape and rel_err(got[fold], exp) < TIGHT, (dt, fold)
            assert rel_err(got[1], got[0]) < 2e-6, dt
    finally:
        _lib.set_option(_lib.OPT_FE_FOLD, old)


@pytest.mark.parametrize("up,dn", [(13, 119), (3, 7), (5, 4), (16, 15), (2, 9)])
def test_front_end_banded_row_split_gives_the_same_samples(up, dn):
    """PRC_OPT_FE_BALANCE > 0 (an A/B option, read when a front-end plan is made): the group kernel's wavefronts take runs of
    tap rows of about equal cost and multiply only the output columns their rows reach -- prefix / suffix windows rounded out
    to whole SGPR pairs.  Same samples as the equal split to float32 rounding (the sums are split differently), the same
    as the oracle, one and two channels."""
    import torch
    from passiveradar_amd import _lib
    from passiveradar_amd.stream import HipBackend
    rng = np.random.default_rng(up * 77 + dn)
    n_in, nblk = 64 * dn + 37, 3
    ra, rb = (rng.integers(-128, 127, 2 * n_in * nblk, endpoint=True).astype(np.int8) for _ in range(2))
    args = (2 * n_in, 100_000, 2_400_000, up, dn)
    exp = O.front_end(rb, *args)
    old = _lib.get_option(_lib.OPT_FE_BALANCE)
    got = {}
    try:
        for bal in (0, 8, 34, 200):
            _lib.set_option(_lib.OPT_FE_BALANCE, bal)
            be = HipBackend(4096, 16, 32, 2.6e5, batch=4, clutter=None)          # a fresh plan: the option is read at creation
            one = be.front_end(rb, *args, max_blocks=2)
            two = be.front_end2(ra, rb, *args, max_blocks=2)[1]
            torch.cuda.synchronize()
            assert torch.equal(one, two), bal
            got[bal] = one.cpu().numpy()
            assert rel_err(got[bal], exp) < TIGHT, bal
    finally:
        _lib.set_option(_lib.OPT_FE_BALANCE, old)
    for bal in (8, 34, 200):
        assert rel_err(got[bal], got[0]) < 2e-6, bal
