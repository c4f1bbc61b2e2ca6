"""Randomised differential check of the drop-ins against the oracle (GPU box): random shapes around the edges the
kernels switch on (FFT vs direct CAF, lag blocking, cached vs per-bin LS chain, NLMS taps-per-lane buckets)."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")   # run from the repo root: python tests/fuzz_parity.py [seed] [seconds] [threads] [log.md]
from oracle import np_oracle as O, c_oracle
from passiveradar_amd import scene
from passiveradar_amd.clutter_removal import LS_Filter, LS_Filter_Multiple, LS_Filter_Toeplitz, NLMS_filter
from passiveradar_amd.range_doppler_processing import fast_xambg, fast_xambg_multi
from passiveradar_amd.signal_utils import decimate_iir, deinterleave_IQ, find_channel_offset, frequency_shift, front_end, resample, xcorr
from passiveradar_amd.target_detection import CFAR_2D, CFAR_2D_abs

import os, threading
# Everything the workers use is imported HERE, on the main thread, before they start.  Round 6: two runs in twenty-six did
# not end -- a worker's first `import torch` (dlopen of torch's HIP libraries: holds the GIL and the dynamic loader's lock
# while their static constructors register with the HIP runtime) met another worker inside prc_caf_plan_create (holds a
# HIP runtime lock while it loads a code object, which needs the loader's lock): lock-order inversion between the two, and
# the remaining workers waited for the GIL for ever (stacks: profiles/r06_fuzz_watchdog_stacks.md).
try:
    import torch                                                         # noqa: F401
    from passiveradar_amd.stream import HipBackend, StreamProcessor      # noqa: F401
except ImportError:
    pass
from passiveradar_amd import clutter_removal as _cr
_cr.set_default_ls_method(int(os.environ.get("PR_FUZZ_LS_METHOD", "0")))      # e.g. 4: the LS kinds on the 4096-point chain
KINDS = [int(x) for x in os.environ["PR_FUZZ_KINDS"].split(",")] if os.environ.get("PR_FUZZ_KINDS") else None
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
nthreads = int(sys.argv[3]) if len(sys.argv) > 3 else 1        # dask-style concurrent callers
logpath = sys.argv[4] if len(sys.argv) > 4 else None           # markdown record of the run (committed under profiles/)
counts = {}
rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-30))
worst, fails, ncase = {}, [], 0
current = {}
lock = threading.Lock()
t0 = time.time()
def note(kind, err, tol, desc):
    global ncase
    with lock:
        ncase += 1
        counts[kind] = counts.get(kind, 0) + 1
        worst[kind] = max(worst.get(kind, 0.0), err)
        if not (err < tol):
            fails.append((kind, err, desc))
def worker(wseed):
  rng = np.random.default_rng(wseed)
  while time.time() - t0 < budget:
      k = rng.integers(0, 26) if KINDS is None else int(rng.choice(KINDS))
      current[wseed] = (int(k), ncase, round(time.time() - t0, 1))          # what this thread is in (printed if it never comes back)
      if k == 20:     # power-of-two Doppler bin counts: the column-FFT Doppler kernel (256 .. 4096), ragged column tiles
          F = int(rng.choice([256, 512, 1024, 2048, 4096])); q = int(rng.integers(4, 40)); N = F * q + int(rng.integers(0, F))
          R = int(rng.integers(1, min(700, N // 2 - 1)))
          ref, srv = scene.make_scene(N, 1e5, min(R, 200), int(rng.integers(1 << 30)))
          w = None if rng.random() < 0.5 else np.kaiser(N, 5.0)
          note("caf_column_doppler", rel(fast_xambg(ref, srv, R, F, N, w), O.fast_xambg(ref, srv, R, F, N, w)), 2e-5, ("cafcol", N, R, F, w is not None))
      elif k == 24:   # CFAR of the complex map in one kernel (|X| on the tile load), either kernel form
          H, W = int(rng.integers(3, 300)), int(rng.integers(3, 300)); fw = int(rng.integers(3, 19)); gw = int(rng.integers(0, fw - 1))
          X = (rng.standard_normal((H, W)) + 1j * rng.standard_normal((H, W))).astype(np.complex64)
          from passiveradar_amd import _lib
          _lib.set_option(_lib.OPT_CFAR_METHOD, int(rng.integers(0, 2)))
          try:
              note("cfar_abs", rel(CFAR_2D_abs(X, fw, gw), O.CFAR_2D(np.abs(X), fw, gw)), 2e-5, ("cfarabs", H, W, fw, gw))
          finally:
              _lib.set_option(_lib.OPT_CFAR_METHOD, 0)
      elif k == 25:   # both channels of a recording through the front end in one launch == one launch per channel, and the oracle
          import torch
          from math import gcd
          from passiveradar_amd.stream import HipBackend
          up, dn = int(rng.integers(1, 20)), int(rng.integers(2, 130)); nb = int(rng.integers(1, 6)); dt = str(rng.choice(["int8", "uint8", "int16", "float32"]))
          n_in = int(rng.integers(2 * dn + 3, 20000)); foff = int(rng.choice([100000, 37500, 2400]))
          if up // gcd(up, dn) != dn // gcd(up, dn):
              ra, rb = ((rng.standard_normal(2 * n_in * nb) * 30).astype(dt) for _ in range(2))
              be = HipBackend(4096, 8, 16, 2.6e5, batch=2, clutter=None)
              a2, b2 = be.front_end2(ra, rb, 2 * n_in, foff, 2400000, up, dn, max_blocks=2)
              same = torch.equal(a2, be.front_end(ra, 2 * n_in, foff, 2400000, up, dn, max_blocks=2)) and torch.equal(b2, be.front_end(rb, 2 * n_in, foff, 2400000, up, dn, max_blocks=2))
              note("front_end2", max(rel(b2.cpu().numpy(), O.front_end(rb, 2 * n_in, foff, 2400000, up, dn)), 0.0 if same else 1.0), 2e-5, ("fe2", n_in, dt, up, dn, nb, foff))
      elif k == 22:   # xcorr of two signals of different lengths (signal_utils.py:29-32 accepts any two)
          n1, n2 = int(rng.integers(1, 30000)), int(rng.integers(1, 30000)); nlead = int(rng.integers(0, 60)); nlag = int(rng.integers(0, 300))
          a = scene.white_reference(n1, int(rng.integers(1 << 30))); b = scene.white_reference(n2, int(rng.integers(1 << 30)))
          g, e = xcorr(a, b, nlead, nlag), O.xcorr(a, b, nlead, nlag)
          scale = float(np.sqrt(min(n1, n2))) * 4                        # a lag's sum of min(n1, n2) unit-variance products
          note("xcorr_uneven", float(np.abs(g - e).max()) / scale if g.shape == e.shape else 1.0, 2e-5, ("xcu", n1, n2, nlead, nlag))
      elif k == 23:   # frequency_shift with one phase per sample: float64 / integer (complex128 out) or float32 (complex64 out)
          n = int(rng.integers(10, 50000)); fc = float(rng.uniform(-3e5, 3e5))
          x = scene.white_reference(n, int(rng.integers(1 << 30)))
          ph = rng.uniform(-50, 50, n).astype(rng.choice([np.float64, np.float32, np.int64]))
          g, e = frequency_shift(x, fc, 2.4e6, ph), O.frequency_shift(x, fc, 2.4e6, ph)
          note("freqshift_phases", rel(g, e) if g.dtype == e.dtype else 1.0, 2e-6, ("fsp", n, fc, str(ph.dtype)))
      elif k == 21:   # several illuminators against one surveillance channel, every mode of prc_caf_execute_multi
          nref = int(rng.integers(1, 6)); F = int(rng.choice([2, 8, 16, 256])); N = int(rng.integers(max(8192, 4 * F), 200000))
          R = int(rng.integers(2, min(4000, N // 2 - 1)))
          refs, srv = scene.make_multi_scene(N, 1e5, min(R, 200), [int(rng.integers(1 << 30)) for _ in range(nref)])
          w = None if rng.random() < 0.5 else np.kaiser(N, 5.0)
          outs = fast_xambg_multi(refs, srv, R, F, N, w, mode=str(rng.choice(["auto", "turns", "shared", "pairs"])))
          i = int(rng.integers(0, nref))
          note("caf_multi", rel(outs[i], O.fast_xambg(refs[i], srv, R, F, N, w)), 2e-5, ("cafmulti", N, R, F, nref, i))
      elif k == 15:   # wide range spans: AUTO takes the 4096-point team kernel (tails, several lag blocks, wrap)
          F = int(rng.choice([2, 4, 16, 33])); N = int(rng.integers(8192, 300000)); R = int(rng.integers(600, min(5000, N // 2 - 1)))
          ref, srv = scene.make_scene(N, 1e5, 300, int(rng.integers(1 << 30)))
          w = None if rng.random() < 0.5 else np.kaiser(N, 4.0)
          note("caf_wide", rel(fast_xambg(ref, srv, R, F, N, w), O.fast_xambg(ref, srv, R, F, N, w)), 2e-5, ("cafwide", N, R, F, w is not None))
      elif k == 16:   # long LS filters on the 4096-point team kernels (770 .. 3073 taps), linear and circular
          L = int(rng.choice([760, 790, 1024, 1500, 2047 - 10, 3063])); N = int(rng.integers(3 * L, 60000))
          ref, srv = scene.make_scene(N, 1e6, 100, int(rng.integers(1 << 30)))
          if rng.random() < 0.7:
              bins = [0.0, 2.0, -1.0][:int(rng.integers(1, 4))]
              note("ls_team", rel(LS_Filter_Multiple(ref, srv, L, 1e6, bins), O.LS_Filter_Multiple(ref, srv, L, 1e6, bins)), 1e-4, ("lsteam", N, L, bins))
          else:
              n2 = min(N, 12000)
              note("ls_team_direct", rel(LS_Filter(ref[:n2], srv[:n2], L), O.LS_Filter(ref[:n2], srv[:n2], L)), 1e-4, ("lsteamdirect", n2, L))
      elif k == 17:   # Doppler bins far from zero (beyond the Taylor range of the wrapped-sample phase), short and long blocks
          N = int(rng.integers(3000, 120000)); L = int(rng.integers(2, 100)); fs = float(rng.choice([1e4, 2.6e5]))
          bins = [0.0] + [float(b) for b in rng.uniform(-0.006 * fs, 0.006 * fs, int(rng.integers(1, 4)))]
          ref, srv = scene.make_scene(N, fs, max(L, 50), int(rng.integers(1 << 30)))
          note("ls_far_bins", rel(LS_Filter_Multiple(ref, srv, L, fs, bins), O.LS_Filter_Multiple(ref, srv, L, fs, bins)), 1e-4, ("lsfar", N, L, fs, bins))
      elif k == 18:   # NLMS with a level step of 30-60 dB in the reference (u^H u must follow exactly)
          N = int(rng.integers(1500, 8000)); L = int(rng.integers(4, 400)); mu = 0.05
          ref, srv = scene.make_scene(N, 1e4, 50, int(rng.integers(1 << 30)))
          g = np.ones(N, np.float32); a0 = int(rng.integers(N // 4, N // 2)); g[a0:a0 + int(rng.integers(50, N // 3))] = 10 ** (-float(rng.uniform(30, 60)) / 20)
          r2, s2 = (ref * g).astype(np.complex64), (srv * g + 0.001 * srv).astype(np.complex64)
          if N > L + 12:
              note("nlms_step", rel(NLMS_filter(r2, s2, L, mu), c_oracle.nlms(r2, s2, L, mu)[0]), 1e-4, ("nlmsstep", N, L, a0))
      elif k == 19:   # multi-bin LS on blocks long enough for the shared-inverse (Gohberg-Semencul) chain, config-2-like taps
          N = int(rng.integers(20000, 400000)); L = int(rng.choice([16, 100, 256, 400])); fs = float(rng.choice([2.6e5, 2.4e6]))
          bins = [0.0, 1.0, -1.0, 2.0, -2.0][:int(rng.integers(2, 6))]
          ref, srv = scene.make_scene(N, fs, min(L, 200), int(rng.integers(1 << 30)))
          note("ls_chain", rel(LS_Filter_Multiple(ref, srv, L, fs, bins), O.LS_Filter_Multiple(ref, srv, L, fs, bins)), 1e-4, ("lschain", N, L, fs, bins))
      elif k == 0:      # CAF
          F = int(rng.choice([2, 8, 16, 51, 64, 128]))
          N = int(rng.integers(max(2 * F, 600), 40000))
          R = int(rng.integers(1, min(300, N // 3)))
          ref, srv = scene.make_scene(N, 1e4, R, int(rng.integers(1 << 30)))
          win = rng.choice([None, "arr"])
          w = None if win is None else np.kaiser(N, 3.0)
          note("caf", rel(fast_xambg(ref, srv, R, F, N, w), O.fast_xambg(ref, srv, R, F, N, w)), 2e-5, ("caf", N, R, F, win))
      elif k == 1:    # LS Toeplitz / direct
          N = int(rng.integers(300, 30000)); L = int(rng.integers(1, min(200, N // 8))); peek = int(rng.integers(0, 12))
          if rng.random() < 0.03:         # beyond 5120 taps: the Levinson recursion out of a global workspace (any length)
              L = int(rng.integers(5111, 6500)); N = 3 * L + int(rng.integers(100, 4000))
          ref, srv = scene.make_scene(N, 1e4, max(min(L, 200), 50), int(rng.integers(1 << 30)))
          if rng.random() < 0.5:
              got, gt = LS_Filter_Toeplitz(ref, srv, L, peek, True); exp, et = O.LS_Filter_Toeplitz(ref, srv, L, peek, True)
              note("ls_toeplitz", max(rel(got, exp), rel(gt, et)), 1e-4, ("toep", N, L, peek))
          else:
              reg = float(rng.choice([0.0, 1.0, 10.0]))
              got = LS_Filter(ref, srv, L, reg, peek); exp = O.LS_Filter(ref, srv, L, reg, peek)
              note("ls_direct", rel(got, exp), 1e-4, ("direct", N, L, reg, peek))
      elif k == 2:    # LS multiple (cached chain when N >= 20000)
          N = int(rng.integers(2000, 60000)); L = int(rng.integers(2, 120))
          fs = float(rng.choice([1e4, 2.4e5, 2.4e6]))
          nb = int(rng.integers(1, 6)); bins = [float(b) for b in rng.integers(-3, 4, nb)]
          ref, srv = scene.make_scene(N, fs, max(L, 50), int(rng.integers(1 << 30)))
          note("ls_multiple", rel(LS_Filter_Multiple(ref, srv, L, fs, bins), O.LS_Filter_Multiple(ref, srv, L, fs, bins)), 1e-4, ("multi", N, L, fs, bins))
      elif k == 3:    # NLMS
          N = int(rng.integers(200, 6000)); L = int(rng.integers(1, 2030)); mu = float(rng.choice([0.01, 0.05, 0.2]))
          if rng.random() < 0.2:          # two / four wavefronts per stream (2049 .. 8192 taps)
              L = int(rng.integers(2030, 8180)); N = L + 10 + int(rng.integers(50, 1200))
          elif rng.random() < 0.05:       # beyond the register-resident kernels: the any-length fallback
              L = int(rng.integers(8183, 11000)); N = L + 10 + int(rng.integers(50, 600))
          ref, srv = scene.make_scene(N, 1e4, 50, int(rng.integers(1 << 30)))
          note("nlms", rel(NLMS_filter(ref, srv, L, mu), c_oracle.nlms(ref, srv, L, mu)[0]) if N > L + 10 else 0.0, 1e-4, ("nlms", N, L, mu))
      elif k == 4:    # xcorr
          N = int(rng.integers(100, 50000)); nlead = int(rng.integers(0, 40)); nlag = int(rng.integers(1, 300))
          a, b = scene.make_scene(N, 1e4, 50, int(rng.integers(1 << 30)))
          note("xcorr", rel(xcorr(a, b, nlead, nlag), O.xcorr(a, b, nlead, nlag)), 2e-5, ("xcorr", N, nlead, nlag))
      elif k == 6:    # big CAF: several lag blocks, long CPIs
          F = int(rng.choice([4, 32, 100])); N = int(rng.integers(20000, 300000)); R = int(rng.integers(200, 1200))
          ref, srv = scene.make_scene(N, 1e5, R, int(rng.integers(1 << 30)))
          note("caf_big", rel(fast_xambg(ref, srv, R, F), c_oracle.fast_xambg(ref, srv, R, F)), 2e-5, ("cafbig", N, R, F))
      elif k == 7:    # LS near the FFT-kernel limit (T - 1 <= 768) and long blocks
          N = int(rng.integers(20000, 200000)); L = int(rng.choice([200, 500, 740, 758, 759, 760, 800]))
          fs = 2.4e6; bins = [0.0, 1.0, -1.0][:int(rng.integers(1, 4))]
          ref, srv = scene.make_scene(N, fs, 100, int(rng.integers(1 << 30)))
          note("ls_long", rel(LS_Filter_Multiple(ref, srv, L, fs, bins), O.LS_Filter_Multiple(ref, srv, L, fs, bins)), 1e-4, ("lslong", N, L, bins))
      elif k == 8:    # front end pieces
          n = int(rng.integers(100, 30000))
          dt = rng.choice(["int8", "uint8", "int16", "float32"])
          raw = (rng.standard_normal(2 * n) * 30).astype(dt)
          ok = np.array_equal(deinterleave_IQ(raw), O.deinterleave_IQ(raw))
          x = O.deinterleave_IQ(raw)[:max(n // 2, 40)]
          up, dn = int(rng.integers(1, 20)), int(rng.integers(1, 130))
          e1 = rel(resample(x, up, dn), O.resample(x, up, dn))
          fc = float(rng.uniform(-2e5, 2e5)); e2 = rel(frequency_shift(x, fc, 2.4e6, 0.3), O.frequency_shift(x, fc, 2.4e6, 0.3))
          # the fused chain on raw blocks (deinterleave -> block-phase tuning -> resample), either kernel form
          from math import gcd
          from passiveradar_amd import _lib
          nb = int(rng.integers(1, 4)); n_in = max(n // nb // 2, 2 * dn + 3); foff = int(rng.choice([100000, 37500, 2400]))
          rawb = (rng.standard_normal(2 * n_in * nb) * 30).astype(dt)
          _lib.set_option(_lib.OPT_FE_METHOD, int(rng.integers(0, 2)))
          try:
              e3 = rel(front_end(rawb, 2 * n_in, foff, 2400000, up, dn, max_blocks=2), O.front_end(rawb, 2 * n_in, foff, 2400000, up, dn)) \
                  if up // gcd(up, dn) != dn // gcd(up, dn) else 0.0
          finally:
              _lib.set_option(_lib.OPT_FE_METHOD, 0)
          note("front_end", max(e1, e2, e3, 0.0 if ok else 1.0), 2e-5, ("fe", n, dt, up, dn, fc, nb, foff))
      elif k == 9:    # CFAR
          H, W = int(rng.integers(20, 300)), int(rng.integers(20, 300)); fw = int(rng.integers(3, 19)); gw = int(rng.integers(0, fw - 1))
          X = np.abs(rng.standard_normal((H, W))).astype(np.float32) + 0.1
          from passiveradar_amd import _lib
          _lib.set_option(_lib.OPT_CFAR_METHOD, int(rng.integers(0, 2)))
          try:
              note("cfar", rel(CFAR_2D(X, fw, gw), O.CFAR_2D(X, fw, gw)), 2e-5, ("cfar", H, W, fw, gw))
          finally:
              _lib.set_option(_lib.OPT_CFAR_METHOD, 0)
      elif k == 10:   # zero-phase IIR decimator
          n = int(rng.integers(28, 60000)); q = int(rng.choice([1, 2, 3, 4, 5, 8, 10]))
          x = scene.white_reference(n, int(rng.integers(1 << 30)))
          note("decimate_iir", rel(decimate_iir(x, q), O.decimate_iir(x, q)), 2e-5, ("dec", n, q))
      elif k == 11:   # CAF variants: zero-pad branch, long decimation FIR, named window, complex128 surveillance
          F = int(rng.choice([8, 16, 50])); N = int(rng.integers(2500, 20000)); R = int(rng.integers(1, 60))
          ref, srv = scene.make_scene(N, 1e4, R, int(rng.integers(1 << 30)))
          v = int(rng.integers(0, 4))
          if v == 0:
              n_in = int(rng.integers(N // 2, N)); a, b = ref[:n_in], srv[:n_in]
              note("caf_pad", rel(fast_xambg(a, b, R, F, N), O.fast_xambg(a, b, R, F, N)), 2e-5, ("cafpad", N, n_in, R, F))
          elif v == 1:
              note("caf_longfilt", rel(fast_xambg(ref, srv, R, F, N, None, False), O.fast_xambg(ref, srv, R, F, N, None, False)), 2e-5, ("caflong", N, R, F))
          elif v == 2:
              w = ("kaiser", float(rng.choice([2.0, 5.0]))) if rng.random() < 0.5 else "hann"
              note("caf_namedwin", rel(fast_xambg(ref, srv, R, F, N, w), O.fast_xambg(ref, srv, R, F, N, w)), 2e-5, ("cafwin", N, R, F, w))
          else:
              note("caf_c128", rel(fast_xambg(ref, srv.astype(np.complex128), R, F), O.fast_xambg(ref, srv.astype(np.complex128), R, F)), 2e-5, ("caf128", N, R, F))
      elif k == 12:   # NLMS warm start
          N = int(rng.integers(400, 4000)); L = int(rng.integers(1, 300)); mu = 0.05
          ref, srv = scene.make_scene(N, 1e4, 50, int(rng.integers(1 << 30)))
          t0_ = (rng.standard_normal(L + 10) + 1j * rng.standard_normal(L + 10)).astype(np.complex64) * 0.05
          if N > L + 12:
              g, gt = NLMS_filter(ref, srv, L, mu, 10, t0_, True); e, et = O.NLMS_filter(ref, srv, L, mu, 10, t0_, True)
              note("nlms_warm", max(rel(g, e), rel(gt, et)), 1e-4, ("nlmswarm", N, L))
      elif k == 13:   # block-phase frequency shift (complex128 result)
          n = int(rng.integers(10, 50000)); fc = float(rng.uniform(-3e5, 3e5)); ph = float(rng.uniform(-6, 6))
          x = scene.white_reference(n, int(rng.integers(1 << 30)))
          g = frequency_shift(x, fc, 2.4e6, np.array([ph])); e = O.frequency_shift(x, fc, 2.4e6, np.array([ph]))
          note("freqshift_block", rel(g, e) if g.dtype == e.dtype else 1.0, 2e-6, ("fsb", n, fc, ph))
      elif k == 14:   # sharded stream == unsharded stream, random geometry
          import torch
          from passiveradar_amd.stream import HipBackend, StreamProcessor
          C = int(rng.choice([4096, 6000, 8192, 24000, 30011])); nch = int(rng.integers(2, 9 if C < 20000 else 5)); R = int(rng.integers(4, 40)); F = int(rng.choice([16, 32, 64]))
          world = int(rng.integers(2, 5)); batch = int(rng.integers(1, 6))
          a, b = scene.make_stream(nch, C, 2.6e5, R, int(rng.integers(1 << 30)))
          be = HipBackend(2 * C, R, F, 2.6e5, batch=batch)
          full = StreamProcessor(be).process(a, b).cpu().numpy()
          parts = np.concatenate([StreamProcessor(be, r, world).process_local(a, b)[0].cpu().numpy() for r in range(world)])
          exp = np.moveaxis(O.process_stream(a, b, 2 * C, R, F, 2.6e5), 2, 0)
          note("stream", max(rel(parts, full), rel(full, exp)), 1e-4, ("stream", C, nch, R, F, world, batch))
      else:           # channel offset
          N = int(rng.integers(2000, 40000)); nd = int(rng.choice([1, 1, 2, 4])); nl = int(rng.integers(10, 3000)); sh = int(rng.integers(-nl // 2, nl // 2 + 1))
          a = scene.white_reference(N + 8000, int(rng.integers(1 << 30)))
          s1, s2 = a[4000:4000 + N], a[4000 - sh:4000 - sh + N]
          note("chan_offset", 0.0 if find_channel_offset(s1, s2, nd, nl) == O.find_channel_offset(s1, s2, nd, nl) else 1.0, 0.5, ("off", N, nd, nl, sh))
def guarded(wseed):
    try:
        worker(wseed)
    except Exception as e:          # a crash in one thread is a failure, not a silent exit
        import traceback
        with lock:
            fails.append(("exception", 1.0, traceback.format_exc()[-600:]))
# a case that never returns (a deadlock between caller threads, a kernel that does not end) must not look like a slow run:
# well past the budget every thread's Python stack goes to stderr and the process exits 3 (the test prints stderr)
import faulthandler
faulthandler.dump_traceback_later(budget * 2 + float(os.environ.get("PR_FUZZ_GRACE", "240")), exit=True)
def overdue():
    while True:
        time.sleep(30)
        if time.time() - t0 > budget + 60:
            print(f"[fuzz] {time.time() - t0:.0f} s, budget {budget:.0f} s: threads still in (kind, case number, started at s): {current}", file=sys.stderr, flush=True)
threading.Thread(target=overdue, daemon=True).start()
threads = [threading.Thread(target=guarded, args=(seed * 1000 + i,)) for i in range(nthreads)]
[t.start() for t in threads]
[t.join() for t in threads]
print(f"{ncase} random cases in {time.time() - t0:.0f} s; worst relative error per kind:", {k: f"{v:.1e}" for k, v in worst.items()})
print("FAILURES:", fails if fails else "none")
if logpath:
    with open(logpath, "w") as fh:
        fh.write(f"# tests/fuzz_parity.py -- randomised drop-in vs oracle run on the MI355X box\n\n"
                 f"seed {seed}, {budget:.0f} s budget, {nthreads} caller thread(s): **{ncase} cases, {len(fails)} failures**.\n"
                 "Error = max|got - want| / max|want| per case (integer kinds: 0 or 1); the bar is 1e-4 for the LS / NLMS kinds,\n"
                 "2e-5 for the CAF / helper kinds, 2e-6 for the block-phase shift.\n\n| kind | cases | worst error |\n|---|---|---|\n")
        for k in sorted(worst):
            fh.write(f"| {k} | {counts[k]} | {worst[k]:.2e} |\n")
        fh.write("\nFailures: " + (repr(fails) if fails else "none") + "\n")
sys.exit(1 if fails else 0)
