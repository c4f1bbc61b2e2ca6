"""Deterministic synthetic two-channel IQ scenes.

The reference ships no data; tests, goldens and bench.py all draw their inputs
from this generator so that the build container and the GPU box see identical
arrays without shipping them.  Counter-based Philox stream keyed by the seed.

    ref[n] : unit-variance complex white Gaussian, complex64
    srv[n] = 1.0 ref[n-2] + 0.3 ref[n-9] + 0.1 ref[n-40]            (direct path + clutter, 0 Hz)
             + sum_m A_m ref[n-d_m] exp(j 2 pi f_m n / Fs)          (moving targets)
             + noise_amp * noise[n]
Delays are circular shifts inside the generated block.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 20260926 * 1000

DEFAULT_CLUTTER = ((2, 1.0), (9, 0.3), (40, 0.1))


def default_targets(rangeBins):
    """(delay samples, Doppler Hz, amplitude) triples of SURVEY 8d, clipped to the lag span."""
    return ((min(60, max(rangeBins - 8, 1)), 80.0, 0.01),
            (max(rangeBins // 2, 1), -35.0, 0.003),
            (max(rangeBins - 5, 1), 120.0, 0.003))


def scene_seed(cfg: int, channel: int = 0) -> int:
    return SEED_BASE + cfg * 10 + channel


def _cwhite(gen, n):
    x = gen.standard_normal(2 * n, dtype=np.float32) * np.float32(np.sqrt(0.5))
    return x.view(np.complex64)


def white_reference(n, seed):
    gen = np.random.Generator(np.random.Philox(key=seed))
    return _cwhite(gen, n)


def make_scene(n, sample_rate, rangeBins, seed, targets=None, clutter=DEFAULT_CLUTTER,
               noise_amp=0.003, colour=None):
    """Return (ref, srv) complex64 arrays of length n.

    colour: optional short real FIR applied (circularly) to the white reference to
    make a mildly coloured illuminator (used for the "mildly coloured" LS cases).
    """
    gen = np.random.Generator(np.random.Philox(key=seed))
    ref = _cwhite(gen, n)
    if colour is not None:
        acc = np.zeros(n, dtype=np.complex128)
        for i, c in enumerate(colour):
            acc += c * np.roll(ref, i)
        ref = (acc / np.sqrt(np.sum(np.square(colour)))).astype(np.complex64)
    noise = _cwhite(gen, n)
    if targets is None:
        targets = default_targets(rangeBins)
    acc = np.zeros(n, dtype=np.complex128)
    for d, a in clutter:
        acc += a * np.roll(ref, d)
    t = np.arange(n, dtype=np.float64) / float(sample_rate)
    for d, fd, a in targets:
        acc += a * np.roll(ref, d) * np.exp(2j * np.pi * fd * t)
    acc += noise_amp * noise
    return ref, acc.astype(np.complex64)


def make_multi_scene(n, sample_rate, rangeBins, seeds, **kw):
    """Several illuminators against ONE surveillance channel: independent white
    reference channels, srv = sum of each channel's scene.  Returns ([ref_i], srv), complex64."""
    refs, acc = [], np.zeros(n, dtype=np.complex128)
    for sd in seeds:
        r, s = make_scene(n, sample_rate, rangeBins, sd, **kw)
        refs.append(r)
        acc += s
    return refs, acc.astype(np.complex64)


def make_stream(nchunks, chunk, sample_rate, rangeBins, seed, **kw):
    """A contiguous IF stream of nchunks*chunk samples (one scene; delays wrap at the ends)."""
    return make_scene(nchunks * chunk, sample_rate, rangeBins, seed, **kw)


def expected_peak_cell(delay, doppler_hz, n, sample_rate, rangeBins, freqBins):
    """(row, col) where fast_xambg puts an echo srv[n]=ref[n-delay] e^{+j2 pi fd n/Fs}:
    range axis reversed, Doppler axis mirrored."""
    cpi = n / float(sample_rate)
    row = int(round(freqBins / 2 - doppler_hz * cpi)) % freqBins
    return row, rangeBins - delay


def make_fm_scene(n, sample_rate, rangeBins, seed, deviation=75e3, audio_bw=15e3, **kw):
    """Report-only scene of SURVEY 8d: the reference channel is an FM broadcast-like signal (band-limited
    Gaussian audio, ``deviation`` Hz peak-ish deviation, unit amplitude) instead of white noise.  Its
    autocorrelation matrix is badly conditioned (1e6..1e7), so element-wise parity against the
    float32-summing reference is not meaningful -- clutter suppression in dB is."""
    gen = np.random.Generator(np.random.Philox(key=seed))
    spec = np.fft.rfft(gen.standard_normal(n))
    f = np.fft.rfftfreq(n, 1.0 / sample_rate)
    spec[f > audio_bw] = 0.0
    audio = np.fft.irfft(spec, n)
    audio /= 3.0 * np.std(audio)                      # +-1 at three sigma
    phase = 2 * np.pi * deviation * np.cumsum(audio) / sample_rate
    ref = np.exp(1j * phase).astype(np.complex64)
    noise = _cwhite(gen, n)
    targets = kw.get("targets", default_targets(rangeBins))
    acc = np.zeros(n, dtype=np.complex128)
    for d, a in kw.get("clutter", DEFAULT_CLUTTER):
        acc += a * np.roll(ref, d)
    t = np.arange(n, dtype=np.float64) / float(sample_rate)
    for d, fd, a in targets:
        acc += a * np.roll(ref, d) * np.exp(2j * np.pi * fd * t)
    acc += kw.get("noise_amp", 0.003) * noise
    return ref, acc.astype(np.complex64)


def make_raw_stream(nchunks, input_chunk_length, input_sample_rate, offset_freq, seed, channel_bw=200e3,
                    clutter=((18, 1.0), (83, 0.3), (366, 0.1)),
                    targets=((549, 80.0, 0.03), (1190, -35.0, 0.01)), sigma=24.0, noise_amp=0.01):
    """Raw two-channel recordings as main.py:44-112 reads them: interleaved int8 I,Q scalars at ``input_sample_rate``,
    ``input_chunk_length`` scalars per block and channel.  The illuminator is Gaussian noise band-limited to
    ``channel_bw`` sitting ``offset_freq`` BELOW the recording's centre (PRconfig.yaml: channel 101.9 MHz in a recording
    centred on 102.0 MHz), so that the reference's frequency_shift(+offset_freq) brings it to baseband; the surveillance
    channel is delayed copies (delays in raw samples: 9.15 per IF sample at 13/119), two moving echoes and receiver
    noise.  Both channels are quantised to int8 independently.  Returns (raw_ref, raw_srv), int8 [nchunks * icl]."""
    from math import pi
    n = nchunks * (int(input_chunk_length) // 2)
    gen = np.random.Generator(np.random.Philox(key=seed))
    white = gen.standard_normal(2 * n).view(np.complex128) * np.sqrt(0.5)
    # windowed-sinc low-pass to the channel bandwidth (two-sided), 65 taps
    m = np.arange(-32, 33)
    fcut = 0.5 * channel_bw / float(input_sample_rate)
    h = 2 * fcut * np.sinc(2 * fcut * m) * np.hamming(65)
    base = np.convolve(white, h, mode="same")
    base /= np.sqrt(np.mean(np.abs(base) ** 2))
    t = np.arange(n, dtype=np.float64) / float(input_sample_rate)
    down = np.exp(-2j * pi * float(offset_freq) * t)
    noise_r = gen.standard_normal(2 * n).view(np.complex128) * np.sqrt(0.5)
    noise_s = gen.standard_normal(2 * n).view(np.complex128) * np.sqrt(0.5)
    acc = np.zeros(n, dtype=np.complex128)
    for d, a in clutter:
        acc += a * np.roll(base, d)
    for d, fd, a in targets:
        acc += a * np.roll(base, d) * np.exp(2j * pi * fd * t)

    def quantise(x):
        iq = np.empty(2 * n, dtype=np.float64)
        iq[0::2], iq[1::2] = x.real, x.imag
        return np.clip(np.rint(iq * sigma), -127, 127).astype(np.int8)
    raw_ref = quantise((base + noise_amp * noise_r) * down)
    raw_srv = quantise((acc + noise_amp * noise_s) * down)
    return raw_ref, raw_srv


def raw_checksum(raw):
    """(sum, sum of squares, CRC32) of an int8 recording: a regenerated stream is checked against its golden's"""
    import zlib
    a = np.asarray(raw).astype(np.int64)
    return int(a.sum()), int((a * a).sum()), int(zlib.crc32(np.ascontiguousarray(raw).tobytes()))
