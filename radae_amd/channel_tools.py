"""Host-side channel tooling: Watterson/Doppler-spread sample generator and synthetic features.

Restates (numpy only, no Octave/scipy dependency at run time)
  * `/root/reference/doppler_spread.m:7-50`  -- Gaussian-PSD filtered complex noise at a low
    sample rate, linearly interpolated up to Fs,
  * `/root/reference/multipath_samples.m:10-31` -- channel presets (mpg/mpp/mpd/lmr60), the rate-Rs H matrix (:33-40) and the
    `hf_gain = 1/sqrt(var(G1)+var(G2))` normalisation,
and the synthetic 20-dim vocoder-feature generator fixed in SURVEY.md section 8(d).

G is an *input* of the channel model (`radae/radae.py:529-539` takes it as a tensor).  The FIR design is
`fir2`'s frequency-sampling recipe, pinned to rounding on scipy.signal.firwin2 (Octave cannot be run here).
"""
from __future__ import annotations

import math

import numpy as np

PRESETS = {  # multipath_samples.m:10-21  (doppler spread Hz, path delay s)
    "mpg": (0.1, 0.5e-3),
    "mpp": (1.0, 2.0e-3),
    "mpd": (2.0, 4.0e-3),
    # land mobile radio, 60 km/h at 450 MHz (multipath_samples.m:17-21): fd = 450e6 * (60e3 / 3600 / 3e8) = 25 Hz, spread = 2 fd (50.00000000000001 in doubles, as in Octave: lowFs becomes 501 -> M = 15); BBFM.md:37 makes the
    # BBFM model's |H| file from it (Rs = 2000, Nc = 1): multipath_h() below
    "lmr60": (2.0 * 450e6 * (60 * 1e3 / 3600 / 3e8), 200e-6),
}


def fir2_from_gaussian_psd(spread_hz: float, low_fs: float, ntaps: int = 100) -> np.ndarray:
    """The Doppler-spread filter's taps by `fir2`'s own recipe as `doppler_spread.m:27-29` calls it
    (`fir2(Ntaps-1, x/(lowFs/2), y)`): the Gaussian sampled at the 51 points 0 : lowFs/100 : lowFs/2, LINEARLY
    interpolated onto a 513-point grid, half-sample linear phase (an even number of taps: type II), inverse
    FFT, Hamming window.  Equal to scipy.signal.firwin2(100, f, m, nfreqs=513) to rounding
    (tests/test_host_cpu.py::test_doppler_filter_design_against_scipy_firwin2) -- an independent
    implementation of the same algorithm; Octave itself cannot be run here.  Since round 5 this is THE design:
    doppler_plan / doppler_spread / multipath_g, the device generator's taps, every golden fixture that carries
    a generated G and bench.py's workload use it (rounds 1-4 sampled the Gaussian on the fine grid directly,
    0.24 % of the largest tap away from it)."""
    sigma = spread_hz / 2.0
    npt = 512
    x = np.arange(51) * low_fs / 100.0
    y = (1.0 / (sigma * math.sqrt(2 * math.pi))) * np.exp(-(x ** 2) / (2 * sigma * sigma))
    mag = np.interp(np.linspace(0.0, 1.0, npt + 1), x / (low_fs / 2.0), y)
    k = np.arange(npt + 1)
    spec = mag * np.exp(-1j * math.pi * k * (ntaps - 1) / (2.0 * npt))
    return np.fft.irfft(spec, 2 * npt)[:ntaps] * np.hamming(ntaps)


def doppler_plan(spread_hz: float, fs: int, nsam: int):
    """(FIR taps, Fs/lowFs ratio, low-rate sample count) doppler_spread() uses: the inputs of the device generator."""
    low_fs = math.ceil(10 * spread_hz)
    m = fs / low_fs
    if m != math.floor(m):
        m = math.floor(m)
        low_fs = fs / m
    m = int(m)
    return fir2_from_gaussian_psd(spread_hz, low_fs, 100), m, max(math.ceil(nsam / m), 2)


def doppler_spread(spread_hz: float, fs: int, nsam: int, rng: np.random.Generator) -> np.ndarray:
    """doppler_spread.m:7-50. Returns nsam complex128 samples at rate fs."""
    low_fs = math.ceil(10 * spread_hz)
    ntaps = 100
    m = fs / low_fs
    if m != math.floor(m):
        m = math.floor(m)
        low_fs = fs / m
    m = int(m)
    nsam_low = max(math.ceil(nsam / m), 2)
    b = fir2_from_gaussian_psd(spread_hz, low_fs, ntaps)
    x = rng.standard_normal(nsam_low + ntaps) + 1j * rng.standard_normal(nsam_low + ntaps)
    y = np.convolve(x, b)[: nsam_low + ntaps][ntaps:]
    # linear interpolation (with extrapolation past the last low-rate point), Octave 1-based
    # abscissae 1, 1+M, ... map to 0-based sample index n -> position n / M
    pos = np.arange(nsam) / m
    i0 = np.minimum(np.floor(pos).astype(np.int64), nsam_low - 2)
    frac = pos - i0
    return y[i0] + (y[i0 + 1] - y[i0]) * frac


def multipath_g(channel: str, fs: int, nsam: int, seed: int) -> np.ndarray:
    """(nsam, 2) complex64 [G1, G2] already multiplied by hf_gain, i.e. what
    `inference.py:160-171` hands to `RADAE.forward` after reading a g_*.f32 file."""
    spread, _delay = PRESETS[channel]
    rng = np.random.default_rng(seed)
    g1 = doppler_spread(spread, fs, nsam, rng)
    g2 = doppler_spread(spread, fs, nsam, rng)
    hf_gain = 1.0 / math.sqrt(np.var(g1) + np.var(g2))
    return (hf_gain * np.stack([g1, g2], axis=1)).astype(np.complex64)


def multipath_h(channel: str, fs: int, rs: int, nc: int, nsym: int, seed: int, complex_: bool = False) -> np.ndarray:
    """The rate-Rs channel matrix of `multipath_samples.m:25-40, :73-80` (what it writes to H_fn): H[t][c] = hf_gain (G1[t M] + G2[t M] exp(-j 2 pi c d Rs)),
    M = Fs / Rs, (nsym, nc); magnitudes unless complex_ (the script's H_complex).  `multipath_samples("lmr60", 8000, 2000, 1, 10, "h_lmr60.f32")` (BBFM.md:37)
    = multipath_h("lmr60", 8000, 2000, 1, 20000, seed)."""
    m = fs // rs
    assert m * rs == fs
    g = multipath_g(channel, fs, (nsym - 1) * m + 1, seed).astype(np.complex128)
    d = PRESETS[channel][1]
    h = g[::m, 0][:nsym, None] + g[::m, 1][:nsym, None] * np.exp(-2j * np.pi * np.arange(nc)[None, :] * d * rs)
    return h.astype(np.complex64) if complex_ else np.abs(h).astype(np.float32)


def synth_features(seed: int, nframes: int, stride: int = 36) -> np.ndarray:
    """SURVEY.md 8(d) synthetic vocoder features: AR(1) process x[t]=0.9x[t-1]+0.436 N(0,1) over
    20 dims, f0=4x0, f1..17=x1..17, f18=0.5x18, f19=clip(0.3x19,+-0.5); zero padded to `stride`."""
    rng = np.random.default_rng(seed)
    e = rng.standard_normal((nframes, 20))
    x = np.zeros((nframes, 20))
    prev = np.zeros(20)
    for t in range(nframes):
        prev = 0.9 * prev + 0.436 * e[t]
        x[t] = prev
    f = x.copy()
    f[:, 0] *= 4.0
    f[:, 18] *= 0.5
    f[:, 19] = np.clip(0.3 * x[:, 19], -0.5, 0.5)
    out = np.zeros((nframes, stride), dtype=np.float32)
    out[:, :20] = f.astype(np.float32)
    return out
