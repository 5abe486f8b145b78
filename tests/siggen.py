"""Deterministic synthetic IQ generators.

All outputs are complex64 (interleaved float32 I/Q, what FileSource delivers:
sfmbase/FileSource.cpp:514-528).  Pure numpy; no reference code involved.
"""
import numpy as np


def _noise(n, sigma, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * sigma


def fm_stereo_mpx(t, stream_id=0, pilot=0.10):
    """MPX = 0.45(L+R) + pilot*sin(wp t) + 0.45(L-R) sin(2 wp t), |MPX| <= 1."""
    fl = 1000.0 + 10.0 * stream_id
    fr = 400.0 + 10.0 * stream_id
    left = np.sin(2 * np.pi * fl * t)
    right = np.sin(2 * np.pi * fr * t)
    th = 2 * np.pi * 19000.0 * t
    return 0.45 * (left + right) + pilot * np.sin(th) + 0.45 * (left - right) * np.sin(2 * th)


def fm_stereo_iq(n, fs, stream_id=0, amplitude=0.3, sigma=1e-3, n0=0, phase0=0.0, pilot=0.10):
    """S-FMst: FM stereo at sample rate fs, 75 kHz peak deviation."""
    t = (n0 + np.arange(n, dtype=np.float64)) / fs
    mpx = fm_stereo_mpx(t, stream_id, pilot)
    ph = phase0 + 2 * np.pi * 75000.0 / fs * np.cumsum(mpx)
    x = amplitude * np.exp(1j * ph)
    if sigma > 0:
        x = x + _noise(n, sigma, 1 + stream_id)
    return x.astype(np.complex64)


def fm_mono_iq(n, fs, tone=1000.0, dev=50000.0, amplitude=0.5, sigma=1e-3, seed=1):
    """S-FMmono: single audio tone, mono FM."""
    t = np.arange(n, dtype=np.float64) / fs
    ph = 2 * np.pi * dev / fs * np.cumsum(np.sin(2 * np.pi * tone * t))
    x = amplitude * np.exp(1j * ph)
    if sigma > 0:
        x = x + _noise(n, sigma, seed)
    return x.astype(np.complex64)


def am_iq(n, fs, offset=37.0, tone=1000.0, depth=0.5, level=0.1, sigma=1e-4, seed=3):
    """S-AM: carrier offset +37 Hz, envelope level*(1+depth sin)."""
    t = np.arange(n, dtype=np.float64) / fs
    env = level * (1 + depth * np.sin(2 * np.pi * tone * t))
    x = env * np.exp(2j * np.pi * offset * t)
    if sigma > 0:
        x = x + _noise(n, sigma, seed)
    return x.astype(np.complex64)


def nbfm_iq(n, fs, tone=1000.0, dev=3000.0, offset=120.0, level=0.2, sigma=1e-4, seed=5):
    """S-NBFM: voice-band tone at +-dev Hz deviation, carrier offset +offset Hz."""
    t = np.arange(n, dtype=np.float64) / fs
    phase = 2 * np.pi * offset * t - (dev / tone) * np.cos(2 * np.pi * tone * t)
    x = level * np.exp(1j * phase)
    if sigma > 0:
        x = x + _noise(n, sigma, seed)
    return x.astype(np.complex64)


def two_ray(x, delay, gain=0.35, angle=1.1, renorm=True):
    """S-MP: x[n] + gain e^{j angle} x[n-delay], optionally renormalised to the input RMS."""
    x = np.asarray(x, dtype=np.complex128)
    y = x.copy()
    y[delay:] += gain * np.exp(1j * angle) * x[:-delay]
    if renorm:
        y *= np.sqrt(np.mean(np.abs(x) ** 2) / np.mean(np.abs(y) ** 2))
    return y.astype(np.complex64)


def tone_iq(n, fs, f, amplitude=1.0, phase=0.0):
    t = np.arange(n, dtype=np.float64) / fs
    return (amplitude * np.exp(1j * (2 * np.pi * f * t + phase))).astype(np.complex64)


def blocks(x, blk):
    for i in range(0, len(x), blk):
        yield x[i:i + blk]
