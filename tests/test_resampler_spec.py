"""Specification tests of the resampler stand-in (oracle side).

r8brain-free-src is absent from the reference tree, so the two resamplers are
"parity unpinned" (SURVEY.md 8c); they are accepted on these specification
tests: rate law, unity DC gain, pass-band flatness, stop-band rejection,
latency-compensated linear phase, chunking independence."""
import numpy as np
import pytest

import oracle_py as ora


def _run(rs, x, blk):
    out = [rs.process(x[i:i + blk]) for i in range(0, len(x), blk)]
    return np.concatenate(out), [len(o) for o in out]


@pytest.mark.parametrize("fin,fout,att", [(10e6, 384e3, 140.0), (1e6, 384e3, 140.0), (384e3, 48e3, 180.0)])
def test_design_and_rate_law(fin, fout, att):
    rs = ora.Resampler(fin, fout, att)
    info = rs.info()
    assert info["L"] * fin == info["M"] * fout
    assert info["LB"] * (fin / info["D"]) == info["MB"] * fout
    n = 200000
    x = np.ones(n)
    y, counts = _run(rs, x, 65536 if fin > 2e6 else 2048)
    # output-count law: about n*L/M minus the look-ahead still pending
    expect = n * fout / fin
    assert expect - (info["NA"] / info["D"] + info["TB"]) * fout / (fin / info["D"]) - 2 <= len(y) <= expect + 1
    # unity DC gain once the start-up transient has passed
    assert np.max(np.abs(y[200:] - 1.0)) < 1e-6


def test_if_block_counts_10M():
    rs = ora.Resampler(10e6, 384e3, 140.0)
    x = np.zeros(65536)
    counts = [len(rs.process(x)) for _ in range(40)]
    assert set(counts[1:]) == {2516, 2517}
    assert abs(np.mean(counts[1:]) - 65536 * 0.0384) < 0.05


def test_chunking_independence():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(50000)
    a, _ = _run(ora.Resampler(1e6, 384e3, 140.0), x, 50000)
    b, _ = _run(ora.Resampler(1e6, 384e3, 140.0), x, 777)
    np.testing.assert_array_equal(a, b[:len(a)])
    assert len(b) == len(a)


@pytest.mark.parametrize("fin,fout,att", [(10e6, 384e3, 140.0), (384e3, 48e3, 180.0)])
def test_passband_tone_amplitude_and_alignment(fin, fout, att):
    """A pass-band tone comes out with unit gain and zero delay (output k sits at
    input time k*M/L): latency-compensated linear phase."""
    f0 = 0.8 * 0.885 * fout / 2
    n = int(fin * 0.02)
    t = np.arange(n) / fin
    rs = ora.Resampler(fin, fout, att)
    y = rs.process(np.cos(2 * np.pi * f0 * t))
    k = np.arange(len(y))
    ref = np.cos(2 * np.pi * f0 * k / fout)
    s = 400
    # IF class (140 dB): the equiripple stage A holds the pass band to +-0.0005 dB (+-6e-5); the Kaiser designs to 2e-6
    assert np.max(np.abs(y[s:] - ref[s:])) < (6e-5 if att <= 150.0 else 2e-6)


@pytest.mark.parametrize("fin,fout,att,floor_db", [(10e6, 384e3, 140.0, -135.0), (384e3, 48e3, 180.0, -170.0)])
def test_stopband_rejection(fin, fout, att, floor_db):
    """Tones that would alias into the protected band are rejected to the design floor."""
    fpass = 0.885 * fout / 2
    n = max(int(fin * 0.01), 60000)
    t = np.arange(n) / fin
    worst = -400.0
    for f in (fout - fpass + 1.0, fout + 0.3 * fpass, 2 * fout - 0.5 * fpass, fin / 2 * 0.9):
        if f >= fin / 2:
            continue
        rs = ora.Resampler(fin, fout, att)
        y = rs.process(np.cos(2 * np.pi * f * t))
        y = y[500:]
        # measure only what lands inside the protected band |f| <= fpass
        spec = np.fft.rfft(y * np.blackman(len(y))) / (np.sum(np.blackman(len(y))) / 2)
        freqs = np.fft.rfftfreq(len(y), 1 / fout)
        lvl = 20 * np.log10(np.max(np.abs(spec[freqs <= fpass])) + 1e-300)
        worst = max(worst, lvl)
    assert worst < floor_db


def test_if_resampler_complex_lockstep():
    import siggen
    x = siggen.fm_stereo_iq(3 * 65536, 10e6)
    r = ora.IfResampler(10e6, 384e3)
    y = np.concatenate([r.process(b) for b in siggen.blocks(x, 65536)])
    assert y.dtype == np.complex64
    # FM signal keeps its constant envelope (0.3) through the front end
    assert np.abs(np.abs(y[300:]).mean() - 0.3) < 1e-3


@pytest.mark.parametrize("fin,fout", [(10e6, 384e3), (30e6, 384e3), (40e6, 384e3), (61.44e6, 384e3), (2e6, 48e3),
                                      (2.048e6, 384e3), (3.2e6, 384e3), (1.92e6, 48e3), (912e3, 48e3), (6e6, 384e3)])
def test_equiripple_stage_a_rule_holds_across_shapes(fin, fout):
    """Stage A of the IF class is an equiripple design of 0.68 x the Kaiser length (a fixed formula, not a search: product
    and oracle cannot pick different lengths) -- where that design meets the class; the design code checks its response
    (every stop band <= -140 dB, ripple <= 0.0012 dB peak to peak) and keeps the Kaiser window where it does not (D = 28
    at 3.6 MHz -> 48 kHz misses by 0.2 dB, D = 61 too).  Measured directly on the PRODUCT's taps (host arithmetic): the
    pass band stays within 0.0012 dB of unity and every alias of the protected band is 140 dB down, for D = 2 ... 61."""
    import importlib
    fmr = importlib.import_module("airspy-fmradion_amd")
    h, d = fmr.design_taps(fin, fout, 140.0, 0)
    D = d["D"]
    assert D >= 2 and len(h) == d["NA"] and np.array_equal(h, h[::-1]) and abs(h.sum() - 1.0) < 1e-12
    mid, fp = fin / D, 0.885 * fout / 2
    fstop = fout - fp
    n = np.arange(len(h))
    resp = lambda f: np.abs(np.exp(-2j * np.pi * np.outer(f, n) / fin) @ h)
    assert np.max(np.abs(20 * np.log10(resp(np.linspace(0, fp, 1501))))) < 1.2e-3
    worst = -400.0
    for k in range(1, D // 2 + 1):
        lo, hi = k * mid - fstop, min(k * mid + fstop, fin / 2)
        if lo >= fin / 2:
            break
        worst = max(worst, 20 * np.log10(resp(np.linspace(lo, hi, 601)).max()))
    assert worst < -139.95           # (the design code checks 65 points per band, this test 601)
    # 0.68 x the Kaiser length, or the Kaiser length itself
    dw = 2 * np.pi * ((mid - fstop) - fp) / fin
    nk = int(np.ceil((140.0 - 7.95) / (2.285 * dw))) + 1
    nk += nk % 2 == 0
    allowed = []
    for pc in range(68, 97, 4):      # 0.68 N, then 4 % of N longer until the response meets the class; else Kaiser
        ne = (nk * pc + 99) // 100
        allowed.append(ne + (ne % 2 == 0))
    assert len(h) in allowed + [nk]
    if (fin, fout) in ((10e6, 384e3), (6e6, 384e3), (2e6, 48e3)):      # the shapes that are in use take the shortest design
        assert len(h) == allowed[0]


def test_very_large_decimation_keeps_the_kaiser_stage_a():
    """Above D = 78 the exchange is not attempted (40 bands is its limit): the Kaiser design of rounds 1 and 2 stays."""
    import importlib
    fmr = importlib.import_module("airspy-fmradion_amd")
    h, d = fmr.design_taps(10e6, 48e3, 140.0, 0)
    assert d["D"] == 80
    fp = 0.885 * 24e3
    n = np.arange(len(h))
    H = np.abs(np.exp(-2j * np.pi * np.outer(np.linspace(0, fp, 501), n) / 10e6) @ h)
    assert np.max(np.abs(20 * np.log10(H))) < 1e-5
