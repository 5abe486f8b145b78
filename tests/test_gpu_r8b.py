"""The R8B resampler class (fmr_config.resampler_class = FMR_RESAMPLER_R8B): the specification of the
r8b::CDSPResampler24 the reference constructs at sfmbase/IfResampler.cpp:25-29 -- pass band 0.98 x Nyquist, stop band
from Nyquist, 180 dB -- built in the same two-stage structure (stage A 195 taps, stage B 48/125 with 3122 taps per
phase: k_ifr_poly5, a dense f32 MFMA product).  The oracle side is ora.IfResampler(fs, 384e3, 180.0, 0.98, True), the
"r8brain-class" resampler of tests/test_resampler_independent.py.

Tolerances: IF samples relative RMS < 2e-6 (fp32 data, 3317 taps), audio RMS < 1e-5 (north star) -- also with a second
station 200 kHz away at equal power, the scene in which the FAST class differs from the reference's filter by 0.058 RMS
(profiles/r02_resampler_spec_gap.json): with the R8B class product and r8brain-class oracle filter it identically.
"""
import importlib

import numpy as np
import pytest

import oracle_py as ora
import siggen

pytestmark = pytest.mark.gpu

fmr = importlib.import_module("airspy-fmradion_amd")


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2))) if a.size else 0.0


def test_r8b_if_resampler_parity():
    fs, blk, nblk = 10e6, 65536, 12
    x = siggen.fm_stereo_iq(blk * nblk, fs)
    ch = fmr.Chain(mode=fmr.MODE_NONE, input_rate=fs, enable_resampler=True, max_block_len=blk, resampler_class=fmr.RESAMPLER_R8B)
    info = ch.resampler_info()
    assert (info["D"], info["NA"], info["LB"], info["MB"], info["TB"]) == (10, 195, 48, 125, 3122)
    got = [ch.resample(b) for b in siggen.blocks(x, blk)]
    r = ora.IfResampler(fs, 384e3, 180.0, 0.98, True)
    ref = [r.process(b) for b in siggen.blocks(x, blk)]
    assert [len(g) for g in got] == [len(q) for q in ref]
    g, q = np.concatenate(got), np.concatenate(ref)
    assert len(q) > 25000
    assert rms(g - q) / rms(q) < 2e-6
    ch.close()


@pytest.mark.parametrize("scene", ["single station", "adjacent +200 kHz, equal power"])
def test_r8b_fm_stereo_end_to_end(scene, pilotcut):
    """10 MS/s FM stereo through IfResampler (R8B class) -> FmDecoder against the oracle with the r8brain-class resampler:
    the decoder is fed what the reference's decoder would be fed, neighbour included."""
    fs, blk, nblk = 10e6, 65536, 100
    n = blk * nblk
    x = siggen.fm_stereo_iq(n, fs).astype(np.complex128)
    if scene != "single station":
        t = np.arange(n) / fs
        x = x + siggen.fm_stereo_iq(n, fs, stream_id=7, sigma=0.0).astype(np.complex128) * np.exp(2j * np.pi * 200e3 * t)
    x = x.astype(np.complex64)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=fs, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=20,
                   resampler_class=fmr.RESAMPLER_R8B)
    got = []
    for i in range(0, nblk, 20):
        a, _ = ch.process_blocks(x[None, i * blk:(i + 20) * blk], [blk] * 20)
        got.append(a[0])
    got = np.concatenate(got)
    r = ora.IfResampler(fs, 384e3, 180.0, 0.98, True)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    ref = np.concatenate([fm.process(r.process(b)) for b in siggen.blocks(x, blk)])
    assert len(got) == len(ref) > 50000
    assert fm.stereo_detected() and ch.status(0).stereo_detected == 1
    assert rms(got - ref) < 1e-5
    ch.close()
