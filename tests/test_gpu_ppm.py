"""GPU parity for ppm-corrected source rates: `-r ppm` multiplies the IF rate by 1 + ppm / 1e6 before IfResampler is
built (main.cpp:708-711), so the resampling ratio is no small rational.  The product takes the fractional-phase form
of stage B (csrc/design.hpp, k_ifr_poly_frac); the oracle holds the same specification in fp64 and is itself checked
against an analytic signal in tests/test_resampler_independent.py.
Tolerances: IF samples 2e-6 relative RMS (fp32 front end), audio RMS error < 1e-5 (north star)."""
import importlib

import numpy as np
import pytest

import oracle_py as ora
import siggen
from conftest import load_filter

pytestmark = pytest.mark.gpu

fmr = importlib.import_module("airspy-fmradion_amd")


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2))) if a.size else 0.0


@pytest.mark.parametrize("fin,mode,blk,nblk", [(10e6 * (1 + 1.5e-6), "fm", 65536, 6), (10000003.7, "fm", 65536, 4),
                                               (1e6 * (1 - 37e-6), "fm", 2048, 60), (384000.4, "am", 2048, 40)])
def test_if_resampler_ppm_corrected_rate(fin, mode, blk, nblk):
    fout = 384e3 if mode == "fm" else 48e3
    x = siggen.fm_stereo_iq(blk * nblk, fin) if mode == "fm" else siggen.am_iq(blk * nblk, fin)
    if mode == "fm":
        ch = fmr.Chain(mode=fmr.MODE_NONE, input_rate=fin, enable_resampler=True, max_block_len=blk)
        got = [ch.resample(b) for b in siggen.blocks(x, blk)]
    else:
        ch = fmr.Chain(mode=fmr.MODE_AM, input_rate=fin, enable_resampler=True, max_block_len=blk,
                       filter_coeff=load_filter("jj1bdx_am_48khz_narrow"))
        got = []
        for b in siggen.blocks(x, blk):
            ch.process(b)
            got.append(ch.debug_read(0))
    info = ch.resampler_info()
    assert info["LT"] == 1024, info
    r = ora.IfResampler(fin, fout)
    ref = [r.process(b) for b in siggen.blocks(x, blk)]
    assert [len(g) for g in got] == [len(q) for q in ref]          # the output-count law, block by block
    g, q = np.concatenate(got), np.concatenate(ref)
    assert len(q) > 1000
    assert rms(g - q) / rms(q) < 2e-6
    ch.close()


def test_fm_stereo_with_ppm_offset(pilotcut):
    """`-r 1.5` on the 10 MS/s FM stereo chain: 0.72 s in calls of 12 blocks, lock included."""
    fin, blk, batch = 10e6 * (1 + 1.5e-6), 65536, 12
    nblk = 108
    x = siggen.fm_stereo_iq(blk * nblk, fin)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=fin, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=batch)
    r = ora.IfResampler(fin, 384e3)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    got, ref = [], []
    for i in range(0, nblk, batch):
        seg = x[i * blk:(i + batch) * blk]
        a, alen = ch.process_blocks(seg[None, :], [blk] * batch)
        got.append(a[0])
        rl = [fm.process(r.process(b)) for b in siggen.blocks(seg, blk)]
        assert list(alen) == [len(v) for v in rl]
        ref += rl
    got, ref = np.concatenate(got), np.concatenate(ref)
    assert len(got) == len(ref) > 60000
    st = ch.status()
    assert st.stereo_detected == int(fm.stereo_detected()) == 1
    assert rms(got - ref) < 1e-5
    ch.close()
