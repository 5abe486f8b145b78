"""The fused front end (k_ifr_fused) at the edges of its fp16 split -- the only arithmetic on the headline path that is
not the reference's own types.  Both FIR stages of that kernel run on the fp16 matrix cores with taps and samples as two
fp16 terms (high + low / 2048, three products, fp32 accumulate: kernels_fused.hpp FusedMfmaA / FusedB16).  The split keeps
22 bits for 6.1e-5 <= |x| <= 65504; below, the low term runs into fp16's subnormals (absolute error <= 1.5e-11); above,
the high term overflows and the tile is recomputed with fp32 tap loops (the non-finite repair path).

What the reference accepts there: IfSimpleAgc's maximum gain is 1e5 (sfmbase/FmDecode.cpp:74, IfSimpleAgc.cpp:37-57), so a
station at 1e-5 of full scale is a legal input that still reaches unit amplitude behind the AGC; FileSource hands over
un-normalised FLOAT files as they are (sfmbase/FileSource.cpp:514-528), so amplitudes of 1e3 .. 1e5 are legal too.

Each case: 10 MS/s FM stereo through IfResampler + FmDecoder -- IF samples of the fused kernel (debug-tap chain, same
kernel and arithmetic, IF-storing epilogue) within 2e-6 relative RMS of the oracle's, and the audio of the PRODUCT
configuration (discriminator epilogue, nothing but MPX and |x|^2 leaves the kernel) within 1e-5 RMS (north star).

The audio is compared where the reference's decoder works.  IfSimpleAgc starts at gain 1 with rate 1e-4
(FmDecode.cpp:74): its update factor z = 1 + r (1 - |g x|^2) turns NEGATIVE once |g x| > 100 and the gain changes sign;
at an amplitude of a few hundred it settles at a negative gain (the oracle at 200: gain -0.005, audio intact), from about
1e3 on it overflows and is reset every few samples (IfSimpleAgc.cpp:46-50), the discriminator sees a carrier whose sign
keeps flipping and the reference decodes noise (the oracle at 1e3: audio RMS 1.37 instead of 0.44, gain pinned at its
maximum) -- by its own arithmetic, and likewise for a few samples behind an upward level step of more than 100 x.  The product's discriminator reads the un-gained samples (atan2 is invariant to a POSITIVE gain,
DESIGN.md section 2) and does not reproduce that; there the IF samples -- what IfResampler hands the decoder, where the
reference has no such limit -- are what is held to the oracle.
Run on the GPU box:  python -m pytest tests -m gpu
"""
import importlib
import json
import os
import time

import numpy as np
import pytest

import oracle_py as ora
import siggen

pytestmark = pytest.mark.gpu

fmr = importlib.import_module("airspy-fmradion_amd")
BLK = 65536
REPORT = {}


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2))) if a.size else 0.0


def _report(key, **kw):
    REPORT[key] = kw
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report_levels.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


def _oracle(x, calls, pilotcut):
    r = ora.IfResampler(10e6, 384e3)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    if_ref, au_ref, pos = [], [], 0
    for ll in calls:
        ifs = []
        for n in ll:
            q = r.process(x[pos:pos + n]); pos += n
            ifs.append(q)
            au_ref.append(fm.process(q))
        if_ref.append(np.concatenate(ifs))
    return if_ref, au_ref, fm


def _product(x, calls, debug_taps):
    """The chain as the product runs it (debug_taps: the same kernel with the IF-storing epilogue, so that the IF samples
    can be read back).  Returns audio per call, block lengths, IF samples per call (debug chain), seconds per call."""
    if debug_taps:
        os.environ["FMR_DEBUG_TAPS"] = "1"
    try:
        ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, max_block_len=BLK, max_blocks=max(len(c) for c in calls))
    finally:
        os.environ.pop("FMR_DEBUG_TAPS", None)
    au, alens, ifs, secs, pos = [], [], [], [], 0
    for ll in calls:
        m = sum(ll)
        t0 = time.perf_counter()
        a, alen = ch.process_blocks(x[None, pos:pos + m], ll)
        secs.append(time.perf_counter() - t0)
        pos += m
        au.append(a[0]); alens += list(alen)
        if debug_taps:
            ifs.append(ch.debug_read(0))
    st = ch.status()
    ch.close()
    return au, alens, ifs, secs, st


def _levels_case(name, x, calls, pilotcut, audio=True):
    if_ref, au_ref, fm = _oracle(x, calls, pilotcut)
    au_ref_all = np.concatenate(au_ref)
    au_d, alens_d, ifs, secs_d, _ = _product(x, calls, debug_taps=True)
    assert alens_d == [len(a) for a in au_ref]
    # IF samples of the calls the fused kernel ran (every call here is long enough for it)
    rels = []
    for g, q in zip(ifs, if_ref):
        assert len(g) == len(q)
        assert np.isfinite(g.view(np.float32)).all(), name
        # relative to the level of the samples themselves, piece by piece: a stream whose level steps by 200 dB must be
        # right in its quiet parts too
        m = 4096
        k = len(q) // m
        num = np.sqrt(np.mean(np.abs((g - q)[:k * m].reshape(k, m)) ** 2, axis=1))
        den = np.sqrt(np.mean(np.abs(q[:k * m].reshape(k, m)) ** 2, axis=1))
        rels.append(float(np.max(num / den)))
    rep = dict(if_rel_rms_worst_piece_per_call=rels, seconds_per_call_debug_chain=[round(s, 4) for s in secs_d])
    assert max(rels) < 2e-6, (name, rels)                       # front-end tolerance (tests/test_gpu_parity.py)
    if not audio:
        _report(name, **rep)
        return None, fm, secs_d
    au_p, alens_p, _, secs, st = _product(x, calls, debug_taps=False)
    assert alens_p == alens_d
    err_p = rms(np.concatenate(au_p) - au_ref_all)
    err_d = rms(np.concatenate(au_d) - au_ref_all)
    _report(name, **rep, audio_rms_err=err_p, audio_rms_err_debug_chain=err_d, audio_rms=rms(au_ref_all),
            seconds_per_call=[round(s, 4) for s in secs], locked=int(st.stereo_detected), ref_locked=int(fm.stereo_detected()),
            agc_gain=float(st.if_agc_gain), ref_agc_gain=float(fm.get_if_agc_gain()), if_rms=float(st.if_rms), ref_if_rms=float(fm.get_if_rms()),
            pll_fallback=int(st.pll_fallback), agc_fallback=int(st.agc_fallback))
    assert err_p < 1e-5 and err_d < 1e-5, (name, err_p, err_d)   # north-star tolerance
    assert st.stereo_detected == int(fm.stereo_detected()) == 1
    assert st.if_rms == pytest.approx(fm.get_if_rms(), rel=1e-5)
    return st, fm, secs


@pytest.mark.parametrize("amp", [1e-5, 1e-4, 30.0])
def test_fused_front_end_at_amplitude(amp, pilotcut):
    """One station at `amp` of full scale (noise 50 dB below it, as in the default signal), 0.79 s in three calls: cold
    start, lock and steady state.  1e-5: the weakest station the AGC's maximum gain of 1e5 still brings to unit amplitude,
    fp16 subnormals in the low terms; 30: near the largest amplitude the reference's AGC survives from its initial gain."""
    nblk = 120
    x = siggen.fm_stereo_iq(nblk * BLK, 10e6, amplitude=amp, sigma=amp / 300.0)
    calls = [[BLK] * 40] * 3
    st, fm, secs = _levels_case(f"fused_level_{amp:g}", x, calls, pilotcut)
    assert st.if_agc_gain == pytest.approx(fm.get_if_agc_gain(), rel=2e-4)


@pytest.mark.parametrize("amp", [1e3, 1e5])
def test_fused_front_end_if_samples_beyond_the_agc_range(amp, pilotcut):
    """Un-normalised FLOAT files (FileSource.cpp:514-528).  1e3 is inside fp16's range; 1e5 is beyond it (65504): every tile
    of both stages takes the fp32 repair path -- exact, and slow (its time is in the report).  IF samples only: at these
    amplitudes the reference's own AGC is unstable (module docstring)."""
    nblk = 80
    x = siggen.fm_stereo_iq(nblk * BLK, 10e6, amplitude=amp, sigma=amp / 300.0)
    calls = [[BLK] * 40] * 2
    _levels_case(f"fused_level_{amp:g}_if_only", x, calls, pilotcut, audio=False)


def _stepped(nblk, edges_levels):
    n = nblk * BLK
    x = siggen.fm_stereo_iq(n, 10e6, amplitude=1.0, sigma=1.0 / 300.0)
    lvl = np.empty(n, dtype=np.float32)
    edges = [e for e, _ in edges_levels] + [n]
    for (a, b), (_, v) in zip(zip(edges[:-1], edges[1:]), edges_levels):
        lvl[a:b] = v
    return (x * lvl).astype(np.complex64)


def test_fused_front_end_level_steps_mid_call(pilotcut):
    """The level of one stream steps INSIDE calls (a source that switches its gain): 1e-5 -> 5e-4 -> 2e-2 -> 1 -> 30 (every
    upward step below the 100 x the reference's AGC survives) -> 1e-4 (110 dB down: the AGC needs 13 ms to follow).
    IF samples and audio."""
    x = _stepped(240, [(0, 1e-5), (31 * BLK + 1234, 5e-4), (66 * BLK + 40001, 2e-2), (101 * BLK + 7, 1.0),
                       (138 * BLK + 55555, 30.0), (175 * BLK + 321, 1e-4)])
    _levels_case("fused_level_steps", x, [[BLK] * 60] * 4, pilotcut)


def test_fused_front_end_if_samples_across_wild_level_steps(pilotcut):
    """1e-4 -> 1e3 -> 1e-5 -> 1e5 -> 0.3 inside calls: tap windows that hold samples 140 .. 200 dB apart, tiles that switch
    between the matrix cores and the fp32 repair path.  IF samples only (module docstring), each 4096-sample piece
    relative to its own level."""
    x = _stepped(200, [(0, 1e-4), (37 * BLK + 1234, 1e3), (81 * BLK + 40001, 1e-5), (118 * BLK + 7, 1e5), (161 * BLK + 55555, 0.3)])
    _levels_case("fused_level_wild_steps_if_only", x, [[BLK] * 50] * 4, pilotcut, audio=False)
