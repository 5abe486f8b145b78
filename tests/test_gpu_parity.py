"""GPU parity tests: the HIP chain (through the C-ABI) against the CPU oracle on
the same seeded inputs.  Tolerances (north star: audio within 1e-5 RMS):
  * decoder fed identical IF samples: audio RMS error < 1e-6 (most stages are
    arithmetic-identical; libm vs ocml atan2f / sincos differ in the last ulp)
  * front end (fp32 FMA on the GPU vs fp64 accumulation in the oracle):
    IF relative RMS error < 2e-6
  * end to end from 10 MS/s IQ: audio RMS error < 1e-5 (measured ~1e-7)
Run on the GPU box:  python -m pytest tests -m gpu
"""
import importlib
import json
import os

import numpy as np
import pytest

import oracle_py as ora
import siggen
from conftest import load_filter

pytestmark = pytest.mark.gpu

fmr = importlib.import_module("airspy-fmradion_amd")
REPORT = {}


def _report(key, **kw):
    REPORT[key] = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kw.items()}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2))) if a.size else 0.0


# ------------------------------------------------------------------ front end
@pytest.mark.parametrize("fin,fout_mode,blk,nblk", [(10e6, "fm", 65536, 6), (1e6, "fm", 2048, 40), (384e3, "am", 2048, 40)])
def test_if_resampler(fin, fout_mode, blk, nblk):
    fout = 384e3 if fout_mode == "fm" else 48e3
    x = siggen.fm_stereo_iq(blk * nblk, fin) if fout_mode == "fm" else siggen.am_iq(blk * nblk, fin)
    # a front-end-only chain runs at the FM rate; for the AM ratio use an AM chain's front end via debug tap
    if fout_mode == "fm":
        ch = fmr.Chain(mode=fmr.MODE_NONE, input_rate=fin, enable_resampler=True, max_block_len=blk)
        got = [ch.resample(b) for b in siggen.blocks(x, blk)]
    else:
        ch = fmr.Chain(mode=fmr.MODE_AM, input_rate=fin, enable_resampler=True, max_block_len=blk,
                       filter_coeff=load_filter("jj1bdx_am_48khz_narrow"))
        got = []
        for b in siggen.blocks(x, blk):
            ch.process(b)
            got.append(ch.debug_read(0))
    r = ora.IfResampler(fin, fout)
    ref = [r.process(b) for b in siggen.blocks(x, blk)]
    assert [len(g) for g in got] == [len(q) for q in ref]
    g, q = np.concatenate(got), np.concatenate(ref)
    rel = rms(g - q) / rms(q)
    _report(f"if_resampler_{int(fin)}_{fout_mode}", n=len(q), rel_rms=rel, info=ch.resampler_info())
    assert rel < 2e-6
    ch.close()


def test_if_resampler_384k_matrix_core_form_against_the_vector_form(monkeypatch):
    """Round 6: 384 k -> 48 k (LB / MB = 3 / 8, TB = 214) runs as sixteen periods per 48 / 128 row block on the banded
    matrix-core kernel of the 48 / 125 shape (k_ifr_poly4<48, 128, 214>).  Ragged blocks in calls of several tiles, so
    that the call starts fall on every output position mod 48; against the oracle (2e-6) and against the kernel it
    replaced (FMR_NO_FUSED=1: k_ifr_poly3, the same taps in the same order on the vector ALUs)."""
    rng = np.random.default_rng(11)
    lens = [int(v) for v in rng.integers(1, 2049, 300)] + [2048] * 200
    n = sum(lens)
    x = siggen.am_iq(n, 384e3)
    calls, i = [], 0
    while i < len(lens):
        k = int(rng.integers(1, 101))
        calls.append(lens[i:i + k]); i += k

    def run():
        ch = fmr.Chain(mode=fmr.MODE_AM, input_rate=384e3, enable_resampler=True, max_block_len=2048, max_blocks=100,
                       filter_coeff=load_filter("jj1bdx_am_48khz_narrow"))
        out, pos = [], 0
        for ll in calls:
            m = sum(ll)
            ch.process_blocks(x[None, pos:pos + m], ll)
            out.append(ch.debug_read(0))
            pos += m
        ch.close()
        return np.concatenate(out)

    a = run()
    monkeypatch.setenv("FMR_NO_FUSED", "1")
    b = run()
    r = ora.IfResampler(384e3, 48e3)
    q = np.concatenate([r.process(blk) for blk in np.split(x, np.cumsum(lens)[:-1])])
    assert len(a) == len(b) == len(q)
    rel_o, rel_v = rms(a - q) / rms(q), rms(a - b) / rms(q)
    _report("if_resampler_384k_poly4_am", n=len(q), rel_rms_vs_oracle=rel_o, rel_rms_vs_vector_form=rel_v, max_abs_vs_vector_form=float(np.max(np.abs(a - b))))
    assert rel_o < 2e-6 and rel_v < 1e-6


def test_fourth_converter_front_end():
    blk, nblk = 16384, 4
    # zero-IF receivers deliver the station at +fs/4 (main.cpp:912-919): put it there
    x = siggen.fm_stereo_iq(blk * nblk, 1.536e6)
    x = (x * (1j ** (np.arange(len(x)) % 4))).astype(np.complex64)
    ch = fmr.Chain(mode=fmr.MODE_NONE, input_rate=1.536e6, enable_resampler=True, fourth_down=True, max_block_len=blk)
    got = np.concatenate([ch.resample(b) for b in siggen.blocks(x, blk)])
    f4, r = ora.FourthConverterIQ(False), ora.IfResampler(1.536e6, 384e3)
    ref = np.concatenate([r.process(f4.process(b)) for b in siggen.blocks(x, blk)])
    assert len(got) == len(ref)
    rel = rms(got - ref) / rms(ref)
    _report("fourth_front_end", rel_rms=rel)
    assert rel < 2e-6
    ch.close()


# ------------------------------------------------------------ FM decoder at 384 kHz
def _fm_case(name, x, blk, batch, *, fir=None, stereo=True, deemph=50.0, pilot_shift=False, stages=0,
             tol=1e-6, pilotcut=None):
    coeff = fmr.DELAY_3TAPS if fir is None else fir
    os.environ["FMR_DEBUG_TAPS"] = "1"      # keep the de-emphasised 384 kHz signal readable (debug_read 2, 3)
    try:
        ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=384e3, enable_resampler=False, fmfilter_enable=fir is not None,
                       filter_coeff=coeff, stereo=stereo, deemphasis_us=deemph, pilot_shift=pilot_shift,
                       multipath_stages=stages, max_block_len=blk, max_blocks=batch)
    finally:
        del os.environ["FMR_DEBUG_TAPS"]
    fm = ora.FmDecoder(fir is not None, coeff, stereo, deemph, pilot_shift, stages, pilotcut)
    nblk = len(x) // blk
    got, ref = [], []
    disc_err = base_err = raw_err = 0.0
    for i in range(0, nblk, batch):
        nb = min(batch, nblk - i)
        seg = x[i * blk:(i + nb) * blk]
        a, alen = ch.process_blocks(seg[None, :], [blk] * nb)
        got.append(a[0])
        rlen = []
        for b in siggen.blocks(seg, blk):
            r = fm.process(b)
            ref.append(r)
            rlen.append(len(r))
        assert list(alen) == rlen
        # stage taps of the last block of the batch
        d_ref = fm.debug_vector(0, blk)
        d_got = ch.debug_read(1)[-blk:]
        disc_err = max(disc_err, rms(d_got - d_ref))
        base_err = max(base_err, rms(ch.debug_read(3)[-blk:] - fm.debug_vector(2, blk)))
        if stereo:
            raw_err = max(raw_err, rms(ch.debug_read(2)[-blk:] - fm.debug_vector(1, blk)))
    got, ref = np.concatenate(got), np.concatenate(ref)
    assert len(got) == len(ref)
    err = rms(got - ref)
    st = ch.status()
    _report(name, n_audio=len(ref), audio_rms_err=err, audio_rms=rms(ref), disc_rms_err=disc_err,
            mono384_rms_err=base_err, stereo384_rms_err=raw_err, if_rms=st.if_rms, ref_if_rms=fm.get_if_rms(),
            pilot=st.pilot_level, ref_pilot=fm.get_pilot_level(), locked=st.stereo_detected,
            ref_locked=int(fm.stereo_detected()), agc=st.if_agc_gain, ref_agc=fm.get_if_agc_gain(),
            mpf_err=st.multipath_error, ref_mpf_err=fm.get_multipath_error(), mpf_resets=st.multipath_resets,
            agc_iters=st.agc_iterations, pll_iters=st.pll_iterations, agc_fallback=st.agc_fallback,
            pll_fallback=st.pll_fallback, pll_resid=st.pll_residual)
    assert st.agc_fallback == 0 and st.pll_fallback == 0
    assert err < tol, (name, err)
    # stage taps of the last block of every call: discriminator output (float), de-emphasised mono and L-R at 384 kHz
    # (the L-R signal carries the regenerated 38 kHz carrier: node errors of the accepted PLL trajectory show there,
    # 3e-7 .. 1e-6 normally, 5.8e-6 in pilot-shift mode where the output IS carrier x MPX -- DESIGN.md 5)
    assert disc_err < 2e-6 and base_err < 1e-6 and raw_err < 1e-5, (name, disc_err, base_err, raw_err)
    assert st.stereo_detected == int(fm.stereo_detected())
    assert st.if_rms == pytest.approx(fm.get_if_rms(), rel=1e-5)
    assert st.baseband_level == pytest.approx(fm.get_baseband_level(), rel=1e-4, abs=1e-7)
    assert st.baseband_mean * 75000.0 == pytest.approx(fm.get_tuning_offset(), rel=1e-3, abs=1e-2)
    assert st.if_agc_gain == pytest.approx(fm.get_if_agc_gain(), rel=1e-5)
    if stereo:
        # get_pilot_level is a status display value.  Its floor is the PLL's acceptance rule, not rounding: a pass is
        # accepted at a chunk-boundary mismatch of 16 units, and the unit of the pilot filter's delay states -- whose
        # magnitude IS the level -- is 1e-7 of the size of the (I, Q) delay pair, so each of the two delays of
        # an accepted trajectory may sit 1.6e-6 of that size from the serial one's: 3.2e-6 in the level at worst; audio is
        # unaffected at the 1e-9 level (tools/diag_rtol.py).  Measured over the suite: 1.5e-8 .. 2.1e-6 (the largest behind a
        # two-tap IF filter and with 2517-sample blocks, whose short calls accept on the first mismatch under the
        # threshold).  Hence 4e-6 (rounds 1-5 asserted 5e-6); 1e-6 would need the acceptance at 5 units: a third pass for
        # every call, +0.15 ms per 2^27-sample step
        # (behind the equaliser the level follows the taps, which are held to 1e-4)
        assert st.pilot_level == pytest.approx(fm.get_pilot_level(), rel=4e-6 if stages == 0 else 1e-4, abs=1e-9)
    return ch, fm, got, ref


def test_fm_stereo_decoder_384k(pilotcut):
    x = siggen.fm_stereo_iq(120 * 2048, 384e3)
    ch, fm, got, ref = _fm_case("fm_stereo_384k", x, 2048, 8, pilotcut=pilotcut)
    assert fm.stereo_detected()
    ch.close()


def test_fm_stereo_single_block_calls_equal_batched(pilotcut):
    x = siggen.fm_stereo_iq(24 * 2048, 384e3)
    ch1 = fmr.Chain(mode=fmr.MODE_FM, input_rate=384e3, stereo=True, max_block_len=2048, max_blocks=1)
    ch8 = fmr.Chain(mode=fmr.MODE_FM, input_rate=384e3, stereo=True, max_block_len=2048, max_blocks=8)
    a1 = np.concatenate([ch1.process(b) for b in siggen.blocks(x, 2048)])
    a8 = np.concatenate([ch8.process_blocks(x[None, i:i + 8 * 2048], [2048] * 8)[0][0] for i in range(0, len(x), 8 * 2048)])
    # the time-parallel PLL/AGC converge to a tolerance, so different call partitions
    # agree to that tolerance, not bit for bit
    assert rms(a1 - a8) < 1e-7
    ch1.close(); ch8.close()


def test_fm_medium_filter(pilotcut, fm_medium):
    x = siggen.fm_stereo_iq(60 * 2517, 384e3)
    ch, *_ = _fm_case("fm_medium_filter_2517", x, 2517, 6, fir=fm_medium, pilotcut=pilotcut)
    ch.close()


@pytest.mark.parametrize("ntaps", [2, 9, 32, 130, 257])
def test_fm_if_filter_of_any_length(pilotcut, ntaps):
    """LowPassFilterFirIQ (Filter.cpp:37-96) takes whatever symmetric coefficient vector the caller hands it: even and odd
    lengths, shorter than a group of four outputs, longer than the golden 127 taps (k_fm_block3's four-outputs-per-lane
    body, its remainder pairs, the middle tap, the head path at every block start and the discriminator epilogue)."""
    h = np.hamming(ntaps).astype(np.float64) * np.sinc((np.arange(ntaps) - (ntaps - 1) / 2) * 0.5)
    h = (h / h.sum()).astype(np.float32)
    assert np.array_equal(h, h[::-1])
    x = siggen.fm_stereo_iq(36 * 2517, 384e3)
    ch, *_ = _fm_case("fm_if_filter_%d_taps" % ntaps, x, 2517, 6, fir=h, pilotcut=pilotcut)
    ch.close()


def test_fm_mono_75us(pilotcut):
    x = siggen.fm_mono_iq(40 * 2048, 384e3)
    ch, *_ = _fm_case("fm_mono_75us", x, 2048, 5, stereo=False, deemph=75.0, pilotcut=pilotcut)
    ch.close()


def test_fm_pilot_shift(pilotcut):
    x = siggen.fm_stereo_iq(110 * 2048, 384e3)
    ch, *_ = _fm_case("fm_pilot_shift", x, 2048, 10, pilot_shift=True, pilotcut=pilotcut)
    ch.close()


def test_fm_ragged_and_tiny_blocks(pilotcut):
    """Ragged block lengths incl. blocks shorter than the FIR order and the halos."""
    x = siggen.fm_stereo_iq(30000, 384e3)
    lens = [1, 7, 100, 2048, 3, 126, 127, 128, 5000, 64, 2517, 10, 4096]
    lens = lens + [30000 - sum(lens)]
    fir = load_filter("jj1bdx_fm_384kHz_narrow")
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=384e3, fmfilter_enable=True, filter_coeff=fir, stereo=True,
                   max_block_len=30000, max_blocks=4)
    fm = ora.FmDecoder(True, fir, True, 50.0, False, 0, pilotcut)
    pos, got, ref = 0, [], []
    for i in range(0, len(lens), 4):
        ls = lens[i:i + 4]
        seg = x[pos:pos + sum(ls)]
        a, alen = ch.process_blocks(seg[None, :], ls)
        got.append(a[0])
        p = 0
        rl = []
        for l in ls:
            r = fm.process(seg[p:p + l]); p += l
            ref.append(r); rl.append(len(r))
        assert list(alen) == rl
        pos += sum(ls)
    got, ref = np.concatenate(got), np.concatenate(ref)
    err = rms(got - ref)
    _report("fm_ragged", audio_rms_err=err)
    assert err < 1e-6
    ch.close()


def test_fm_multipath_config4(pilotcut):
    """-E 64 on the 2-ray channel: 100 warm-up blocks then the equaliser (config 4 at the IF rate)."""
    clean = siggen.fm_stereo_iq(150 * 2517, 384e3)
    x = siggen.two_ray(clean, 20)
    ch, fm, got, ref = _fm_case("fm_multipath_E64", x, 2517, 10, stages=64, tol=1e-5, pilotcut=pilotcut)
    c_got, c_ref = ch.multipath_coefficients(), fm.get_multipath_coefficients()
    cerr = rms(c_got - c_ref)
    _report("fm_multipath_E64_coeff", coeff_rms_err=cerr, coeff_rms=rms(c_ref))
    assert cerr < 1e-4
    assert abs(ch.status().multipath_error) < 0.1
    ch.close()


@pytest.mark.parametrize("stages", [100, 300])
def test_fm_multipath_long_equalisers(pilotcut, stages):
    """-E 100 (401 taps: ten per lane of the chain wave) and -E 300 (1201 taps, the longest the chain takes: twenty per
    lane, an eight-slot snapshot ring) -- the other two shapes of k_mpf4 (MultipathFilter.cpp:39-75: order = 4 stages + 1,
    reference tap 3 stages + 1)."""
    x = siggen.two_ray(siggen.fm_stereo_iq(130 * 2517, 384e3), 20)
    ch, fm, got, ref = _fm_case("fm_multipath_E%d" % stages, x, 2517, 10, stages=stages, tol=1e-5, pilotcut=pilotcut)
    c_got, c_ref = ch.multipath_coefficients(), fm.get_multipath_coefficients()
    assert len(c_got) == len(c_ref) == 4 * stages + 1
    cerr = rms(c_got - c_ref)
    _report("fm_multipath_E%d_coeff" % stages, coeff_rms_err=cerr, coeff_rms=rms(c_ref))
    assert cerr < 1e-4
    assert ch.status().multipath_resets == 0
    ch.close()


def test_fm_multipath_nan_recovery(pilotcut):
    """A NaN burst makes the equaliser fail; taps are re-initialised and the block falls back
    to the un-equalised IF (FmDecode.cpp:116-123)."""
    x = siggen.two_ray(siggen.fm_stereo_iq(112 * 2048, 384e3), 20).copy()
    x[105 * 2048 + 77] = np.nan + 0j
    coeff = fmr.DELAY_3TAPS
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=384e3, stereo=True, multipath_stages=8, max_block_len=2048, max_blocks=4)
    fm = ora.FmDecoder(False, coeff, True, 50.0, False, 8, pilotcut)
    got, ref = [], []
    for i in range(0, 112, 4):
        seg = x[i * 2048:(i + 4) * 2048]
        got.append(ch.process_blocks(seg[None, :], [2048] * 4)[0][0])
        ref += [fm.process(b) for b in siggen.blocks(seg, 2048)]
    got, ref = np.concatenate(got), np.concatenate(ref)
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), ok)
    assert ch.status().multipath_resets >= 1
    err = rms(got[ok] - ref[ok])
    _report("fm_multipath_nan", audio_rms_err=err, resets=ch.status().multipath_resets)
    assert err < 1e-5
    ch.close()


# ------------------------------------------------------------------------- AM
@pytest.mark.parametrize("mode", ["am", "dsb"])
def test_am_decoder_48k(mode, am_narrow):
    m = fmr.MODE_AM if mode == "am" else fmr.MODE_DSB
    x = siggen.am_iq(200 * 256, 48e3)
    ch = fmr.Chain(mode=m, input_rate=48e3, filter_coeff=am_narrow, max_block_len=256, max_blocks=10)
    am = ora.AmDecoder(am_narrow, ora.MODE_AM if mode == "am" else ora.MODE_DSB)
    got, ref = [], []
    for i in range(0, 200, 10):
        seg = x[i * 256:(i + 10) * 256]
        a, alen = ch.process_blocks(seg[None, :], [256] * 10)
        got.append(a[0])
        ref += [am.process(b) for b in siggen.blocks(seg, 256)]
    got, ref = np.concatenate(got), np.concatenate(ref)
    err = rms(got - ref)
    st = ch.status()
    _report(f"{mode}_48k", audio_rms_err=err, audio_rms=rms(ref), if_agc=st.if_agc_gain, ref_if_agc=am.get_if_agc_current_gain(),
            agc_iters=st.agc_iterations, agc_fallback=st.agc_fallback, agc_hist=[float(v) for v in st.agc_residual_history[:st.agc_iterations]],
            af_agc=st.af_agc_gain, ref_af_agc=am.get_af_agc_current_gain())
    assert len(got) == len(ref) and err < 1e-6
    assert st.agc_fallback == 0 and st.af_agc_fallback == 0        # both AGCs and the audio tail ran time-parallel
    assert st.if_agc_gain == pytest.approx(am.get_if_agc_current_gain(), rel=1e-5)
    assert st.af_agc_gain == pytest.approx(am.get_af_agc_current_gain(), rel=1e-6)
    assert st.if_rms == pytest.approx(am.get_if_rms(), rel=1e-5)
    ch.close()


@pytest.mark.parametrize("mode", ["usb", "lsb", "cw", "wspr"])
def test_am_decoder_ssb_cw_modes(mode, am_narrow):
    """AmDecoder USB / LSB / CW / WSPR (AmDecode.cpp:103-147): FineTuner mixers around the 2049-tap filters."""
    fs, blk, nblk, batch = 48e3, 2048, 96, 8
    n = nblk * blk
    t = np.arange(n) / fs
    rng = np.random.default_rng(11)
    # a two-tone upper-sideband signal (+700 Hz, +1900 Hz), a weaker lower-sideband tone and noise
    x = (0.05 * np.exp(2j * np.pi * 700 * t) + 0.03 * np.exp(2j * np.pi * 1900 * t) + 0.02 * np.exp(-2j * np.pi * 1100 * t)
         + 1e-4 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    gm = {"usb": fmr.MODE_USB, "lsb": fmr.MODE_LSB, "cw": fmr.MODE_CW, "wspr": fmr.MODE_WSPR}[mode]
    om = {"usb": ora.MODE_USB, "lsb": ora.MODE_LSB, "cw": ora.MODE_CW, "wspr": ora.MODE_WSPR}[mode]
    ch = fmr.Chain(mode=gm, input_rate=fs, enable_resampler=False, filter_coeff=am_narrow, max_block_len=blk, max_blocks=batch)
    am = ora.AmDecoder(am_narrow, om, load_filter("jj1bdx_cw_48khz_500hz"), load_filter("jj1bdx_ssb_48khz_1500hz"))
    got, ref = [], []
    for i in range(0, nblk, batch):
        seg = x[i * blk:(i + batch) * blk]
        a, alen = ch.process_blocks(seg[None, :], [blk] * batch)
        got.append(a[0])
        ref += [am.process(b) for b in siggen.blocks(seg, blk)]
    got, ref = np.concatenate(got), np.concatenate(ref)
    assert len(got) == len(ref) == n
    err = rms(got - ref)
    st = ch.status()
    _report(f"am_{mode}", audio_rms_err=err, audio_rms=rms(ref), if_rms=st.if_rms, ref_if_rms=am.get_if_rms(),
            if_agc=st.if_agc_gain, ref_if_agc=am.get_if_agc_current_gain(), af_agc=st.af_agc_gain,
            ref_af_agc=am.get_af_agc_current_gain(), agc_iters=st.agc_iterations, agc_fallback=st.agc_fallback,
            af_fallback=st.af_agc_fallback, agc_hist=[float(v) for v in st.agc_residual_history[:st.agc_iterations]])
    assert rms(ref) > 1e-3
    assert err < 1e-6
    assert st.agc_fallback == 0 and st.af_agc_fallback == 0        # CW / WSPR included: no serial fallback
    assert st.if_rms == pytest.approx(am.get_if_rms(), rel=1e-5)
    assert st.if_agc_gain == pytest.approx(am.get_if_agc_current_gain(), rel=1e-4)
    assert st.af_agc_gain == pytest.approx(am.get_af_agc_current_gain(), rel=1e-6)
    ch.close()


def test_am_config3_full_chain(am_narrow):
    """Config 3: 384 kS/s IQ -> IfResampler(48 k) -> AmDecoder narrow."""
    x = siggen.am_iq(300 * 2048, 384e3)
    ch = fmr.Chain(mode=fmr.MODE_AM, input_rate=384e3, enable_resampler=True, filter_coeff=am_narrow,
                   max_block_len=2048, max_blocks=10)
    r, am = ora.IfResampler(384e3, 48e3), ora.AmDecoder(am_narrow, ora.MODE_AM)
    got, ref = [], []
    for i in range(0, 300, 10):
        seg = x[i * 2048:(i + 10) * 2048]
        got.append(ch.process_blocks(seg[None, :], [2048] * 10)[0][0])
        ref += [am.process(r.process(b)) for b in siggen.blocks(seg, 2048)]
    got, ref = np.concatenate(got), np.concatenate(ref)
    assert len(got) == len(ref)
    err = rms(got - ref)
    st = ch.status()
    _report("am_config3", audio_rms_err=err, audio_rms=rms(ref), agc_fallback=st.agc_fallback, af_fallback=st.af_agc_fallback)
    assert err < 1e-5
    assert st.agc_fallback == 0 and st.af_agc_fallback == 0
    ch.close()


# ------------------------------------------------------- fused ingest formats
@pytest.mark.parametrize("fmt,dtype,scale,off", [(1, np.int16, 32767.0, 0), (2, np.uint8, 127.0, 128), (3, np.int8, 127.0, 0)])
@pytest.mark.parametrize("fourth", [False, True])
def test_raw_input_formats(fmt, dtype, scale, off, fourth, pilotcut):
    """S16_LE / U8 / S8 IQ read by the front-end kernel itself (4 or 2 bytes per sample on HBM) against
    oracle conversion -> [FourthConverterIQ] -> IfResampler -> FmDecoder; ragged blocks exercise the tile edges."""
    lens = [65536, 65536, 4099, 1, 65536, 30001, 65536, 65536]
    n = sum(lens)
    x = siggen.fm_stereo_iq(n, 10e6)
    if fourth:
        x = (x * np.exp(2j * np.pi * 0.25 * np.arange(n))).astype(np.complex64)
    raw = np.stack([np.round(x.real / 0.3 * 0.8 * scale) + off, np.round(x.imag / 0.3 * 0.8 * scale) + off], axis=1).astype(dtype)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, fourth_down=fourth, stereo=True,
                   max_block_len=65536, max_blocks=4, input_format=fmt)
    r = ora.IfResampler(10e6, 384e3)
    f4 = ora.FourthConverterIQ(False) if fourth else None
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    xf = ora.iq_convert(fmt, raw)
    got, ref, pos = [], [], 0
    for i in range(0, len(lens), 4):
        ll = lens[i:i + 4]
        seg = raw[pos:pos + sum(ll)]
        a, alen = ch.process_blocks(seg[None, :, :], ll)
        got.append(a[0])
        o = pos
        for bl in ll:
            b = xf[o:o + bl]
            if f4 is not None:
                b = f4.process(b)
            ref.append(fm.process(r.process(b)))
            o += bl
        pos += sum(ll)
        assert list(alen) == [len(q) for q in ref[-len(ll):]]
    got, ref = np.concatenate(got), np.concatenate(ref)
    err = rms(got - ref)
    _report(f"raw_fmt{fmt}_fourth{int(fourth)}", audio_rms_err=err, audio_rms=rms(ref), n=len(ref))
    assert len(ref) > 1000 and err < 1e-5
    ch.close()


# ----------------------------------------------------------------- NbfmDecoder
@pytest.mark.parametrize("dev", [8000.0, 17000.0])
def test_nbfm_decoder_48k(dev, nbfm_default, nbfm_audio):
    """NbfmDecoder at its own rate (48 kHz), 2048-sample blocks in batches of 8 plus ragged single blocks."""
    x = siggen.nbfm_iq(200 * 2048, 48e3, dev=3000.0 if dev == 8000.0 else 9000.0)
    ch = fmr.Chain(mode=fmr.MODE_NBFM, input_rate=48e3, enable_resampler=False, filter_coeff=nbfm_default,
                   nbfm_freq_dev=dev, max_block_len=2048, max_blocks=8)
    nb = ora.NbfmDecoder(nbfm_default, dev, nbfm_audio)
    lens = [2048] * 8 * 20 + [1000, 1, 47, 2048, 999] * 8
    got, ref, pos = [], [], 0
    for i in range(0, len(lens), 8):
        ll = lens[i:i + 8]
        seg = x[pos:pos + sum(ll)]
        pos += sum(ll)
        a, alen = ch.process_blocks(seg[None, :], ll)
        got.append(a[0])
        o = 0
        for n in ll:
            ref.append(nb.process(seg[o:o + n]))
            o += n
        assert list(alen) == [len(r) for r in ref[-len(ll):]]
    got, ref = np.concatenate(got), np.concatenate(ref)
    assert len(got) == len(ref) == pos
    err = rms(got - ref)
    st = ch.status()
    _report(f"nbfm_48k_dev{int(dev)}", audio_rms_err=err, audio_rms=rms(ref), n=len(ref), agc_iters=st.agc_iterations,
            agc_fallback=st.agc_fallback, if_rms=st.if_rms, ref_if_rms=nb.get_if_rms())
    assert err < 1e-6
    assert st.agc_fallback == 0
    assert st.if_rms == pytest.approx(nb.get_if_rms(), rel=1e-5)
    assert st.baseband_level == pytest.approx(nb.get_baseband_level(), rel=1e-4, abs=1e-7)
    assert st.baseband_mean * dev == pytest.approx(nb.get_tuning_offset(), rel=1e-3, abs=1e-2)
    assert st.if_agc_gain == pytest.approx(nb.get_if_agc_current_gain(), rel=1e-4)
    ch.close()


def test_nbfm_full_chain_from_384k(nbfm_default, nbfm_audio):
    """384 kS/s IQ -> IfResampler(48 k) -> NbfmDecoder (the `-m nbfm` chain of main.cpp:718-719,828-829,959-962)."""
    x = siggen.nbfm_iq(300 * 2048, 384e3)
    ch = fmr.Chain(mode=fmr.MODE_NBFM, input_rate=384e3, enable_resampler=True, filter_coeff=nbfm_default,
                   max_block_len=2048, max_blocks=10)
    r, nb = ora.IfResampler(384e3, 48e3), ora.NbfmDecoder(nbfm_default, 8000.0, nbfm_audio)
    got, ref = [], []
    for i in range(0, 300, 10):
        seg = x[i * 2048:(i + 10) * 2048]
        got.append(ch.process_blocks(seg[None, :], [2048] * 10)[0][0])
        ref += [nb.process(r.process(b)) for b in siggen.blocks(seg, 2048)]
    got, ref = np.concatenate(got), np.concatenate(ref)
    assert len(got) == len(ref)
    err = rms(got - ref)
    _report("nbfm_from_384k", audio_rms_err=err, audio_rms=rms(ref))
    assert err < 1e-5
    ch.close()


# ------------------------------------------------------- full chains from raw IQ
def test_fm_mono_config1(pilotcut):
    """Config 1: 1 MS/s mono FM, FileSource block length 2048."""
    x = siggen.fm_mono_iq(400 * 2048, 1e6)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=1e6, enable_resampler=True, stereo=False, max_block_len=2048, max_blocks=20)
    r = ora.IfResampler(1e6, 384e3)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, False, 50.0, False, 0, pilotcut)
    got, ref = [], []
    for i in range(0, 400, 20):
        seg = x[i * 2048:(i + 20) * 2048]
        a, alen = ch.process_blocks(seg[None, :], [2048] * 20)
        got.append(a[0])
        rr = [fm.process(r.process(b)) for b in siggen.blocks(seg, 2048)]
        assert list(alen) == [len(q) for q in rr]
        ref += rr
    got, ref = np.concatenate(got), np.concatenate(ref)
    err = rms(got - ref)
    _report("fm_mono_config1", audio_rms_err=err, audio_rms=rms(ref), n=len(ref))
    assert err < 1e-5
    ch.close()


def test_fm_stereo_config2(pilotcut):
    """Config 2: 10 MS/s FM stereo, 65536-sample blocks, PLL on; 0.85 s so that the lock happens."""
    nblk, blk, batch = 130, 65536, 10
    x = siggen.fm_stereo_iq(nblk * blk, 10e6)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=batch)
    r = ora.IfResampler(10e6, 384e3)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    got, ref = [], []
    for i in range(0, nblk, batch):
        seg = x[i * blk:(i + batch) * blk]
        a, alen = ch.process_blocks(seg[None, :], [blk] * batch)
        got.append(a[0])
        rr = [fm.process(r.process(b)) for b in siggen.blocks(seg, blk)]
        assert list(alen) == [len(q) for q in rr]
        ref += rr
    got, ref = np.concatenate(got), np.concatenate(ref)
    err = rms(got - ref)
    tail = slice(len(ref) - 2 * 9600, len(ref))
    st = ch.status()
    _report("fm_stereo_config2", audio_rms_err=err, audio_rms_err_postlock=rms(got[tail] - ref[tail]),
            audio_rms=rms(ref), n=len(ref), locked=st.stereo_detected, pilot=st.pilot_level,
            agc_iters=st.agc_iterations, pll_iters=st.pll_iterations, agc_fallback=st.agc_fallback,
            pll_fallback=st.pll_fallback, pll_resid=st.pll_residual,
            agc_hist=[float(v) for v in st.agc_residual_history[:st.agc_iterations]],
            pll_hist=[float(v) for v in st.pll_residual_history[:st.pll_iterations]],
            pll_rhist=[float(v) for v in st.pll_mismatch_history[:st.pll_iterations]],
            pll_r_accepted=st.pll_mismatch_accepted)
    assert st.agc_fallback == 0 and st.pll_fallback == 0
    assert fm.stereo_detected() and st.stereo_detected == 1
    assert err < 1e-5
    ch.close()


def test_cold_first_call_is_split(pilotcut):
    """A chain's first call covering > 1.6 s of signal is cut after ~0.8 s (fmradion_amd.hip run_cold_aware): the head
    goes through the serial fallback, the rest -- starting locked -- through the time-parallel path.  Same audio as
    260 sequential process() calls."""
    nblk, blk = 260, 65536
    x = siggen.fm_stereo_iq(nblk * blk, 10e6)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=nblk)
    a, alen = ch.process_blocks(x[None, :], [blk] * nblk)
    st = ch.status()
    r = ora.IfResampler(10e6, 384e3)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    ref = [fm.process(r.process(b)) for b in siggen.blocks(x, blk)]
    assert list(alen) == [len(q) for q in ref]
    ref = np.concatenate(ref)
    err = rms(a[0] - ref)
    _report("cold_first_call", audio_rms_err=err, n=len(ref), pll_iters=st.pll_iterations, pll_fallback=st.pll_fallback,
            agc_fallback=st.agc_fallback, locked=st.stereo_detected)
    assert st.pll_fallback == 0 and st.agc_fallback == 0       # status of the second part
    assert st.stereo_detected == 1 and fm.stereo_detected()
    assert err < 1e-5
    ch.close()


@pytest.mark.parametrize("pilot,sigma,amp", [(0.04, 1e-3, 0.3), (0.10, 3e-2, 0.3), (0.10, 1e-3, 0.02)])
def test_fm_stereo_hard_signals(pilot, sigma, amp, pilotcut):
    """Weak pilot (4 %), 20 dB carrier-to-noise, and a weak carrier (AGC near 50x): the time-parallel PLL / AGC must
    either converge or fall back, and match the serial oracle either way; with these inputs no fallback is needed."""
    nblk, blk, batch = 120, 65536, 30
    x = siggen.fm_stereo_iq(nblk * blk, 10e6, amplitude=amp, sigma=sigma, pilot=pilot)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=batch)
    r = ora.IfResampler(10e6, 384e3)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    got, ref, its = [], [], []
    for i in range(0, nblk, batch):
        seg = x[i * blk:(i + batch) * blk]
        a, alen = ch.process_blocks(seg[None, :], [blk] * batch)
        got.append(a[0])
        ref += [fm.process(r.process(b)) for b in siggen.blocks(seg, blk)]
        st = ch.status()
        its.append((st.pll_iterations, st.pll_fallback, st.agc_iterations, st.agc_fallback))
    got, ref = np.concatenate(got), np.concatenate(ref)
    assert len(got) == len(ref)
    err = rms(got - ref)
    st = ch.status()
    _report(f"fm_hard_p{pilot}_s{sigma}_a{amp}", audio_rms_err=err, audio_rms=rms(ref), rounds=its, locked=st.stereo_detected,
            ref_locked=int(fm.stereo_detected()), pilot=st.pilot_level, ref_pilot=fm.get_pilot_level())
    assert st.stereo_detected == int(fm.stereo_detected())
    assert err < 1e-5
    assert all(f == 0 for _, f, _, _ in its[1:]), its     # after the first (cold) call no serial PLL fallback
    # what the headline's two-pass step costs on harder signals (bench.py --sigma prints the same for the 2^27-sample step):
    # never more than a third integration pass -- 4 % pilot: three; 17 dB C/N: two or three; a weak but clean carrier: two
    assert all(r <= 3 for r, _, _, _ in its[1:]), its
    if sigma <= 1e-3 and pilot >= 0.10:
        assert all(r == 2 for r, _, _, _ in its[1:]), its
    # the IF AGC, solved for its carried state only: one Newton round once the chain runs (the round-1 acceptance of
    # k_agc_round / agc_node_pass: the first-order correction is in, what is left is second order), no serial fallback,
    # and the state the reference carries
    assert all(a == 1 and f == 0 for _, _, a, f in its[1:]), its
    assert st.if_agc_gain == pytest.approx(fm.get_if_agc_gain(), rel=2e-4)


def test_randomised_block_partition(pilotcut):
    """Seeded random block lengths (1 .. 65536) and batch sizes over ~3.3 s of 10 MS/s FM stereo, two streams: the
    chain must equal block-by-block process() calls whatever the partition (count law, halos, tile edges)."""
    rng = np.random.default_rng(20260927)
    lens = []
    while sum(lens) < 33_000_000:
        r = rng.random()
        lens.append(int(rng.integers(1, 300)) if r < 0.15 else int(rng.integers(300, 65537)))
    n = sum(lens)
    xs = np.stack([siggen.fm_stereo_iq(n, 10e6, stream_id=s) for s in range(2)])
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, n_streams=2,
                   max_block_len=65536, max_blocks=64)
    got = [[], []]
    alens = []
    i = pos = 0
    while i < len(lens):
        k = int(rng.integers(1, 65))
        ll = lens[i:i + k]
        m = sum(ll)
        a, alen = ch.process_blocks(xs[:, pos:pos + m], ll)
        for s in range(2):
            got[s].append(a[s])
        alens += list(alen)
        i += k
        pos += m
    for s in range(2):
        r = ora.IfResampler(10e6, 384e3)
        fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
        ref, o = [], 0
        for bl in lens:
            ref.append(fm.process(r.process(xs[s, o:o + bl])))
            o += bl
        if s == 0:
            assert alens == [len(q) for q in ref]
        ref, g = np.concatenate(ref), np.concatenate(got[s])
        assert len(g) == len(ref)
        err = rms(g - ref)
        _report(f"random_partition_{s}", audio_rms_err=err, n=len(ref), blocks=len(lens))
        assert err < 1e-5
        assert ch.status(s).stereo_detected == int(fm.stereo_detected()) == 1
    ch.close()


def test_multi_stream_batch(pilotcut):
    """Three independent streams in one chain (the sharding unit of config 5)."""
    S, nblk, blk = 3, 12, 65536
    xs = np.stack([siggen.fm_stereo_iq(nblk * blk, 10e6, stream_id=s) for s in range(S)])
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, n_streams=S,
                   max_block_len=blk, max_blocks=4)
    got = [[] for _ in range(S)]
    for i in range(0, nblk, 4):
        a, alen = ch.process_blocks(xs[:, i * blk:(i + 4) * blk], [blk] * 4)
        for s in range(S):
            got[s].append(a[s])
    for s in range(S):
        r = ora.IfResampler(10e6, 384e3)
        fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
        ref = np.concatenate([fm.process(r.process(b)) for b in siggen.blocks(xs[s], blk)])
        g = np.concatenate(got[s])
        assert len(g) == len(ref)
        err = rms(g - ref)
        _report(f"multi_stream_{s}", audio_rms_err=err)
        assert err < 1e-5
        assert ch.status(s).if_agc_gain == pytest.approx(fm.get_if_agc_gain(), rel=1e-4)
    ch.close()


def test_pipelined_front_end_equals_plain(pilotcut, monkeypatch):
    """FMR_PIPELINE=1 (front end of call N+1 on its own stream and IF buffer, beside the decoder of call N)
    changes scheduling only: audio is bit-identical to the plain chain over many calls."""
    nblk, blk, batch = 24, 65536, 2
    x = siggen.fm_stereo_iq(nblk * blk, 10e6)
    outs = []
    monkeypatch.setenv("FMR_NO_FUSED", "1")     # the pipelined chain keeps the three-kernel front end: compare like with like
    for flag in ("0", "1"):
        monkeypatch.setenv("FMR_PIPELINE", flag)
        ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, max_block_len=blk,
                       max_blocks=batch)
        got = []
        for i in range(0, nblk, batch):
            a, alen = ch.process_blocks(x[None, i * blk:(i + batch) * blk], [blk] * batch)
            got.append(a[0].copy())
        outs.append(np.concatenate(got))
        ch.close()
    assert len(outs[0]) == len(outs[1]) and len(outs[0]) > 0
    assert np.array_equal(outs[0], outs[1])


def test_library_is_the_hip_path():
    """The product never routes through the oracle: its shared object holds gfx950 code objects."""
    data = open(fmr.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    assert b"ora_fm_process" not in data


def test_probe_read_bandwidth_is_plausible():
    """fmr_probe_read_bandwidth (the measurement aid behind bench.py's roofline.box_streaming_read): a plain read-only
    kernel over 1 GiB of device memory (four times the last-level cache) reports a rate between a tenth of and the whole data-sheet peak, and refuses
    a null buffer."""
    import torch
    buf = torch.zeros(1 << 28, dtype=torch.float32, device="cuda:0")
    gbs = fmr.probe_read_bandwidth(0, buf.data_ptr(), buf.numel() * 4, reps=3)
    assert 800.0 < gbs < 8000.0, gbs
    with pytest.raises(fmr.FmrError):
        fmr.probe_read_bandwidth(0, 0, 1 << 28)
