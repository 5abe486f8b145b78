"""The pipelined chain (the default for FM with the resampler: front end and PLL stage alternating on the decoder stream, the
audio tail of the call before beside them on its own stream, ring slots between the stages) against the in-order chain (FMR_PIPELINE=0) and
against the oracle.  Pipelining changes scheduling and buffer placement only: the audio must be BIT-IDENTICAL.

The calls are enqueued without a synchronisation in between (fmr_process_blocks_device, sync = 0) -- that is the only
way the stages of different calls really overlap -- from input and into output buffers that stay valid until the
final fmr_synchronize (IfResampler.cpp:37-78 is independent of the decoder's state; FmDecode.cpp:85-221 is
block-sequential; main.cpp:916-956 the loop).
"""
import importlib

import numpy as np
import pytest

import oracle_py as ora
import siggen
from conftest import load_filter

pytestmark = pytest.mark.gpu

fmr = importlib.import_module("airspy-fmradion_amd")
BLK = 65536


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2))) if a.size else 0.0


def _run_async(x, calls, monkeypatch, env, **chain_kw):
    """x: (S, n) complex64; calls: list of block-length lists.  Every call is enqueued asynchronously; one synchronise
    at the end.  Returns (audio per stream concatenated over the calls, audio_len per call, status of stream 0)."""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    S = x.shape[0]
    max_blocks = max(len(c) for c in calls)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, n_streams=S, max_block_len=BLK,
                   max_blocks=max_blocks, **chain_kw)
    d_x = torch.from_numpy(np.ascontiguousarray(x).view(np.float32).reshape(S, -1)).cuda()      # (S, 2 n) float32
    n = x.shape[1]
    cap = int(max(sum(c) for c in calls) * 0.0048 * 2) + 256
    d_a = torch.zeros((len(calls), S, cap), dtype=torch.float64, device="cuda")
    alens, off = [], 0
    for i, c in enumerate(calls):
        # stream s of this call starts at d_x[s, 2 off]: stride between streams = n samples
        alens.append(ch.process_blocks_device(d_x.data_ptr() + 8 * off, n, c, d_a[i].data_ptr(), cap, sync=False))
        off += sum(c)
    ch.synchronize()
    a = d_a.cpu().numpy()
    out = [np.concatenate([a[i, s, :int(alens[i].sum())] for i in range(len(calls))]) for s in range(S)]
    st = ch.status(0)
    ch.close()
    for k in env:
        monkeypatch.delenv(k, raising=False)
    return out, alens, st


def _oracle(x, calls, pilotcut, stereo=True, fir=None):
    r = ora.IfResampler(10e6, 384e3)
    fm = ora.FmDecoder(fir is not None, fmr.DELAY_3TAPS if fir is None else fir, stereo, 50.0, False, 0, pilotcut)
    ref, o = [], 0
    for c in calls:
        for bl in c:
            ref.append(fm.process(r.process(x[o:o + bl])))
            o += bl
    return np.concatenate(ref), fm


# a long first call (cold start and lock), full-size calls, a call of tiny blocks (three-kernel front end, discriminator in
# the decoder stage), a one-block call, ragged blocks
CALLS = [[BLK] * 90, [BLK] * 12, [BLK] * 12, [1500, 2000, 1200], [BLK] * 12, [BLK], [BLK] * 7 + [30000, 5000, 65000], [BLK] * 12, [BLK] * 12]


def test_pipelined_chain_is_bit_identical_to_the_in_order_chain(pilotcut, monkeypatch):
    n = sum(sum(c) for c in CALLS)
    x = siggen.fm_stereo_iq(n, 10e6)[None, :]
    plain, al0, st0 = _run_async(x, CALLS, monkeypatch, {"FMR_PIPELINE": "0"})
    piped, al1, st1 = _run_async(x, CALLS, monkeypatch, {"FMR_PIPELINE": "1"})
    assert [list(a) for a in al0] == [list(a) for a in al1]
    assert len(plain[0]) == len(piped[0]) > 0
    assert np.array_equal(plain[0], piped[0])
    assert st0.stereo_detected == st1.stereo_detected == 1
    assert st0.pll_fallback == st1.pll_fallback
    ref, fm = _oracle(x[0], CALLS, pilotcut)
    assert len(ref) == len(piped[0])
    assert rms(piped[0] - ref) < 1e-5
    assert st1.pilot_level == pytest.approx(fm.get_pilot_level(), rel=4e-6)      # (the acceptance rule's floor: test_gpu_parity._fm_case)


@pytest.mark.parametrize("env", [{"FMR_FE_CUS": "200"}, {"FMR_FE_CUS": "256"}, {"FMR_FE_CUS": "61"}])
def test_front_end_workgroup_count_does_not_change_the_audio(pilotcut, monkeypatch, env):
    """How many CUs the persistent front-end kernel takes (all but one per XCD by default) changes how the macro tiles are
    cut over the workgroups, nothing else: a sample's lane and arithmetic are functions of its absolute index."""
    calls = [[BLK] * 90] + [[BLK] * 10] * 6
    n = sum(sum(c) for c in calls)
    x = siggen.fm_stereo_iq(n, 10e6)[None, :]
    base, _, _ = _run_async(x, calls, monkeypatch, {"FMR_PIPELINE": "0"})
    got, _, st = _run_async(x, calls, monkeypatch, dict(env, FMR_PIPELINE="1"))
    assert np.array_equal(base[0], got[0])
    assert st.stereo_detected == 1


def test_pipelined_two_streams_mono_and_if_filter(pilotcut, monkeypatch):
    """The other shapes of the stage hand-off: two streams in one chain; a mono decoder (no PLL stage: the tail starts
    from the discriminator); the IF filter on (the fused kernel stores IF samples, the FIR history is a ring halo)."""
    calls = [[BLK] * 90] + [[BLK] * 9, [BLK] * 11, [BLK] * 10]
    n = sum(sum(c) for c in calls)
    x2 = np.stack([siggen.fm_stereo_iq(n, 10e6, stream_id=s) for s in range(2)])
    a0, _, _ = _run_async(x2, calls, monkeypatch, {"FMR_PIPELINE": "0"})
    a1, _, _ = _run_async(x2, calls, monkeypatch, {"FMR_PIPELINE": "1"})
    for s in range(2):
        assert np.array_equal(a0[s], a1[s])
        ref, _ = _oracle(x2[s], calls, pilotcut)
        assert rms(a1[s] - ref) < 1e-5
    x = x2[:1]
    m0, _, _ = _run_async(x, calls, monkeypatch, {"FMR_PIPELINE": "0"}, stereo=False)
    m1, _, _ = _run_async(x, calls, monkeypatch, {"FMR_PIPELINE": "1"}, stereo=False)
    assert np.array_equal(m0[0], m1[0])
    ref, _ = _oracle(x[0], calls, pilotcut, stereo=False)
    assert rms(m1[0] - ref) < 1e-5
    fir = load_filter("jj1bdx_fm_384kHz_medium")
    f0, _, _ = _run_async(x, calls, monkeypatch, {"FMR_PIPELINE": "0"}, fmfilter_enable=True, filter_coeff=fir)
    f1, _, _ = _run_async(x, calls, monkeypatch, {"FMR_PIPELINE": "1"}, fmfilter_enable=True, filter_coeff=fir)
    assert np.array_equal(f0[0], f1[0])
    ref, _ = _oracle(x[0], calls, pilotcut, fir=fir)
    assert rms(f1[0] - ref) < 1e-5


def test_fused_runs_of_equal_weight_and_call_edges(pilotcut, monkeypatch):
    """Round 6: (i) a call of more than ~450 blocks -- only its last ~400 blocks carry the per-block partial sums, and the
    front end's runs are cut by WEIGHT (a macro tile with sums counts 1.10), so the workgroups' runs differ in length; (ii)
    the epochs at the two ends of a call go through the loader's DMA path with a per-lane source (samples / in_halo / a zero
    block) when the call's length is even, and through element loads when it is odd.  Neither may change a sample: against
    the three-kernel front end (no runs, no epochs) to rounding, against the oracle to the tolerance, and the statistics the
    partial sums feed to theirs."""
    calls = [[BLK] * 470, [BLK] * 3 + [4099], [BLK] * 4, [BLK] * 2 + [30001, 1], [BLK] * 8]      # even | odd length | (odd start) | odd | ...
    n = sum(sum(c) for c in calls)
    x = siggen.fm_stereo_iq(n, 10e6)[None, :]
    fused, al1, st1 = _run_async(x, calls, monkeypatch, {})
    three, al0, st0 = _run_async(x, calls, monkeypatch, {"FMR_NO_FUSED": "1"})
    assert [list(a) for a in al0] == [list(a) for a in al1]
    assert rms(fused[0] - three[0]) < 1e-6
    ref, fm = _oracle(x[0], calls, pilotcut)
    assert len(ref) == len(fused[0])
    assert rms(fused[0] - ref) < 1e-5
    assert st1.stereo_detected == 1 and st1.pll_fallback == 0
    assert st1.if_rms == pytest.approx(fm.get_if_rms(), rel=1e-5)
    assert st1.baseband_level == pytest.approx(fm.get_baseband_level(), rel=1e-4, abs=1e-7)


def test_probe_shader_clock_is_plausible():
    """fmr_probe_shader_clock (bench.py's `clock` object): between 0.5 and 3 GHz, and an output pointer is required."""
    mhz = fmr.probe_shader_clock(0)
    assert 500.0 < mhz < 3000.0, mhz
