"""Error behaviour of the C-ABI on the device path (include/fmradion_amd.h): a refused call must leave the decoder state
untouched (the reference's process() cannot fail half way; a caller that sized its buffer wrongly must be able to
retry), empty blocks are legal (FmDecode.cpp:89-92), shapes the kernels cannot take are refused at create."""
import ctypes as C
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


def _raw_process(ch, iq, audio_cap):
    out = np.empty(max(audio_cap, 1), dtype=np.float64)
    n = C.c_size_t()
    rc = fmr.lib().fmr_process(ch.h, iq.ctypes.data_as(C.POINTER(C.c_float)), len(iq),
                               out.ctypes.data_as(C.POINTER(C.c_double)), audio_cap, C.byref(n))
    return rc, out[:n.value].copy()


def test_refused_calls_leave_the_state_untouched(pilotcut):
    fs, blk = 384e3, 2517
    x = siggen.fm_stereo_iq(40 * blk, fs)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=fs, stereo=True, max_block_len=blk, max_blocks=4)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    got, ref = [], []
    for i, b in enumerate(siggen.blocks(x, blk)):
        if i == 7:                                         # audio buffer too small: refused, nothing consumed
            rc, a = _raw_process(ch, b, 8)
            assert rc == fmr.ERR_CAPACITY and len(a) == 0
        if i == 9:                                         # more blocks than the chain was built for
            with pytest.raises(fmr.FmrError):
                ch.process_blocks(np.tile(b, 5)[None, :], [blk] * 5)
        if i == 11:                                        # a block longer than max_block_len
            with pytest.raises(fmr.FmrError):
                ch.process_blocks(np.tile(b, 2)[None, :], [2 * blk])
        if i == 13:                                        # n_blocks = 0
            with pytest.raises(fmr.FmrError):
                ch.process_blocks(b[None, :], [])
        got.append(ch.process(b))
        ref.append(fm.process(b))
    assert [len(g) for g in got] == [len(r) for r in ref]
    assert rms(np.concatenate(got) - np.concatenate(ref)) < 1e-6
    ch.close()


def test_empty_blocks_inside_a_call(pilotcut):
    """Zero-length blocks between real ones (a source that delivered nothing): FmDecode.cpp:89-92 returns at once."""
    fs, blk = 10e6, 65536
    x = siggen.fm_stereo_iq(24 * blk, fs)
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=fs, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=16)
    r = ora.IfResampler(fs, 384e3)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    got, ref = [], []
    for c in range(2):
        seg = x[c * 12 * blk:(c + 1) * 12 * blk]
        lens = [blk, 0, blk, blk, 0, 0] + [blk] * 9 + [0]
        a, alen = ch.process_blocks(seg[None, :], lens)
        got.append(a[0])
        o, rl = 0, []
        for n in lens:
            q = fm.process(r.process(seg[o:o + n]))
            rl.append(len(q)); ref.append(q); o += n
        assert list(alen) == rl
    assert rms(np.concatenate(got) - np.concatenate(ref)) < 1e-5
    ch.close()


def test_unsupported_shapes_are_refused_at_create():
    # raw integer input needs the v2 front-end kernel, i.e. an integer pre-decimation >= 2: refused when the
    # chain is built, not on every call
    with pytest.raises(fmr.FmrError):
        fmr.Chain(mode=fmr.MODE_FM, input_rate=1.0e6, enable_resampler=True, input_format=fmr.IQ_U8, max_block_len=2048)
    with pytest.raises(fmr.FmrError):                     # raw input without the resampler in front
        fmr.Chain(mode=fmr.MODE_FM, input_rate=384e3, enable_resampler=False, input_format=fmr.IQ_S16, max_block_len=2048)
    with pytest.raises(fmr.FmrError):                     # a ratio outside the design range
        fmr.Chain(mode=fmr.MODE_NONE, input_rate=5e9, enable_resampler=True, max_block_len=2048)
