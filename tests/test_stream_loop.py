"""The reference's stream loop (main.cpp:879-1002) built against the facade header -- variant A of INTEGRATION.md, the
classes used separately exactly as main.cpp uses them: FourthConverterIQ -> IfResampler -> Fm/Am/NbfmDecoder, one
process() per block, get_pps_events() / erase_first_pps_event() after every block.  tests/stream_loop.cpp is that
loop; here its audio and PPS output are compared with the oracle driven the same way.
Tolerance: audio RMS error < 1e-5 (north star); PPS indices exact, block_position to 1e-9."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import oracle_py as ora
import siggen
from conftest import ROOT

fmr = importlib.import_module("airspy-fmradion_amd")


def _build(tmp_path):
    fmr.build_library()
    exe = os.path.join(tmp_path, "stream_loop")
    libdir = os.path.join(ROOT, "airspy-fmradion_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "stream_loop.cpp"),
                    "-L", libdir, "-lfmradion_amd", f"-Wl,-rpath,{libdir}"], check=True)
    return exe


def test_stream_loop_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(str(tmp_path))
    x = siggen.fm_stereo_iq(4 * 2048, 384e3)
    fin = os.path.join(tmp_path, "in.cf32")
    x.tofile(fin)
    r = subprocess.run([exe, "fm", "384000", "0", "2048", fin, os.path.join(tmp_path, "a.f64"), os.path.join(tmp_path, "p.txt")],
                       capture_output=True, text=True)
    if not torch.cuda.is_available():
        assert r.returncode == 10 and "no HIP device" in r.stdout
    else:
        assert r.returncode == 0, r.stdout + r.stderr


def _run(exe, tmp_path, mode, ifrate, fourth, blk, x):
    fin, fau, fpps = (os.path.join(tmp_path, n) for n in ("in.cf32", "audio.f64", "pps.txt"))
    np.asarray(x, dtype=np.complex64).tofile(fin)
    r = subprocess.run([exe, mode, repr(float(ifrate)), str(int(fourth)), str(blk), fin, fau, fpps], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    audio = np.fromfile(fau, dtype=np.float64)
    pps = [tuple(float(v) for v in line.split()) for line in open(fpps) if line.strip()]
    return audio, pps


@pytest.mark.gpu
def test_stream_loop_fm_zero_if_with_pps(tmp_path, pilotcut):
    """FM stereo from a zero-IF source (Fs/4 shift), 1.536 MS/s, 3.4 s: lock after 0.5 s, one PPS event per second."""
    fs, blk = 1.536e6, 16384
    n = int(3.4 * fs) // blk * blk
    x = siggen.fm_stereo_iq(n, fs)
    x = (x * (1j ** (np.arange(n) % 4))).astype(np.complex64)      # the station sits at +fs/4 (main.cpp:912-919)
    exe = _build(str(tmp_path))
    audio, pps = _run(exe, str(tmp_path), "fm", fs, True, blk, x)
    f4, r = ora.FourthConverterIQ(False), ora.IfResampler(fs, 384e3, 180.0, 0.98, True)      # the facade's default class: r8brain's
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    ref, ev_ref = [], []
    for b, seg in enumerate(siggen.blocks(x, blk)):
        if_s = r.process(f4.process(seg))
        if len(if_s) == 0:
            continue
        a = fm.process(if_s)
        if len(a):
            ref.append(0.5 * a)                      # main.cpp:1000-1002
            ev_ref += [(pi, si, bp, b) for (pi, si, bp) in fm.get_pps_events()]
    ref = np.concatenate(ref)
    assert len(audio) == len(ref)
    err = float(np.sqrt(np.mean((audio - ref) ** 2)))
    assert fm.stereo_detected()
    assert err < 1e-5
    assert len(ev_ref) >= 2 and len(pps) == len(ev_ref), (pps, ev_ref)
    for g, q in zip(pps, ev_ref):
        assert int(g[0]) == q[0] and int(g[1]) == q[1] and int(g[3]) == q[3], (g, q)
        assert g[2] == pytest.approx(q[2], abs=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,fs", [("am", 384e3), ("nbfm", 384e3), ("am", 384e3 * (1 + 25e-6))])
def test_stream_loop_48k_modes(tmp_path, mode, fs, am_narrow, nbfm_default, nbfm_audio):
    """AM / NBFM: 384 kS/s IQ -> IfResampler(384 k, 48 k) -> decoder, the `-m am` / `-m nbfm` chains (main.cpp:718-723,775-777);
    the third case with `-r 25` (ifrate * (1 + 25e-6), main.cpp:708-711): the facade's IfResampler takes the fractional-phase form."""
    blk, nblk = 2048, 200
    x = siggen.am_iq(nblk * blk, fs) if mode == "am" else siggen.nbfm_iq(nblk * blk, fs)
    exe = _build(str(tmp_path))
    audio, _ = _run(exe, str(tmp_path), mode, fs, False, blk, x)
    r = ora.IfResampler(fs, 48e3, 180.0, 0.98, True)       # the facade's default resampler class (R8B)
    dec = ora.AmDecoder(am_narrow, ora.MODE_AM) if mode == "am" else ora.NbfmDecoder(nbfm_default, 8000.0, nbfm_audio)
    ref = np.concatenate([0.5 * dec.process(r.process(seg)) for seg in siggen.blocks(x, blk)])
    assert len(audio) == len(ref) > 1000
    err = float(np.sqrt(np.mean((audio - ref) ** 2)))
    assert err < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("nblk", [160, 163])
def test_stream_loop_fm_batched_facade(tmp_path, pilotcut, monkeypatch, nblk):
    """FmDecoder::set_batch_blocks(8): process() returns nothing for seven calls and the audio of eight blocks on the
    eighth -- the same samples as call-by-call decoding (main.cpp:981-984 skips the empty returns).  163 blocks: the
    last three are held back when the source ends and come out of flush() -- no block is lost."""
    fs, blk = 384e3, 2517
    x = siggen.fm_stereo_iq(nblk * blk, fs)
    exe = _build(str(tmp_path))
    monkeypatch.setenv("FMR_LOOP_BATCH", "8")
    audio, _ = _run(exe, str(tmp_path), "fm", fs, False, blk, x)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    ref = np.concatenate([0.5 * fm.process(seg) for seg in siggen.blocks(x, blk)])
    assert len(audio) == len(ref)
    assert float(np.sqrt(np.mean((audio - ref) ** 2))) < 1e-6
