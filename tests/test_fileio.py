"""File containers on both sides of the hot path (host/fmradion_fileio.hpp): IQ readers for
WAV (PCM u8 / 16 / 24, IEEE float) and RAW (U8_LE, S8_LE, S16_LE, S24_LE, FLOAT) with the conversions sf_read_float
applies (sfmbase/FileSource.cpp:120-128,491-531), audio writers RAW / WAV in int16 / float32 with the -6 dB of
main.cpp:1000-1002 (sfmbase/AudioOutput.cpp:34-167), PPS text records (main.cpp:1084-1111).  CPU only."""
import os
import struct
import subprocess
import wave

import numpy as np
import pytest
from scipy.io import wavfile

from conftest import ROOT


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("fileio")
    out = os.path.join(d, "fileio_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", out, os.path.join(ROOT, "tests", "fileio_check.cpp")], check=True)
    return out


def _read(exe, tmp_path, path, raw, fmt, blk=1000):
    out = os.path.join(tmp_path, "dump.cf32")
    r = subprocess.run([exe, "read", path, str(int(raw)), fmt, str(blk), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rate, blocks, samples = (int(v) for v in r.stdout.split()[1::2])
    return np.fromfile(out, dtype=np.complex64), rate, blocks, samples


RNG = np.random.default_rng(7)


@pytest.mark.parametrize("fmt,dtype,scale,off", [("U8_LE", np.uint8, 128.0, 128), ("S8_LE", np.int8, 128.0, 0),
                                                   ("S16_LE", np.int16, 32768.0, 0), ("FLOAT", np.float32, 1.0, 0)])
def test_raw_formats(exe, tmp_path, fmt, dtype, scale, off):
    n = 4321
    if dtype == np.float32:
        raw = RNG.standard_normal((n, 2)).astype(np.float32)
    else:
        info = np.iinfo(dtype)
        raw = RNG.integers(info.min, info.max + 1, size=(n, 2)).astype(dtype)
    path = os.path.join(tmp_path, "iq.raw")
    raw.tofile(path)
    got, rate, blocks, samples = _read(exe, str(tmp_path), path, True, fmt)
    want = ((raw.astype(np.float64) - off) / scale).astype(np.float32)
    assert samples == n and blocks == 5 and rate == 384000
    assert np.array_equal(got.view(np.float32).reshape(-1, 2), want)


def test_raw_s24(exe, tmp_path):
    n = 1000
    v = RNG.integers(-(1 << 23), 1 << 23, size=(n, 2)).astype(np.int32)
    b = np.zeros((n, 2, 3), dtype=np.uint8)
    for k in range(3):
        b[:, :, k] = (v >> (8 * k)) & 0xFF
    path = os.path.join(tmp_path, "iq.s24")
    b.tofile(path)
    got, *_ = _read(exe, str(tmp_path), path, True, "S24_LE", blk=333)
    assert np.array_equal(got.view(np.float32).reshape(-1, 2), (v / 8388608.0).astype(np.float32))


def test_wav_pcm16_and_float_and_u8(exe, tmp_path):
    n, rate = 5000, 1_000_000
    i16 = RNG.integers(-32768, 32768, size=(n, 2)).astype(np.int16)
    p16 = os.path.join(tmp_path, "a.wav")
    wavfile.write(p16, rate, i16)
    got, r, _, s = _read(exe, str(tmp_path), p16, False, "FLOAT")
    assert r == rate and s == n and np.array_equal(got.view(np.float32).reshape(-1, 2), (i16 / 32768.0).astype(np.float32))
    f32 = RNG.standard_normal((n, 2)).astype(np.float32)
    pf = os.path.join(tmp_path, "b.wav")
    wavfile.write(pf, rate, f32)
    got, r, _, s = _read(exe, str(tmp_path), pf, False, "S16_LE")        # the header decides, not the argument
    assert r == rate and s == n and np.array_equal(got.view(np.float32).reshape(-1, 2), f32)
    u8 = RNG.integers(0, 256, size=(n, 2)).astype(np.uint8)
    pu = os.path.join(tmp_path, "c.wav")
    wavfile.write(pu, rate, u8)
    got, *_ = _read(exe, str(tmp_path), pu, False, "FLOAT")
    assert np.array_equal(got.view(np.float32).reshape(-1, 2), ((u8.astype(np.float64) - 128) / 128).astype(np.float32))


def test_wav_pcm24_with_extra_chunks(exe, tmp_path):
    """PCM 24 written by the standard library's wave module, with a LIST chunk squeezed in front of the data."""
    n, rate = 777, 2_500_000
    v = RNG.integers(-(1 << 23), 1 << 23, size=(n, 2)).astype(np.int32)
    b = np.zeros((n, 2, 3), dtype=np.uint8)
    for k in range(3):
        b[:, :, k] = (v >> (8 * k)) & 0xFF
    p = os.path.join(tmp_path, "d.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(3); w.setframerate(rate)
        w.writeframes(b.tobytes())
    data = open(p, "rb").read()
    k = data.index(b"data")
    extra = b"LIST" + struct.pack("<I", 5) + b"hello" + b"\0"             # odd size + pad byte
    patched = data[:k] + extra + data[k:]
    patched = patched[:4] + struct.pack("<I", len(patched) - 8) + patched[8:]
    open(p, "wb").write(patched)
    got, r, _, s = _read(exe, str(tmp_path), p, False, "FLOAT")
    assert r == rate and s == n
    assert np.array_equal(got.view(np.float32).reshape(-1, 2), (v / 8388608.0).astype(np.float32))


@pytest.mark.parametrize("fmt", ["RAW_INT16", "RAW_FLOAT32", "WAV_INT16", "WAV_FLOAT32"])
def test_audio_writers(exe, tmp_path, fmt):
    audio = (0.9 * np.sin(np.arange(4801) * 0.01)).astype(np.float64)       # odd count: mono, pad byte for 16-bit WAV
    src = os.path.join(tmp_path, "a.f64")
    audio.tofile(src)
    dst = os.path.join(tmp_path, "out.bin")
    r = subprocess.run([exe, "write", fmt, src, dst, "48000", "0", "0.5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    half = 0.5 * audio                                                       # main.cpp:1000-1002
    if fmt == "RAW_INT16":
        got = np.fromfile(dst, dtype=np.int16)
        assert np.array_equal(got, np.rint(half * 32767.0).astype(np.int16))
    elif fmt == "RAW_FLOAT32":
        assert np.array_equal(np.fromfile(dst, dtype=np.float32), half.astype(np.float32))
    else:
        rate, got = wavfile.read(dst)
        assert rate == 48000 and got.ndim == 1 and len(got) == len(audio)
        if fmt == "WAV_INT16":
            assert got.dtype == np.int16 and np.array_equal(got, np.rint(half * 32767.0).astype(np.int16))
        else:
            assert got.dtype == np.float32 and np.array_equal(got, half.astype(np.float32))
    lines = r.stdout.splitlines()
    assert lines[0] == "       3      123456789  1700000000.250000   -12.346"     # "{:>8} {:>14} {:18.6f} {:+9.3f}"
    assert lines[1] == "         42  1700000000.500000    +3.200"                # "{:11} {:18.6f} {:+9.3f}"


def test_stereo_wav_roundtrip_through_reader(exe, tmp_path):
    """A float32 stereo WAV written by AudioFileWriter is a valid 2-channel file for IqFileReader."""
    audio = RNG.standard_normal(2 * 1234)
    src = os.path.join(tmp_path, "s.f64")
    audio.tofile(src)
    dst = os.path.join(tmp_path, "s.wav")
    subprocess.run([exe, "write", "WAV_FLOAT32", src, dst, "48000", "1", "1.0"], check=True, capture_output=True)
    got, rate, _, s = _read(exe, str(tmp_path), dst, False, "FLOAT")
    assert rate == 48000 and s == 1234
    assert np.array_equal(got.view(np.float32), audio.astype(np.float32))


def test_wav_with_unfinalised_data_size(exe, tmp_path):
    """Streaming recorders leave 0 (or 0xFFFFFFFF) in the data chunk's size: the data then run to the end of the file."""
    n = 3000
    iq = RNG.integers(-32768, 32768, size=(n, 2)).astype(np.int16)
    for size_field in (0, 0xFFFFFFFF):
        path = os.path.join(tmp_path, f"stream_{size_field}.wav")
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 0xFFFFFFFF if size_field else 0) + b"WAVE")
            f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 912000, 912000 * 4, 4, 16))
            f.write(b"data" + struct.pack("<I", size_field))
            f.write(iq.tobytes())
        got, rate, blocks, samples = _read(exe, str(tmp_path), path, False, "FLOAT")
        assert rate == 912000 and samples == n
        assert np.array_equal(got.view(np.float32).reshape(-1, 2), (iq / 32768.0).astype(np.float32))


def test_wav_with_empty_data_chunk_and_metadata_behind_it(exe, tmp_path):
    """A size of 0 means "runs to the end of the file" only if the data chunk is the last one: an empty recording with a
    LIST chunk behind it holds no samples (the chunk's bytes must not be decoded as IQ), and a data size larger than what
    the file holds is clamped to what is there."""
    path = os.path.join(tmp_path, "empty_list.wav")
    info = b"INFOISFT" + struct.pack("<I", 8) + b"recorder"
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 0) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 912000, 912000 * 4, 4, 16))
        f.write(b"data" + struct.pack("<I", 0))
        f.write(b"LIST" + struct.pack("<I", len(info)) + info)
    got, rate, blocks, samples = _read(exe, str(tmp_path), path, False, "FLOAT")
    assert rate == 912000 and samples == 0 and len(got) == 0
    n = 500
    iq = RNG.integers(-32768, 32768, size=(n, 2)).astype(np.int16)
    path = os.path.join(tmp_path, "truncated.wav")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + 4000 * 4) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 912000, 912000 * 4, 4, 16))
        f.write(b"data" + struct.pack("<I", 4000 * 4))         # announces 4000 samples, holds 500
        f.write(iq.tobytes())
    got, rate, blocks, samples = _read(exe, str(tmp_path), path, False, "FLOAT")
    assert samples == n
    assert np.array_equal(got.view(np.float32).reshape(-1, 2), (iq / 32768.0).astype(np.float32))


def test_wav_header_chunk_sizes_are_bounded(exe, tmp_path):
    """A header that announces a gigabyte-sized fmt chunk is refused, not allocated."""
    path = os.path.join(tmp_path, "evil.wav")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 100) + b"WAVE" + b"fmt " + struct.pack("<I", 0xF0000000) + b"\0" * 64)
    r = subprocess.run([exe, "read", path, "0", "FLOAT", "1000", os.path.join(tmp_path, "o.cf32")], capture_output=True, text=True)
    assert r.returncode == 3 and "bad fmt chunk" in r.stdout


def test_wav_writer_file_is_valid_before_close(exe, tmp_path):
    """The header is written at open and refreshed while the data grow (SFC_SET_UPDATE_HEADER_AUTO, AudioOutput.cpp:91-93):
    a copy of the file taken BEFORE close() is a readable WAV holding all but (at most) the last second."""
    n = 5 * 48000 * 2 + 777 * 2
    x = RNG.uniform(-0.9, 0.9, n)
    fin, fout, snap = (os.path.join(tmp_path, k) for k in ("a.f64", "a.wav", "snapshot.wav"))
    x.tofile(fin)
    r = subprocess.run([exe, "write", "WAV_INT16", fin, fout, "48000", "1", "1.0", snap], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    rate, full = wavfile.read(fout)
    assert rate == 48000 and full.shape == (n // 2, 2)
    rate_s, part = wavfile.read(snap)              # scipy refuses a zero / garbage header
    assert rate_s == 48000 and part.shape[1] == 2
    assert n // 2 - 48000 - 1 <= part.shape[0] <= n // 2
    assert np.array_equal(part, full[:part.shape[0]])
