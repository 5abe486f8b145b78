"""The hand-off between a live source's callback thread and the decoder: a bounded ring of page-locked
host blocks (host/fmradion_ring.hpp, fmr_host_alloc) in the role of DataBuffer (include/DataBuffer.h:35-90) and of the
per-callback vectors of AirspySource::callback / RtlSdrSource::get_samples.  The ring logic is tested on the CPU with
malloc as its allocator; on the GPU a driver thread feeds raw RTL-SDR bytes / Airspy floats and the decoded audio is
compared with the oracle fed block by block (the reference's conversions done on the host)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import oracle_py as ora
import siggen
from conftest import ROOT

fmr = importlib.import_module("airspy-fmradion_amd")
LIBDIR = os.path.join(ROOT, "airspy-fmradion_amd")


def _build(tmp_path, name):
    fmr.build_library()
    exe = os.path.join(tmp_path, name)
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-o", exe, os.path.join(ROOT, "tests", name + ".cpp"),
                    "-L", LIBDIR, "-lfmradion_amd", f"-Wl,-rpath,{LIBDIR}"], check=True)
    return exe


def test_ring_logic_on_cpu(tmp_path):
    exe = _build(str(tmp_path), "ring_check")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "RING OK" in r.stdout, r.stdout + r.stderr


def test_pinned_allocation_fails_loudly_without_gpu():
    import ctypes as C
    import torch
    L = fmr.lib()
    L.fmr_host_alloc.restype = C.c_void_p
    L.fmr_host_alloc.argtypes = [C.c_size_t]
    L.fmr_host_free.argtypes = [C.c_void_p]
    p = L.fmr_host_alloc(4096)
    if torch.cuda.is_available():
        assert p
        L.fmr_host_free(p)
    else:
        assert not p and b"no HIP device" in L.fmr_last_error()      # no pageable fallback


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["u8", "cf32"])
def test_live_source_through_the_ring(tmp_path, fmt, pilotcut):
    """0.75 s of FM stereo at 2.4 MS/s (an RTL-SDR rate) in 16384-sample callback buffers.  u8: the bytes go to the GPU
    as they are (2 B per sample over PCIe), the oracle gets (b - 128) / 128 (RtlSdrSource.cpp:359-365)."""
    fs, blk, nblk = 2.4e6, 16384, 110
    x = siggen.fm_stereo_iq(nblk * blk, fs)
    exe = _build(str(tmp_path), "ring_loop")
    fin, fout = os.path.join(tmp_path, "in.raw"), os.path.join(tmp_path, "audio.f64")
    if fmt == "u8":
        q = np.empty((len(x), 2), dtype=np.uint8)
        q[:, 0] = np.clip(np.round(x.real * 100.0) + 128, 0, 255)
        q[:, 1] = np.clip(np.round(x.imag * 100.0) + 128, 0, 255)
        q.tofile(fin)
        xo = ((q[:, 0].astype(np.int32) - 128) / np.float32(128) + 1j * ((q[:, 1].astype(np.int32) - 128) / np.float32(128))).astype(np.complex64)
    else:
        x.astype(np.complex64).tofile(fin)
        xo = x.astype(np.complex64)
    r = subprocess.run([exe, fmt, repr(fs), str(blk), "16", fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    info = dict(zip(r.stdout.split()[::2], r.stdout.split()[1::2]))
    assert int(info["blocks"]) == nblk and int(info["overruns"]) == 0 and int(info["stereo"]) == 1
    assert int(info["calls"]) < nblk and int(info["longest_run"]) > 1          # the backlog was decoded in batches
    audio = np.fromfile(fout, dtype=np.float64)
    rs = ora.IfResampler(fs, 384e3)
    fm = ora.FmDecoder(False, fmr.DELAY_3TAPS, True, 50.0, False, 0, pilotcut)
    ref = np.concatenate([fm.process(rs.process(b)) for b in siggen.blocks(xo, blk)])
    assert fm.stereo_detected()
    assert len(audio) == len(ref) > 60000
    assert float(np.sqrt(np.mean((audio - ref) ** 2))) < 1e-5
