"""The C++ facade (reference class names and signatures) builds against the C-ABI."""
import importlib
import os
import subprocess

import pytest

from conftest import ROOT

fmr = importlib.import_module("airspy-fmradion_amd")


def _build(tmp_path):
    fmr.build_library()
    exe = os.path.join(tmp_path, "facade_smoke")
    libdir = os.path.join(ROOT, "airspy-fmradion_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "facade_smoke.cpp"),
                    "-L", libdir, "-lfmradion_amd", f"-Wl,-rpath,{libdir}"], check=True)
    return exe


def test_facade_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 10 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_facade_runs_on_gpu(tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu path" in r.stdout
