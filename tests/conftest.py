import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    """GPU runs: bring torch's HIP context up before any test loads the A/B partner build of the library
    (libfmradion_amd_ab.so, the same kernels a second time): torch.cuda initialised AFTER both libraries are loaded has been
    seen to report "No HIP GPUs are available" on the GPU box (the tests that allocate through torch then fail for a
    reason that has nothing to do with them)."""
    if not any("gpu" in it.keywords for it in items):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


GOLD = os.path.join(HERE, "golden")


def load_filter(name):
    return np.load(os.path.join(GOLD, "filters", name + ".npy"))


@pytest.fixture(scope="session")
def pilotcut():
    return load_filter("jj1bdx_48khz_fmaudio")


@pytest.fixture(scope="session")
def fm_medium():
    return load_filter("jj1bdx_fm_384kHz_medium")


@pytest.fixture(scope="session")
def am_narrow():
    return load_filter("jj1bdx_am_48khz_narrow")


@pytest.fixture(scope="session")
def nbfm_default():
    return load_filter("jj1bdx_nbfm_48khz_default")


@pytest.fixture(scope="session")
def nbfm_audio():
    return load_filter("jj1bdx_48khz_nbfmaudio")

