"""CPU-side checks of the product library: it loads, exports every symbol that
include/fmradion_amd.h declares, serves the filter tables, and fails loudly
without a GPU (no CPU fallback)."""
import importlib
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_filter

fmr = importlib.import_module("airspy-fmradion_amd")


@pytest.fixture(scope="module")
def lib():
    fmr.build_library()
    return fmr.lib()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "fmradion_amd.h")).read()
    declared = set(re.findall(r"\b(fmr_[a-z_]+)\s*\(", hdr))
    assert declared == set(fmr.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_filter_tables_match_fixtures(lib):
    for f in os.listdir(os.path.join(ROOT, "tests", "golden", "filters")):
        name = f[:-4]
        np.testing.assert_array_equal(fmr.filter_table(name), load_filter(name), err_msg=name)


def test_no_silent_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fmr.FmrError, match="no HIP device"):
        fmr.Chain()


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "airspy-fmradion_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in txt and "libfmoracle" not in txt and "fmradion_oracle" not in txt, f


def test_config_struct_size_is_checked(lib):
    """fmr_config grows at its end from version to version: a caller built against another header (or one that did not
    zero-initialise the struct) is refused by name, before anything else is looked at -- also without a GPU."""
    import ctypes as C
    cfg = fmr.Config()
    cfg.n_streams, cfg.max_block_len, cfg.max_blocks = 1, 1024, 1
    cfg.struct_size = C.sizeof(fmr.Config) - 8
    h = C.c_void_p()
    assert lib.fmr_create(C.byref(cfg), C.byref(h)) == fmr.ERR_BAD_ARG
    assert b"struct_size" in lib.fmr_last_error()
    assert b"0.4" in lib.fmr_version()
