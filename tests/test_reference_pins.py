"""Pins oracle constants/fixtures against the NUMERIC DATA of the reference
tree.  Runs only where /root/reference exists (the build container); skipped on
the GPU box.  Reads numbers, never code."""
import os
import re

import numpy as np
import pytest

import oracle_py as ora
from conftest import GOLD, load_filter

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _cpp_tables():
    txt = open(os.path.join(REF, "sfmbase/FilterParameters.cpp")).read()
    out = {}
    for m in re.finditer(r"FilterParameters::(\w+)\s*=\s*\{([^}]*)\}", txt):
        out[m.group(1)] = np.array([float(x) for x in re.findall(r"[-+]?\d[\d.]*(?:[eE][-+]?\d+)?", m.group(2))])
    return out


def test_filter_fixtures_equal_reference_tables():
    cpp = _cpp_tables()
    names = [f[:-4] for f in os.listdir(os.path.join(GOLD, "filters")) if f.endswith(".npy")]
    assert len(names) == 14
    for n in names:
        fx = load_filter(n)
        ref = cpp[n].astype(fx.dtype)  # IQSampleCoeff = vector<float>, SampleCoeff = vector<double>
        np.testing.assert_array_equal(fx, ref, err_msg=n)
        # symmetric, as the FIR code assumes (Filter.cpp:57 note)
        np.testing.assert_array_equal(fx, fx[::-1], err_msg=n)


def test_fast_atan_table_regeneration_matches_reference_text():
    fx = np.load(os.path.join(GOLD, "fast_atan_table.npy"))
    np.testing.assert_array_equal(ora.fast_atan_table(), fx)


def test_pll_constants_follow_documented_formulas():
    # doc/fm-pll-constants-20210116.txt:24-45
    bw = 30 / 384000.0
    p1 = np.exp(-1.146 * bw * 2 * np.pi)
    p2 = np.exp(-5.331 * bw * 2 * np.pi)
    a1, a2 = -p1 - p2, p1 * p2
    assert float("%.9g" % a1) == -1.99682419
    assert float("%.9g" % a2) == 0.996825659
    assert float("%.9g" % (1 + a1 + a2)) == pytest.approx(1.46974784e-06, rel=2e-7)
    q1 = np.exp(-0.1153 * bw * 2 * np.pi)
    b0 = 0.62 * bw * 2 * np.pi
    assert float("%.9g" % b0) == 0.000304341788
    assert float("%.9g" % (-b0 * q1)) == -0.000304324564
