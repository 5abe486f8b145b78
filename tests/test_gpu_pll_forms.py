"""The two forms of the PLL's Newton round against each other.

The product runs a round in THREE launches whose workgroups hand results to each other inside a launch (last-arrival
tickets, agent-scope atomic accesses instead of release / acquire fences: kernels_par.hpp, pll_last_arrival); that is
outside the HIP memory model and validated on gfx950 only.  FMR_PLL_V1=1 selects the seven-launch form of round 1, in
which every hand-off is a kernel boundary.  Both integrate the same chunks with the same arithmetic and reduce maxima
(order-independent); the node pass differs in its association only (one 7 x 8 product with a stored prefix composite
instead of a 32-step chain), i.e. at the last bits of the start states.  So block lengths, lock decisions, round counts,
fallbacks and PPS indices must be IDENTICAL and the audio equal to 1e-9 -- a stale or torn hand-off in the three-launch
form moves a chunk's start state by whole units, not by rounding.  Many calls, several streams, ragged blocks, signals that
lock, lose the pilot and relock (different round counts per call).
"""
import importlib
import os

import numpy as np
import pytest

import siggen

pytestmark = pytest.mark.gpu

fmr = importlib.import_module("airspy-fmradion_amd")


def _run(env_v1, xs, calls):
    old = os.environ.pop("FMR_PLL_V1", None)
    if env_v1:
        os.environ["FMR_PLL_V1"] = "1"
    try:
        S = xs.shape[0]
        ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=384e3, enable_resampler=False, stereo=True, n_streams=S,
                       max_block_len=8192, max_blocks=64, ab=bool(env_v1))
        audio, meta = [], []
        pos = 0
        for ll in calls:
            m = sum(ll)
            a, alen = ch.process_blocks(xs[:, pos:pos + m], ll)
            pos += m
            audio.append([np.array(a[s]) for s in range(S)])
            st = [ch.status(s) for s in range(S)]
            meta.append([(int(alen.sum()), s_.stereo_detected, s_.pll_iterations, s_.pll_fallback, s_.pilot_level,
                          tuple(ch.pps_events(i))) for i, s_ in enumerate(st)])
        ch.close()
        return audio, meta
    finally:
        os.environ.pop("FMR_PLL_V1", None)
        if old is not None:
            os.environ["FMR_PLL_V1"] = old


def test_three_launch_round_equals_seven_launch_round():
    fs, S = 384e3, 5
    rng = np.random.default_rng(11)
    calls = []
    total = 0
    while total < int(3.0 * fs):
        nb = int(rng.integers(1, 33))
        ll = [int(rng.integers(1, 8193)) if rng.random() < 0.3 else 2517 for _ in range(nb)]
        calls.append(ll)
        total += sum(ll)
    xs = []
    for s in range(S):
        x = siggen.fm_stereo_iq(total, fs, stream_id=s)
        if s == 1:      # a mono station: never locks (serial fallback in both forms)
            x = siggen.fm_stereo_iq(total, fs, stream_id=s, pilot=0.0)
        if s == 2:      # the pilot drops out for 0.3 s and returns
            a, b = int(1.2 * fs), int(1.5 * fs)
            x[a:b] = siggen.fm_stereo_iq(b - a, fs, stream_id=s, pilot=0.0, n0=a)
        xs.append(x)
    xs = np.stack(xs)
    a3, m3 = _run(False, xs, calls)
    a7, m7 = _run(True, xs, calls)
    assert len(a3) == len(a7) == len(calls) > 15
    rounds = set()
    for c in range(len(calls)):
        for s in range(S):
            p3, p7 = m3[c][s], m7[c][s]
            assert p3[:4] == p7[:4], (c, s, p3, p7)                      # audio length, lock flag, rounds, fallback
            assert p3[4] == pytest.approx(p7[4], rel=1e-6, abs=1e-12)    # pilot level (an unlocked loop amplifies the rounding difference)
            assert len(p3[5]) == len(p7[5])
            for e3, e7 in zip(p3[5], p7[5]):                             # PPS: indices exact, position in the block to 1e-9
                assert e3[0] == e7[0] and e3[1] == e7[1] and e3[3] == e7[3] and e3[2] == pytest.approx(e7[2], abs=1e-9)
            assert a3[c][s].shape == a7[c][s].shape
            if a3[c][s].size:
                assert float(np.max(np.abs(a3[c][s] - a7[c][s]))) < 1e-9, (c, s)
            rounds.add(m3[c][s][2])
    assert len(rounds) >= 2                 # calls of different round counts were compared
    assert any(m3[c][0][1] for c in range(len(calls)))      # stream 0 locked
