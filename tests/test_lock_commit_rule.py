"""k_pll_commit (csrc/kernels_par.hpp) decides whether a call ends in lock without walking its blocks: the lock counter of
PilotPhaseLock::process (sfmbase/PilotPhaseLock.cpp:154-167) restarts at every block below the signal threshold and counts
samples until it reaches the lock delay, so it ends at or above the delay exactly when the samples behind the last such
block -- with the carried count when there is none -- reach it.  Checked here against the per-block rule itself."""
import numpy as np


def walk(lock0, blocks, delay):
    """the reference's rule, block by block: (samples, level above the threshold) per block; empty blocks are skipped"""
    cnt = lock0
    for n, ok in blocks:
        if n == 0:
            continue
        if ok:
            if cnt < delay:
                cnt += n
        else:
            cnt = 0
    return cnt


def closed_form(lock0, blocks, delay):
    """k_pll_commit: two reductions over the blocks"""
    last_low = -1
    for i, (n, ok) in enumerate(blocks):
        if n != 0 and not ok:
            last_low = i
    ns_after = sum(n for i, (n, ok) in enumerate(blocks) if n != 0 and i > last_low)
    cnt_end = (lock0 if last_low < 0 else 0) + ns_after
    return cnt_end >= delay


def test_ends_in_lock_rule_equals_the_walk():
    rng = np.random.default_rng(20260928)
    delay = 7680          # 20 ms at 384 kHz (PilotPhaseLock.cpp:60)
    checked_true = checked_false = 0
    for case in range(4000):
        nb = int(rng.integers(1, 40))
        p_low = rng.choice([0.0, 0.02, 0.2, 0.6])
        blocks = []
        for _ in range(nb):
            n = 0 if rng.random() < 0.1 else int(rng.integers(1, 3000))
            blocks.append((n, bool(rng.random() >= p_low)))
        lock0 = int(rng.choice([0, 1, delay - 1, delay, delay + 2517, int(rng.integers(0, 2 * delay))]))
        want = walk(lock0, blocks, delay) >= delay
        assert closed_form(lock0, blocks, delay) == want, (case, lock0, blocks)
        checked_true += want
        checked_false += not want
    assert checked_true > 500 and checked_false > 500      # both outcomes are exercised


def test_counter_below_the_delay_is_exact():
    """below the delay nothing has saturated: the closed form's count IS the walk's counter (what a later call carries on)"""
    rng = np.random.default_rng(7)
    delay = 7680
    for _ in range(2000):
        blocks = [(int(rng.integers(0, 400)), bool(rng.random() > 0.1)) for _ in range(int(rng.integers(1, 30)))]
        lock0 = int(rng.integers(0, delay))
        cnt = walk(lock0, blocks, delay)
        if cnt < delay:
            last_low = max([i for i, (n, ok) in enumerate(blocks) if n and not ok], default=-1)
            ns_after = sum(n for i, (n, ok) in enumerate(blocks) if n and i > last_low)
            assert cnt == (lock0 if last_low < 0 else 0) + ns_after
