#!/usr/bin/env python3
"""Cycle account of one group (four samples -> error -> coefficient update) of the round-3 equaliser kernel k_mpf3 -- the
measurement k_mpf4 (chain wave + helpers) was designed from.

FMR_MPF_ACCOUNT=1 python tools/mpf_account.py [--stages 64]
Runs FM stereo + -E at the IF rate through a chain whose equaliser kernel carries s_memtime stamps at its phase
boundaries (each behind a wait for what the phase started), and the same workload through the product kernel for the
undisturbed time per group."""
import argparse, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import siggen

ap = argparse.ArgumentParser()
ap.add_argument("--stages", type=int, default=64)
args = ap.parse_args()
fmr = importlib.import_module("airspy-fmradion_amd")
fs, blk, nblk, batch = 384e3, 2517, 256, 64
x = siggen.two_ray(siggen.fm_stereo_iq(nblk * blk, fs), 20)
NAMES = ["LDS reads of the state window (10 per lane)", "complex MACs + DPP row sums", "exchange between the waves (LDS write, barrier, reads, adds)",
         "finite checks + outputs into LDS", "error, factor, coefficient update"]
res = {}
for acct in ("0", "1"):
    os.environ["FMR_MPF_ACCOUNT"] = acct
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=fs, stereo=True, multipath_stages=args.stages, max_block_len=blk, max_blocks=batch)
    for i in range(0, 128, batch):                       # the equaliser starts after 100 blocks
        ch.process_blocks(x[None, i * blk:(i + batch) * blk], [blk] * batch)
    ch.enable_kernel_timing(1)
    ms = []
    for i in range(128, nblk, batch):
        ch.process_blocks(x[None, i * blk:(i + batch) * blk], [blk] * batch)
        ms.append(dict(ch.kernel_times()).get("mpf", 0.0))
    groups = (nblk - 128) * blk / 4.0
    res[acct] = (sum(ms), groups)
    if acct == "1":
        cnt = ch.debug_read(5, cap=16)
        g = float(cnt[8])
        print("# cycle account of a group of k_mpf3<4, 5> (N = %d taps), wave 0 of stream 0, %d groups" % (4 * args.stages + 1, int(g)))
        tot = 0.0
        for i, nm in enumerate(NAMES):
            print("%-64s %8.0f cycles" % (nm, cnt[i] / g)); tot += cnt[i] / g
        print("%-64s %8.0f cycles (phases serialised by the stamps; loop control outside the stamps not counted)" % ("sum", tot))
    ch.close()
for acct, (ms, groups) in res.items():
    print("kernel time, %s: %.3f ms for %.0f groups = %.0f ns per group" % ("k_mpf3 with the stamps" if acct == "1" else "the product's kernel (k_mpf4 unless FMR_MPF3=1)", ms, groups, ms * 1e6 / groups))
