#!/usr/bin/env python3
"""One 65536-sample block per fmr_process() through host buffers (the reference's own call pattern, main.cpp:916-956): the
kernels of ONE steady-state call on the chain's streams (fmr_enable_kernel_timing(3)) and the host's enqueue time.
python tools/block1_trace.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import siggen  # noqa: E402
fmr = importlib.import_module("airspy-fmradion_amd")
blk = 65536
x = siggen.fm_stereo_iq(200 * blk, 10e6)
ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=1, in_order=True)
blocks = [np.ascontiguousarray(x[i * blk:(i + 1) * blk]) for i in range(200)]
for b in blocks[:150]:
    ch.process(b)
lat = []
for b in blocks[150:190]:
    t = time.perf_counter(); ch.process(b); lat.append((time.perf_counter() - t) * 1e6)
print("untraced latency us: p50 %.1f min %.1f" % (np.percentile(lat, 50), min(lat)))
ch.enable_kernel_timing(3)
ch.process(blocks[190])
tr = ch.kernel_trace()
ch.enable_kernel_timing(0)
tr.sort(key=lambda r: r[2])
t0 = tr[0][2]
names = {0: "dec", 1: "side", 2: "agc", 4: "tail"}
for n, st, a, b in tr:
    print("%8.1f %8.1f  %-5s %s" % ((a - t0) * 1e3, (b - a) * 1e3, names.get(st, st), n))
print("traced kernels: %d, span %.1f us" % (len(tr), (max(r[3] for r in tr) - t0) * 1e3))
ch.close()
