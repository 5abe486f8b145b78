"""PCIe-inclusive rate of the drop-in single-block API (host buffers in, host buffers out):
fmr_process on 65536-sample blocks, one call per block, as main.cpp's loop would call it."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import siggen
fmr = importlib.import_module("airspy-fmradion_amd")
blk, nblk = 65536, 200
x = siggen.fm_stereo_iq(64 * blk, 10e6)
ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=1, in_order=True)
for i in range(100):
    ch.process(x[(i % 64) * blk:(i % 64 + 1) * blk])
t0 = time.perf_counter()
for i in range(nblk):
    ch.process(x[(i % 64) * blk:(i % 64 + 1) * blk])
dt = time.perf_counter() - t0
st = ch.status()
print(f"single-block host API: {nblk * blk / dt / 1e6:.1f} MS/s ({dt / nblk * 1e6:.1f} us per 65536-sample block), "
      f"pll rounds {st.pll_iterations} fallback {st.pll_fallback}, agc rounds {st.agc_iterations}")
