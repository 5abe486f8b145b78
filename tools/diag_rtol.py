"""Audio difference between PLL acceptance thresholds (GPU vs GPU, same input): python tools/diag_rtol.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import siggen
fmr = importlib.import_module("airspy-fmradion_amd")
nblk, blk, batch = 160, 65536, 40
x = siggen.fm_stereo_iq(nblk * blk, 10e6)
outs = {}
for rtol in ("0.01", "1", "10", "100"):
    os.environ["FMR_PLL_RTOL"] = rtol
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=batch)
    got, its = [], []
    for i in range(0, nblk, batch):
        a, _ = ch.process_blocks(x[None, i * blk:(i + batch) * blk], [blk] * batch)
        got.append(a[0].copy())
        st = ch.status()
        its.append((st.pll_iterations, st.pll_fallback))
    outs[rtol] = np.concatenate(got)
    ch.close()
    print("rtol", rtol, "rounds per call", its)
ref = outs["0.01"]
tail = slice(len(ref) // 2, None)
for k, v in outs.items():
    d = v - ref
    print(f"rtol {k:>5}: rms diff vs 0.01 = {np.sqrt(np.mean(d**2)):.3e} (second half {np.sqrt(np.mean(d[tail]**2)):.3e}), max {np.abs(d).max():.3e}; audio rms {np.sqrt(np.mean(ref**2)):.3f}")
