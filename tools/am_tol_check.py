#!/usr/bin/env python3
"""AM (config 3): audio error of LONG calls against the oracle, call by call, with the IF AGC's rounds -- what an acceptance
threshold of the Newton rounds costs in the audio (FMR_X_AMTOL in a -DFMR_DIAG_KNOBS build).  python tools/am_tol_check.py [blocks per call] [calls]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import siggen                      # noqa: E402
import oracle_py as ora           # noqa: E402
fmr = importlib.import_module("airspy-fmradion_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
narrow = np.load(os.path.join(ROOT, "tests", "golden", "filters", "jj1bdx_am_48khz_narrow.npy"))
x = siggen.am_iq(K * B * 2048, 384e3)
ch = fmr.Chain(mode=fmr.MODE_AM, input_rate=384e3, enable_resampler=True, filter_coeff=narrow, max_block_len=2048, max_blocks=B)
r, am = ora.IfResampler(384e3, 48e3), ora.AmDecoder(narrow, ora.MODE_AM)
for c in range(K):
    seg = x[c * B * 2048:(c + 1) * B * 2048]
    got = ch.process_blocks(seg[None, :], [2048] * B)[0][0]
    ref = np.concatenate([am.process(r.process(b)) for b in siggen.blocks(seg, 2048)])
    st = ch.status()
    print("call %d: audio rms err %.3e (audio rms %.3f)  agc rounds %d residuals %s fallback %d" %
          (c, float(np.sqrt(np.mean((got - ref) ** 2))), float(np.sqrt(np.mean(ref ** 2))), st.agc_iterations,
           ["%.2e" % v for v in st.agc_residual_history[:st.agc_iterations]], st.agc_fallback))
ch.close()
