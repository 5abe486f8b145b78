import importlib, sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import bench
fmr = importlib.import_module("airspy-fmradion_amd")
dev = torch.device("cuda", 0)
B, blk = 2048, bench.BLK
n = B * blk
iq = torch.stack([bench.synth_fm_stereo_torch(n, bench.FS, 0, dev)])
audio = torch.zeros((1, 2 * (int(n * 0.0048) + 64)), dtype=torch.float64, device=dev)
for cls in (fmr.RESAMPLER_FAST, fmr.RESAMPLER_R8B):
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=bench.FS, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=B, resampler_class=cls)
    for i in range(6):
        ch.process_blocks_device(iq.data_ptr(), n, [blk] * B, audio.data_ptr(), audio.shape[1], sync=True)
        st = ch.status(0)
        print(cls, i, "iters", st.pll_iterations, "fallback", st.pll_fallback, "mismatch", [round(x, 3) for x in list(st.pll_mismatch_history)[:5]], "resid", [round(x, 4) for x in list(st.pll_residual_history)[:4]])
    ch.close()
