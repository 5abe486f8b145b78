"""GPU diagnostic: Newton-round residual histories of the time-parallel AGC/PLL for a given batch size."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import bench
fmr = importlib.import_module("airspy-fmradion_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
n = B * 65536
iq = bench.synth_fm_stereo_torch(n, 10e6, 0, dev)[None]
audio = torch.zeros((1, 2 * (int(n * 0.0048) + 64)), dtype=torch.float64, device=dev)
ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=10e6, enable_resampler=True, stereo=True, max_block_len=65536, max_blocks=B)
ch.enable_kernel_timing(True)
for call in range(4):
    ch.process_blocks_device(iq.data_ptr(), n, [65536] * B, audio.data_ptr(), audio.shape[1], sync=True)
    st = ch.status()
    kt = dict(ch.kernel_times())
    print(f"call {call}: agc it={st.agc_iterations} fb={st.agc_fallback} pll it={st.pll_iterations} fb={st.pll_fallback} "
          f"locked={st.stereo_detected} agc_ms={kt.get('if_agc',0):.3f} pll_ms={kt.get('pll',0):.3f}")
    print("   agc", ["%.1e" % v for v in st.agc_residual_history[:st.agc_iterations]])
    print("   pll", ["%.1e" % v for v in st.pll_residual_history[:st.pll_iterations]])
    print("   pll components (phase,freq,u,wi1,wi2,wq1,wq2)", ["%.1e" % v for v in st.pll_residual_components[:7]])
