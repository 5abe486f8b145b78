"""Residual histories of the first calls of the bench workload (acquisition): python tools/diag_acquisition.py [blocks]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
fmr = importlib.import_module("airspy-fmradion_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
n = B * bench.BLK
iq = torch.stack([bench.synth_fm_stereo_torch(n, bench.FS, 0, dev)])
audio = torch.zeros((1, 2 * (int(n * 0.0048) + 64)), dtype=torch.float64, device=dev)
ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=bench.FS, enable_resampler=True, stereo=True, max_block_len=bench.BLK, max_blocks=B)
for call in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ch.process_blocks_device(iq.data_ptr(), n, [bench.BLK] * B, audio.data_ptr(), audio.shape[1], sync=True)
    dt = time.perf_counter() - t0
    st = ch.status(0)
    print(f"call {call}: {dt*1e3:9.2f} ms  agc it {st.agc_iterations} fb {st.agc_fallback} hist {[float('%.2g'%v) for v in st.agc_residual_history[:st.agc_iterations]]}"
          f"  pll it {st.pll_iterations} fb {st.pll_fallback} d {[float('%.3g'%v) for v in st.pll_residual_history[:st.pll_iterations]]}"
          f" r {[float('%.3g'%v) for v in st.pll_mismatch_history[:st.pll_iterations]]} locked {st.stereo_detected}")
