#!/usr/bin/env python3
"""ms per step of the default workload without bench.py's checks (diagnostic builds whose audio is not the product's:
-DFMR_DIAG_NO_TAIL ...).  python tools/step_time.py [--steps 100] [--lib tools/tmp_x.so]"""
import argparse, importlib, os, shutil, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--blocks", type=int, default=2048)
args = ap.parse_args()
import bench  # noqa: E402
import torch  # noqa: E402
fmr = importlib.import_module("airspy-fmradion_amd")
dev = torch.device("cuda", 0)
B, blk = args.blocks, bench.BLK
n = B * blk
iq = torch.stack([bench.synth_fm_stereo_torch(n, bench.FS, 0, dev)])
audio = torch.zeros((1, 2 * (int(n * 0.0048) + 64)), dtype=torch.float64, device=dev)
ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=bench.FS, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=B)
bl = [blk] * B
def step():
    ch.process_blocks_device(iq.data_ptr(), n, bl, audio.data_ptr(), audio.shape[1], sync=False)
step(); ch.synchronize()
t = time.perf_counter()
while time.perf_counter() - t < 0.08:
    step()
ch.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
ch.synchronize(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps * 1e3
ch.enable_kernel_timing(1)
step(); ch.synchronize()
kt = {k: round(v, 4) for k, v in ch.kernel_times()}
st = ch.status(0)
print("ms_per_step %.4f  pll rounds %d  kernels %s" % (dt, st.pll_iterations, kt))
ch.close()
