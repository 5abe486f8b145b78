#!/usr/bin/env python3
"""The first calls after a synchronisation, kernel by kernel (fmr_enable_kernel_timing(3)): where a short timed region
(the driver's 20 steps) spends what a long one does not.   python tools/startup_trace.py [--calls 3]"""
import argparse, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--calls", type=int, default=3); args = ap.parse_args()
import torch
fmr = importlib.import_module("airspy-fmradion_amd")
dev = torch.device("cuda", 0)
B, blk = 2048, bench.BLK
n = B * blk
iq = torch.stack([bench.synth_fm_stereo_torch(n, bench.FS, 0, dev)])
audio = torch.zeros((1, 2 * (int(n * 0.0048) + 64)), dtype=torch.float64, device=dev)
ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=bench.FS, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=B)
step = lambda: ch.process_blocks_device(iq.data_ptr(), n, [blk] * B, audio.data_ptr(), audio.shape[1], sync=False)
for _ in range(30):
    step()
ch.synchronize()
for k in (1, 2, 3, 5):            # untraced: total time of k calls + synchronise
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        step()
    te = time.perf_counter() - t0
    ch.synchronize(); torch.cuda.synchronize()
    print("%d call(s): enqueued after %.3f ms, done after %.3f ms" % (k, te * 1e3, (time.perf_counter() - t0) * 1e3))
ch.enable_kernel_timing(3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.calls):
    step()
te = time.perf_counter() - t0
tr = ch.kernel_trace()
print("# traced: %d calls enqueued after %.3f ms, trace read after %.3f ms" % (args.calls, te * 1e3, (time.perf_counter() - t0) * 1e3))
tr.sort(key=lambda r: r[2])
b0 = tr[0][2]
names = ["dec ", "side", "agc ", "fe  ", "tail"]
for name, st, a, b in tr:
    if name in ("pll",):
        continue
    print("%9.1f %8.1f  %s %s" % ((a - b0) * 1e3, (b - a) * 1e3, names[st], name))
ch.close()
