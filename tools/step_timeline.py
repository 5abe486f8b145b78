#!/usr/bin/env python3
"""Schedule of the chain's streams over a few steady-state steps of the benchmark workload, as the GPU ran it.

python tools/step_timeline.py [--blocks 2048] [--steps 8] [--show 2] [--out file]
Uses the library's own trace mode (fmr_enable_kernel_timing(3): an event pair around every instrumented kernel on the
stream it runs on, fmr_get_kernel_trace) -- no profiler in the host's launch path, so the gaps are the GPU's.  The event
markers themselves cost a few microseconds per kernel: the step is a few per cent longer than in bench.py.
"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--show", type=int, default=2, help="front-end periods printed")
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--out", default="")
    ap.add_argument("--r8b", action="store_true", help="the R8B resampler class (three-kernel front end, fp16 stage B)")
    ap.add_argument("--if-filter", action="store_true", help="the IF filter (-f medium) behind the fused front end")
    ap.add_argument("--sigma", type=float, default=1e-3, help="noise per I / Q component (bench.py --sigma)")
    ap.add_argument("--am", action="store_true", help="config 3: AM 384 kS/s -> 48 k narrow (8192 blocks of 2048 samples per step)")
    args = ap.parse_args()
    import torch
    fmr = importlib.import_module("airspy-fmradion_amd")
    dev = torch.device("cuda", 0)
    B, blk = args.blocks, bench.BLK
    if args.am:
        B, blk = (8192 if args.blocks == 2048 else args.blocks), bench.AM_BLK
    n = B * blk
    if args.am:
        iq = torch.stack([bench.synth_am_torch(n, bench.AM_FS, 0, dev)])
        audio = torch.zeros((1, int(n * 0.125) + 64), dtype=torch.float64, device=dev)
    else:
        iq = torch.stack([bench.synth_fm_stereo_torch(n, bench.FS, 0, dev, sigma=args.sigma)])
        audio = torch.zeros((1, 2 * (int(n * 0.0048) + 64)), dtype=torch.float64, device=dev)
    kw = {}
    if args.r8b:
        kw["resampler_class"] = fmr.RESAMPLER_R8B
    if args.if_filter:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import load_filter
        kw.update(fmfilter_enable=True, filter_coeff=load_filter("jj1bdx_fm_384kHz_medium"))
    if args.am:
        import numpy as np
        narrow = np.load(os.path.join(ROOT, "tests", "golden", "filters", "jj1bdx_am_48khz_narrow.npy"))
        ch = fmr.Chain(mode=fmr.MODE_AM, input_rate=bench.AM_FS, enable_resampler=True, filter_coeff=narrow, max_block_len=blk, max_blocks=B)
    else:
        ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=bench.FS, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=B, **kw)
    bl = [blk] * B

    def step():
        ch.process_blocks_device(iq.data_ptr(), n, bl, audio.data_ptr(), audio.shape[1], sync=False)

    for _ in range(1 + args.warmup):
        step()
    ch.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    ch.synchronize()
    plain_ms = (time.perf_counter() - t0) / 20 * 1e3
    ch.enable_kernel_timing(3)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    tr = ch.kernel_trace()
    traced_ms = (time.perf_counter() - t0) / args.steps * 1e3
    ch.enable_kernel_timing(0)
    ch.close()
    tr.sort(key=lambda r: r[2])
    fe = [r for r in tr if r[0] in ("ifr_fused", "ifr_decim")]
    lines = ["# %d blocks per step; %.4f ms per step untraced, %.4f ms with the trace's event markers" % (B, plain_ms, traced_ms),
             "# front-end start-to-start (us): " + " ".join("%.0f" % ((b[2] - a[2]) * 1e3) for a, b in zip(fe, fe[1:])),
             "# columns: start us, duration us, stream (decoder | side | agc [in-order chain only] | tail), kernel"]
    if len(fe) > args.show + 2:
        w0, w1 = fe[-args.show - 2][2], fe[-2][2]
        cols = {0: 0, 1: 1, 2: 2, 3: 3, 4: 3}
        names = ["dec ", "side", "agc ", "fe  ", "tail"]
        for name, st, a, b in tr:
            if b < w0 or a > w1:
                continue
            lines.append("%9.1f %8.1f  %s %s%s" % ((a - w0) * 1e3, (b - a) * 1e3, names[st], "      " * cols[st], name))
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
