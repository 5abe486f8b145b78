#!/usr/bin/env python3
"""Average duration of the chain's instrumented kernels over a few steady-state steps (the chain's own trace mode, as
tools/step_timeline.py; no lock assertion: usable with ablation builds whose output is garbage).
python tools/kernel_avg.py [--r8b | --am | --if-filter] [--steps 16]"""
import argparse, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--r8b", action="store_true"); ap.add_argument("--am", action="store_true"); ap.add_argument("--if-filter", action="store_true")
    ap.add_argument("--steps", type=int, default=16); ap.add_argument("--tag", default="")
    ap.add_argument("--front-only", action="store_true", help="MODE_NONE: the resampler alone (no decoder behind it, nothing beside it)")
    args = ap.parse_args()
    import numpy as np, torch
    fmr = importlib.import_module("airspy-fmradion_amd")
    dev = torch.device("cuda", 0)
    if args.am:
        B, blk = 8192, bench.AM_BLK
        n = B * blk
        iq = torch.stack([bench.synth_am_torch(n, bench.AM_FS, 0, dev)])
        audio = torch.zeros((1, int(n * 0.125) + 64), dtype=torch.float64, device=dev)
        ch = fmr.Chain(mode=fmr.MODE_AM, input_rate=bench.AM_FS, enable_resampler=True, max_block_len=blk, max_blocks=B,
                       filter_coeff=np.load(os.path.join(ROOT, "tests", "golden", "filters", "jj1bdx_am_48khz_narrow.npy")))
    else:
        B, blk = 2048, bench.BLK
        n = B * blk
        iq = torch.stack([bench.synth_fm_stereo_torch(n, bench.FS, 0, dev)])
        audio = torch.zeros((1, 2 * (int(n * 0.0048) + 64)), dtype=torch.float64, device=dev)
        kw = {}
        if args.r8b: kw["resampler_class"] = fmr.RESAMPLER_R8B
        if args.if_filter:
            kw.update(fmfilter_enable=True, filter_coeff=np.load(os.path.join(ROOT, "tests", "golden", "filters", "jj1bdx_fm_384kHz_medium.npy")))
        if args.front_only:
            ch = fmr.Chain(mode=fmr.MODE_NONE, input_rate=bench.FS, enable_resampler=True, max_block_len=blk, max_blocks=B, **kw)
        else:
            ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=bench.FS, enable_resampler=True, stereo=True, max_block_len=blk, max_blocks=B, **kw)
    bl = [blk] * B
    step = lambda: ch.process_blocks_device(iq.data_ptr(), n, bl, audio.data_ptr(), audio.shape[1], sync=False)
    for _ in range(60): step()
    ch.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): step()
    ch.synchronize()
    plain = (time.perf_counter() - t0) / 40 * 1e3
    ch.enable_kernel_timing(3)
    for _ in range(args.steps): step()
    tr = ch.kernel_trace()
    ch.enable_kernel_timing(0); ch.close()
    acc = {}
    for name, st, a, b in tr: acc.setdefault(name, []).append((b - a) * 1e3)
    print("%-10s %.4f ms/step untraced | " % (args.tag, plain) + "  ".join("%s %.1f" % (k, sum(v) / len(v)) for k, v in acc.items() if sum(v) / len(v) > 15 or k.startswith("ifr") or k == "disc"))


if __name__ == "__main__":
    main()
