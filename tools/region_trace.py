#!/usr/bin/env python3
"""The first steps of a timed region, from an idle chain: front end and PLL stage of every step (the chain's own event
trace, fmr_enable_kernel_timing(3)).  Answers what bench.py's 20-step form pays on top of its steps.

python tools/region_trace.py [--steps 24] [--warmup 5] [--out file]
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--out", default="")
    ap.add_argument("--dump", default="", help="steps (1-based, comma separated) whose kernels are listed one by one")
    args = ap.parse_args()
    import torch
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

    step()
    ch.synchronize()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    ch.synchronize()
    ch.enable_kernel_timing(3)
    torch.cuda.synchronize()
    for _ in range(args.steps):
        step()
    tr = ch.kernel_trace()
    ch.enable_kernel_timing(0)
    ch.close()
    tr.sort(key=lambda r: r[2])
    fe = [r for r in tr if r[0] == "ifr_fused"]
    pll = [r for r in tr if r[0] == "pll"]
    jac = [r for r in tr if r[0] == "pll_shoot_jac"]
    t0 = fe[0][2]
    lines = ["# step: front end start (ms from the first), front end us, wait for the PLL's first pass us, Jacobian pass us, PLL stage us, "
             "start-to-start us"]
    for i, f in enumerate(fe):
        p = pll[i] if i < len(pll) else None
        j = jac[i] if i < len(jac) else None
        nxt = fe[i + 1][2] if i + 1 < len(fe) else None
        lines.append("%3d %8.3f %7.1f %7.1f %7.1f %7.1f %s" % (
            i + 1, (f[2] - t0), (f[3] - f[2]) * 1e3, ((p[2] - f[3]) * 1e3) if p else -1, ((j[3] - j[2]) * 1e3) if j else -1,
            ((p[3] - p[2]) * 1e3) if p else -1, ("%7.1f" % ((nxt - f[2]) * 1e3)) if nxt else "      -"))
    last = max(r[3] for r in tr)
    lines.append("# last kernel of the region ends %.3f ms after the first front end started; last PLL stage ended at %.3f"
                 % (last - t0, pll[-1][3] - t0))
    names = ["dec ", "side", "agc ", "fe  ", "tail"]
    cols = {0: 0, 1: 1, 2: 2, 3: 3, 4: 3}
    for d in [int(v) for v in args.dump.split(",") if v]:
        if d < 1 or d >= len(fe):
            continue
        w0, w1 = fe[d - 1][2], fe[d][2]
        lines.append("# step %d: start us, duration us, stream, kernel" % d)
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
