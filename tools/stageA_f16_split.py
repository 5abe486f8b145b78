#!/usr/bin/env python3
"""Feasibility of stage A of the 10 MS/s front end (103-tap equiripple decimator, D = 10) on the fp16 matrix cores with the
three-product split k_ifr_poly5h uses for the R8B stage B: x = xh + xl, h = hh + hl (fp16 terms, power-of-two scaling per
tile / per table), hh xh + hh xl + hl xh accumulated in fp32.  CPU model of the arithmetic (numpy), no GPU: error of the mid
samples against fp64, beside the error of a plain fp32 accumulation (what the kernel does now), and the share of a banded
16 x 32 tile that holds taps.   python tools/stageA_f16_split.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import siggen

fmr = importlib.import_module("airspy-fmradion_amd")
D, n = 10, 400000
taps, info = fmr.design_taps_class(10e6, 384e3, fmr.RESAMPLER_FAST, 0)      # stage A of the FAST class (host arithmetic, no GPU)
taps = np.asarray(taps, dtype=np.float64)
NA = len(taps)
assert info["D"] == D and NA == info["NA"], info
x = siggen.fm_stereo_iq(n, 10e6).astype(np.complex64)
for amp, name in ((1.0, "full scale (0.3)"), (1e-3, "1e-3 of it")):
    xs = (x * amp).astype(np.complex64)
    m = (n - NA) // D
    idx = (np.arange(m) * D)[:, None] + np.arange(NA)[None, :]
    win_r, win_i = xs.real[idx].astype(np.float64), xs.imag[idx].astype(np.float64)
    h = taps[::-1]
    ref = win_r @ h + 1j * (win_i @ h)
    # plain fp32 accumulation in tap order
    acc = np.zeros(m, np.float32); acci = np.zeros(m, np.float32)
    h32 = h.astype(np.float32)
    for t in range(NA):
        acc = (acc + xs.real[idx[:, t]] * h32[t]).astype(np.float32); acci = (acci + xs.imag[idx[:, t]] * h32[t]).astype(np.float32)
    e32 = np.sqrt(np.mean(np.abs((acc + 1j * acci) - ref) ** 2)) / np.sqrt(np.mean(np.abs(ref) ** 2))
    # three-product split: taps scaled so that the largest sits in [512, 1024), window scaled per tile of 256 outputs likewise
    sh = 2.0 ** (9 - np.floor(np.log2(np.max(np.abs(h)))))
    hh = (h * sh).astype(np.float16); hl = ((h * sh) - hh.astype(np.float64)).astype(np.float16)
    out = np.zeros(m, np.complex128)
    for t0 in range(0, m, 256):
        sl = slice(t0, min(m, t0 + 256))
        wr, wi = win_r[sl], win_i[sl]
        sx = 2.0 ** (9 - np.floor(np.log2(max(np.max(np.abs(wr)), np.max(np.abs(wi)), 1e-30))))
        def split(w):
            a = (w * sx).astype(np.float16); b = ((w * sx) - a.astype(np.float64)).astype(np.float16)
            return a.astype(np.float32), b.astype(np.float32)
        rh, rl = split(wr); ih, il = split(wi)
        H, L = hh.astype(np.float32), hl.astype(np.float32)
        def prod(ah, al):       # fp32 accumulation (the MFMA accumulates in fp32; products of two fp16 values are exact in fp32)
            return (ah @ H + al @ H + ah @ L).astype(np.float32)
        out[sl] = (prod(rh, rl).astype(np.float64) + 1j * prod(ih, il).astype(np.float64)) / (sh * sx)
    e16 = np.sqrt(np.mean(np.abs(out - ref) ** 2)) / np.sqrt(np.mean(np.abs(ref) ** 2))
    print("input %-18s mid samples, relative RMS error against fp64: fp32 accumulation %.2e, fp16 three-product split %.2e" % (name, e32, e16))
# tile density: 16 outputs (10 inputs apart) x 32 consecutive inputs per MFMA; taps of output r cover inputs 10 r .. 10 r + NA - 1
span = 10 * 15 + NA
tiles = -(-span // 32)
print("banded product: 16 outputs span %d inputs = %d k-tiles of 32; %d of %d tile entries hold a tap (%.0f %%)" % (span, tiles, 16 * NA, 16 * 32 * tiles, 100.0 * 16 * NA / (16 * 32 * tiles)))
