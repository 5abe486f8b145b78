"""Rate of the equaliser chain alone: nanoseconds per group of four IF samples (one coefficient update,
MultipathFilter.cpp:176,186) of k_mpf4, and per sample of the AGC kernel that feeds it, from the chain's own event pairs.
Usage (GPU box): python tools/mpf_rate.py"""
import importlib, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import siggen
fmr = importlib.import_module("airspy-fmradion_amd")
fs, blk, nblk, batch = 384e3, 2517, 256, 64
x = siggen.two_ray(siggen.fm_stereo_iq(nblk * blk, fs), 20)
ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=fs, stereo=True, multipath_stages=64, max_block_len=blk, max_blocks=batch)
for i in range(0, 128, batch):
    ch.process_blocks(x[None, i * blk:(i + batch) * blk], [blk] * batch)
ch.enable_kernel_timing(1)
mpf, agc = [], []
for i in range(128, nblk, batch):
    ch.process_blocks(x[None, i * blk:(i + batch) * blk], [blk] * batch)
    t = dict(ch.kernel_times())
    mpf.append(t.get("mpf", 0.0)); agc.append(t.get("if_agc", 0.0))
n = (nblk - 128) * blk
mode, kern = "AGC beside", "k_mpf4"
print("%s, %s: equaliser %.0f ns per group of four samples; AGC kernel %.1f ns per sample" % (kern, mode, sum(mpf) * 1e6 / (n / 4.0), sum(agc) * 1e6 / n))
