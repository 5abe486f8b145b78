"""How much of one chain's decoder can hide behind another chain's front end on the same GPU?

python tools/bench_two_chains.py [--chains 2] [--blocks 2048] [--steps 100]
K independent chains (one FM stereo stream each, their own HIP streams), the same resident input, calls issued round
robin without synchronising: the GPU is free to run chain A's decoder beside chain B's fused front end.  Prints the
aggregate rate for 1 .. K chains.  (A feasibility probe for cross-call overlap inside ONE chain, DESIGN.md.)
"""
import argparse, importlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=2)
ap.add_argument("--blocks", type=int, default=2048)
ap.add_argument("--steps", type=int, default=100)
args = ap.parse_args()
fmr = importlib.import_module("airspy-fmradion_amd")
dev = torch.device("cuda", 0)
blk, fs = 65536, 10e6
n = args.blocks * blk
iq = bench.synth_fm_stereo_torch(n, fs, 0, dev)[None]
block_len = [blk] * args.blocks
max_au = int(n * 0.0048) + 64
out = {}
for K in range(1, args.chains + 1):
    chains = [fmr.Chain(mode=fmr.MODE_FM, input_rate=fs, enable_resampler=True, stereo=True, n_streams=1,
                        max_block_len=blk, max_blocks=args.blocks, device=0) for _ in range(K)]
    audio = [torch.zeros((1, 2 * max_au), dtype=torch.float64, device=dev) for _ in range(K)]
    def step(i):
        return chains[i].process_blocks_device(iq.data_ptr(), n, block_len, audio[i].data_ptr(), audio[i].shape[1], sync=False)
    for w in range(25):
        for i in range(K): step(i)
    for c in chains: c.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        for i in range(K): step(i)
    for c in chains: c.synchronize()
    dt = time.perf_counter() - t0
    locked = [c.status(0).stereo_detected for c in chains]
    out[K] = {"chains": K, "agg_GSps": round(K * args.steps * n / dt / 1e9, 2), "ms_per_chain_step": round(dt / args.steps * 1e3, 4), "locked": locked}
    print(json.dumps(out[K]), flush=True)
    for c in chains: c.close()
