#!/usr/bin/env python3
"""bench.py -- throughput of the FM-stereo hot path on MI355X.

Workload (BASELINE.json configs[1]): one FM stereo stream per GPU, 10 MS/s
complex-float IQ resident in HBM, PilotPhaseLock on; a "step" is one pass of the
whole chain (IfResampler -> FmDecoder -> f64 stereo audio at 48 kHz) over one
batch of `--blocks` consecutive 65536-sample blocks.  N GPUs = N independent
streams, one process per GPU, no collective on the data path (weak scaling).

Prints ONE JSON line on rank 0: value = whole-job IQ MS/s, plus
  roofline     -- the HBM-bound front-end kernel (ifr_decim), timed with HIP
                  events on the chain's own stream inside the timed region
  cpu_baseline -- the CPU oracle (port of the reference algorithm) on the same
                  workload, 1 core, bounded sample (rank 0, N=1 only)
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
BLK = 65536                    # Airspy block length (main.cpp:687)
FS = 10e6


def synth_fm_stereo_torch(n, fs, stream_id, device):
    """S-FMst generated on the GPU; same formula as tests/siggen.py, except that
    every tone is snapped to an integer number of cycles in the n-sample buffer, so that replaying
    the buffer step after step is one continuous stream (no pilot-phase jump at the seam)."""
    import torch
    T = n / fs

    def snap(f):
        return round(f * T) / T

    t = torch.arange(n, dtype=torch.float64, device=device) / fs
    fl, fr, fp = snap(1000.0 + 10.0 * stream_id), snap(400.0 + 10.0 * stream_id), snap(19000.0)
    left, right = torch.sin(2 * np.pi * fl * t), torch.sin(2 * np.pi * fr * t)
    th = 2 * np.pi * fp * t
    mpx = 0.45 * (left + right) + 0.10 * torch.sin(th) + 0.45 * (left - right) * torch.sin(2 * th)
    mpx = mpx - mpx.mean()                      # exact zero mean: the FM phase closes on itself
    ph = 2 * np.pi * 75000.0 / fs * torch.cumsum(mpx, 0)
    g = torch.Generator(device=device)
    g.manual_seed(1 + stream_id)
    noise = torch.randn(n, 2, dtype=torch.float32, device=device, generator=g) * 1e-3
    iq = torch.stack((0.3 * torch.cos(ph), 0.3 * torch.sin(ph)), dim=1).to(torch.float32) + noise
    return iq.contiguous()   # (n, 2) float32 == interleaved complex float


def cpu_baseline(target_seconds=12.0):
    """Time the CPU oracle (1 core) on the same workload; bounded sample."""
    import oracle_py as ora
    import siggen
    pilotcut = np.load(os.path.join(ROOT, "tests", "golden", "filters", "jj1bdx_48khz_fmaudio.npy"))
    ifr = ora.IfResampler(FS, 384e3)
    fm = ora.FmDecoder(False, np.array([0, 1, 0], dtype=np.float32), True, 50.0, False, 0, pilotcut)
    x = siggen.fm_stereo_iq(64 * BLK, FS)
    t0 = time.perf_counter()
    for b in siggen.blocks(x[:8 * BLK], BLK):
        fm.process(ifr.process(b))
    rate = 8 * BLK / (time.perf_counter() - t0)
    reps = int(max(1, min(400, round(target_seconds * rate / len(x)))))
    nblk = 64 * reps
    t0 = time.perf_counter()
    for _ in range(reps):
        for b in siggen.blocks(x, BLK):
            fm.process(ifr.process(b))
    dt = time.perf_counter() - t0
    return {"value": round(nblk * BLK / dt / 1e6, 3), "unit": "MS/s", "cores": 1, "kind": "port",
            "sample": f"{nblk} blocks x {BLK} IQ samples (a 64-block S-FMst stream replayed {reps}x, {dt:.1f} s of CPU), "
                      "oracle = C restatement of IfResampler+FmDecoder, gcc -O3 no fast-math"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=2048,
                    help="65536-sample blocks per step (2048 = 2^27 samples = 1 GiB of IQ, SURVEY.md 8d)")
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--multipath-stages", type=int, default=0,
                    help="configs[3]: FM stereo with the MultipathFilter equaliser (-E N); 0 = configs[1], the headline")
    ap.add_argument("--input-format", choices=["cf32", "s16", "u8"], default="cf32",
                    help="source sample format the front-end kernel reads (the headline metric is cf32)")
    ap.add_argument("--no-region-events", action="store_true",
                    help="diagnostic: no HIP events inside the timed region (roofline from the instrumented step)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    fmr = importlib.import_module("airspy-fmradion_amd")
    S, B = args.streams, args.blocks
    n = B * BLK
    iq = torch.stack([synth_fm_stereo_torch(n, FS, rank * S + s, dev) for s in range(S)])  # (S, n, 2)
    max_au = int(n * 0.0048) + 64
    audio = torch.zeros((S, 2 * max_au), dtype=torch.float64, device=dev)
    fmt = {"cf32": 0, "s16": 1, "u8": 2}[args.input_format]
    bps = {0: 8, 1: 4, 2: 2}[fmt]
    if fmt == 1:      # quantise the same stream to the FileSource default format (S16_LE)
        iq = torch.round(iq / 0.3 * 0.8 * 32767.0).to(torch.int16).contiguous()
    elif fmt == 2:    # RTL-SDR offset binary
        iq = (torch.round(iq / 0.3 * 0.8 * 127.0) + 128).to(torch.uint8).contiguous()
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=FS, enable_resampler=True, stereo=True, n_streams=S,
                   max_block_len=BLK, max_blocks=B, device=local_rank, input_format=fmt,
                   multipath_stages=args.multipath_stages)
    block_len = [BLK] * B
    torch.cuda.synchronize()

    def step():
        return ch.process_blocks_device(iq.data_ptr(), n, block_len, audio.data_ptr(), audio.shape[1], sync=False)

    # Set-up, not warm-up: one call brings the chain from its cold state (initial AGC gain, PLL unlocked) into
    # lock -- the metric's configuration is "PilotPhaseLock on", i.e. the locked steady state.  The cold call is
    # reported separately (cold_first_call_ms); the W warm-up steps below are ordinary locked steps.
    t_c = time.perf_counter()
    step()
    ch.synchronize()
    cold_ms = (time.perf_counter() - t_c) * 1e3
    for _ in range(args.warmup):
        step()
    ch.synchronize()
    # Timed region: only the dominant kernel carries HIP events (two per step, on the chain's own
    # stream); the host never synchronises inside the region, so launches run ahead of the GPU.
    ch.enable_kernel_timing(0 if args.no_region_events else 2)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        alen = step()
    t_enq = time.perf_counter() - t0          # host time to enqueue all K steps (launch-bound check)
    ch.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dom = [ms for name, ms in ch.kernel_times() if name == "ifr_decim"]
    assert len(dom) == (0 if args.no_region_events else args.steps)
    # one extra, untimed step with every kernel instrumented: the per-kernel table
    ch.enable_kernel_timing(1)
    step()
    ktot = {}
    for name, ms in ch.kernel_times():
        a = ktot.setdefault(name, [0.0, 0])
        a[0] += ms
        a[1] += 1
    ch.enable_kernel_timing(0)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_samples = world * S * n * args.steps
    value = total_samples / dt / 1e6
    st = ch.status(0)
    assert st.stereo_detected == 1, "PLL did not lock: the timed work is not the stereo path"
    assert int(alen.sum()) > 0 and bool(torch.isfinite(audio[0, :int(alen.sum())]).all())

    if rank == 0:
        kavg = {k: v[0] / v[1] for k, v in ktot.items()}
        dec_ms = float(np.mean(dom)) if dom else kavg.get("ifr_decim", 0.0)   # average launch duration over the K timed steps
        # HBM bytes of the dominant kernel from the rocprofv3 --pmc passes of the same command
        # (profiles/r01_pmc_traffic.json, corrected as MI355X_MICROARCH.md prescribes); PMC counters
        # cannot be read inside this process, so the figure is only attached for the profiled batch size.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if pmc.get("blocks_per_step") == B and S == 1 and fmt == 0:
                traffic = [v["hbm_bytes"] for k, v in pmc["kernels"].items() if "k_ifr_decim" in k][0]
        except Exception:
            traffic = None
        bytes_per_launch = float(bps) * S * n     # algorithmic: 8 B per cf32 input IQ sample; 4 / 2 B for s16 / u8
        achieved = bytes_per_launch / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        out = {
            "metric": "IQ MS/s (FM stereo, 10 MS/s in), whole job",
            "value": round(value, 3), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 front end / f64 after the discriminator", "data": "synthetic",
            "config": {"workload": ("configs[1]: single FM stereo stream per GPU, 10 MS/s complex-float IQ in HBM, "
                                    "PilotPhaseLock on, IfResampler+FmDecoder -> f64 stereo 48 kHz") if not args.multipath_stages
                       else f"configs[3]: as configs[1] with the MultipathFilter equaliser -E {args.multipath_stages}",
                       "input_format": args.input_format, "streams_per_gpu": S, "blocks_per_step": B, "block_len": BLK,
                       "samples_per_step_per_gpu": S * n, "per_gpu_msps": round(value / world, 3),
                       "resampler": ch.resampler_info()},
            "roofline": {"bound": "hbm", "kernel": "ifr_decim (front-end stage A, reads every IQ sample)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "avg_launch_ms": round(dec_ms, 5), "algorithmic_bytes_per_launch": bytes_per_launch},
            "kernel_ms_per_step": {k: round(v, 5) for k, v in kavg.items()},
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4),
            "cold_first_call_ms": round(cold_ms, 2),
            "audio_check": {"stereo_locked": int(st.stereo_detected), "pilot_level": round(st.pilot_level, 6)},
            "recurrences": {"agc_newton_rounds": st.agc_iterations, "pll_newton_rounds": st.pll_iterations,
                            "pll_residuals": [float("%.3g" % v) for v in st.pll_residual_history[:st.pll_iterations]],
                            "pll_mismatches": [float("%.3g" % v) for v in st.pll_mismatch_history[:st.pll_iterations]],
                            "agc_serial_fallback": st.agc_fallback, "pll_serial_fallback": st.pll_fallback},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
