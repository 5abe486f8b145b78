#!/usr/bin/env python3
"""bench.py -- throughput of the FM/AM demodulation hot path on MI355X.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on): one FM stereo stream per GPU,
10 MS/s complex-float IQ resident in HBM, PilotPhaseLock on; a "step" is one pass of the whole chain (IfResampler ->
FmDecoder -> f64 stereo audio at 48 kHz) over one batch of `--blocks` consecutive 65536-sample blocks.  N GPUs = N
independent shards of streams, one process per GPU, no collective on the data path (weak scaling).

Other configs of BASELINE.json (extra lines kept under profiles/, never the headline):
  --multipath-stages 64          configs[3]  FM stereo + MultipathFilter -E 64
  --streams 32                   configs[4]  32 independent streams per GPU (the per-GPU shard of the 256-stream job)
  --mode am                      configs[2]  AM 384 kS/s -> IfResampler(48 k) -> AmDecoder narrow filter
  --api-mode block               the drop-in call: one 65536-sample block per fmr_process() through host buffers

Prints ONE JSON line on rank 0: value = whole-job IQ MS/s, plus
  roofline     -- the HBM-bound kernel that reads every IQ sample, timed with HIP events on the chain's own stream
                  inside the timed region; `stage` = the whole FIR+discriminator stage (north star's 60 % target)
  audio_check  -- RMS error of this chain's audio (its first call, cold start included) against the CPU oracle
  cpu_baseline -- the CPU oracle (port of the reference algorithm) on the same workload: 1 core, and N streams on
                  N cores (bounded sample; rank 0, N=1 only)
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
AM_BLK, AM_FS = 2048, 384e3    # FileSource default block length (FileSource.h:34), configs[2] rate
STAGE_KERNELS = ("ifr_fused", "ifr_decim", "ifr_poly", "fm_block", "disc")


def synth_fm_stereo_torch(n, fs, stream_id, device, pilot=0.10, sigma=1e-3):
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
    mpx = 0.45 * (left + right) + pilot * torch.sin(th) + (0.45 * (left - right) * torch.sin(2 * th) if pilot > 0 else 0.0)
    del left, right, th
    mpx = mpx - mpx.mean()                      # exact zero mean: the FM phase closes on itself
    ph = 2 * np.pi * 75000.0 / fs * torch.cumsum(mpx, 0)
    del mpx, t
    g = torch.Generator(device=device)
    g.manual_seed(1 + stream_id)
    noise = torch.randn(n, 2, dtype=torch.float32, device=device, generator=g) * sigma       # per component; carrier amplitude 0.3
    iq = torch.stack((0.3 * torch.cos(ph), 0.3 * torch.sin(ph)), dim=1).to(torch.float32) + noise
    return iq.contiguous()   # (n, 2) float32 == interleaved complex float


def synth_am_torch(n, fs, stream_id, device):
    """S-AM: carrier offset +37 Hz, envelope 0.1 (1 + 0.5 sin 2 pi 1000 t), sigma 1e-4; tones snapped
    to the buffer length so that the replayed buffer is continuous."""
    import torch
    T = n / fs
    t = torch.arange(n, dtype=torch.float64, device=device) / fs
    f_off, f_tone = round(37.0 * T) / T, round((1000.0 + 10.0 * stream_id) * T) / T
    env = 0.1 * (1 + 0.5 * torch.sin(2 * np.pi * f_tone * t))
    ph = 2 * np.pi * f_off * t
    g = torch.Generator(device=device)
    g.manual_seed(3 + stream_id)
    noise = torch.randn(n, 2, dtype=torch.float32, device=device, generator=g) * 1e-4
    iq = torch.stack((env * torch.cos(ph), env * torch.sin(ph)), dim=1).to(torch.float32) + noise
    return iq.contiguous()


# ----------------------------------------------------------------------------- CPU oracle legs
R8B = False       # set by main(): the oracle's IfResampler then takes the r8brain-class specification too


IF_FILTER = False         # --if-filter: the FM IF filter "medium" (main.cpp -f medium) between resampler and discriminator


def _oracle_chain(mode, stages=0):
    import oracle_py as ora
    if mode == "am":
        narrow = np.load(os.path.join(ROOT, "tests", "golden", "filters", "jj1bdx_am_48khz_narrow.npy"))
        ifr, dec = ora.IfResampler(AM_FS, 48e3), ora.AmDecoder(narrow, ora.MODE_AM)
    else:
        pilotcut = np.load(os.path.join(ROOT, "tests", "golden", "filters", "jj1bdx_48khz_fmaudio.npy"))
        ifr = ora.IfResampler(FS, 384e3, 180.0, 0.98, True) if R8B else ora.IfResampler(FS, 384e3)
        fir = (np.load(os.path.join(ROOT, "tests", "golden", "filters", "jj1bdx_fm_384kHz_medium.npy")) if IF_FILTER
               else np.array([0, 1, 0], dtype=np.float32))
        dec = ora.FmDecoder(IF_FILTER, fir, True, 50.0, False, stages, pilotcut)
    return ifr, dec


def _cpu_worker(args):
    """One stream on one core for ~`seconds` of CPU: returns (samples, elapsed)."""
    mode, stages, stream_id, seconds = args
    import siggen
    blk = AM_BLK if mode == "am" else BLK
    nb = 512 if mode == "am" else 16
    x = siggen.am_iq(nb * blk, AM_FS) if mode == "am" else siggen.fm_stereo_iq(nb * blk, FS, stream_id=stream_id)
    ifr, dec = _oracle_chain(mode, stages)
    done, t0 = 0, time.perf_counter()
    while True:
        for i in range(0, len(x), blk):
            dec.process(ifr.process(x[i:i + blk]))
        done += len(x)
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return done, dt


def csrc_hash():
    """sha256 (16 hex digits) over the kernel sources: stamps the PMC summary with the code it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "airspy-fmradion_amd", "csrc", "*"))):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def committed_pmc_traffic(dom_name, blocks, streams, args):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/rNN_pmc_traffic.json,
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, tools/collect_profiles.sh): counters
    cannot be read from inside the timed run.  None when no summary matches this configuration."""
    import glob
    if streams != 1 or args.mode != "fm" or args.input_format != "cf32" or args.multipath_stages or args.no_pilot or args.if_filter:
        return None, None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(fn))
        except (OSError, ValueError):
            continue
        if d.get("blocks_per_step") != blocks:
            continue
        if d.get("csrc_sha256_16") != csrc_hash():
            # the counters were collected on other kernel sources: not this code's traffic
            return None, os.path.relpath(fn, ROOT) + " is stale (kernel sources changed since the PMC pass): traffic dropped"
        for k, v in d.get("kernels", {}).items():
            if ("k_" + dom_name) in k:
                return v["hbm_bytes"], os.path.relpath(fn, ROOT) + " (PMC pass of the same command on these kernel sources; read bytes = 2 x FETCH_SIZE x 1024, gfx950)"
    return None, None


def verify_timed_step(iq0, audio_last, alen_last, B, blk, blocks_done, n_cmp=20, settle=150):
    """Oracle check of the LAST TIMED step.  The oracle cannot follow the whole run (57 MS/s), but every recurrence of the
    chain forgets: started cold at a stream position where the resampler phase repeats (a multiple of 625 blocks of
    65536 samples = 24 x 65536 IF samples) and `settle` blocks before the compared ones, it agrees with an oracle that
    ran from sample 0 to 4e-12 (tests/test_oracle_end_to_end.py).  iq0: stream 0's periodic buffer (B blocks)."""
    g_end = blocks_done                                  # the last timed step covered stream blocks [g_end - B, g_end)
    q = max(0, (g_end - (settle + n_cmp)) // 625)
    g0 = 625 * q
    if g_end - g0 > 1500 or g_end - g0 < n_cmp + (settle if g0 else 0):
        return None                                      # (cannot happen for B >= 20: kept as a guard)
    ifr, dec = _oracle_chain("fm")
    ref = []
    for g in range(g0, g_end):
        b = g % B
        a = dec.process(ifr.process(iq0[b * blk:(b + 1) * blk]))
        if g >= g_end - n_cmp:
            ref.append(a)
    offs = np.concatenate([[0], np.cumsum(alen_last)]).astype(np.int64)
    errs, n = 0.0, 0
    for i, a in enumerate(ref):
        loc = B - n_cmp + i
        got = audio_last[offs[loc]:offs[loc + 1]]
        assert len(got) == len(a), (len(got), len(a), loc)
        errs += float(np.sum((got - a) ** 2))
        n += len(a)
    return {"rms_err_vs_oracle": float("%.3e" % np.sqrt(errs / max(n, 1))), "blocks_compared": n_cmp, "audio_samples_compared": n,
            "oracle_started_at_stream_block": g0, "stream_blocks_before_the_compared_ones": g_end - n_cmp - g0,
            "what": "stream 0, the last blocks of the LAST TIMED step"}


def cpu_baseline(mode, stages, seconds=8.0, max_procs=None):
    """The CPU oracle on the same workload: (i) 1 core, 1 stream -- the reference is single-threaded per stream;
    (ii) N streams on N cores, N = the box's core count."""
    import multiprocessing as mp
    n1, t1 = _cpu_worker((mode, stages, 0, seconds))
    ncores = os.cpu_count() or 1
    if max_procs:
        ncores = min(ncores, max_procs)
    with mp.get_context("fork").Pool(ncores) as pool:
        res = pool.map(_cpu_worker, [(mode, stages, s, seconds) for s in range(ncores)])
    agg = sum(n / t for n, t in res)
    return {"value": round(n1 / t1 / 1e6, 3), "unit": "MS/s", "cores": 1, "kind": "port",
            "sample": f"one stream replayed for {t1:.1f} s of CPU ({n1} IQ samples); oracle = C restatement of "
                      f"IfResampler + {'AmDecoder' if mode == 'am' else 'FmDecoder'} (VOLK-generic semantics), gcc -O3 no fast-math",
            "all_cores": {"value": round(agg / 1e6, 3), "unit": "MS/s", "cores": ncores,
                          "sample": f"{ncores} streams on {ncores} processes, {seconds:.0f} s each"}}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` with no launcher in the environment: start one process per GPU here (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* as torch.distributed.run would set them), wait for all of them and pass rank 0's JSON line
    through.  A run that asks for N GPUs can therefore never quietly measure one."""
    import subprocess
    n = args.gpus
    if not args.cpu_dry_run:
        import torch
        have = torch.cuda.device_count()
        assert have >= n, f"--gpus {n} but only {have} GPU(s) are visible"
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # wait for all of them; if one rank dies the others would sit in a collective until its time-out: stop them (our own
    # children, by handle) and report
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        failed = next((r for r, p in enumerate(procs) if p.poll() not in (None, 0)), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"rank return codes {rcs}" + (f" (rank {failed} failed first; the others were stopped)" if failed is not None else ""))
    if args.all_configs:
        run_other_configs(n)           # each of them starts its own n ranks the same way


def _gpu_numa_cores(local_rank):
    """Cores of the NUMA node the rank's GPU hangs off (sysfs: the PCI device's numa_node and that node's cpulist), or None."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cores = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cores.update(range(int(a), int(b or a) + 1))
        return sorted(cores)
    except Exception:
        return None


def pin_rank_to_cores(use_gpu_topology=True):
    """The rank's host thread on cores near its GPU: a step is ~95 kernel launches from one host thread, and N ranks enqueueing
    from the same few cores would contend.  The cores of the GPU's NUMA node, shared evenly between the ranks whose GPUs sit on
    the same node; an equal slice of the allowed cores when the topology cannot be read (or on CPU)."""
    lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if lw <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if use_gpu_topology:
            near = [_gpu_numa_cores(r) for r in range(lw)]
            if all(c for c in near):
                mine = [c for c in near[lr] if c in allowed]
                peers = [r for r in range(lw) if near[r] == near[lr]]          # ranks sharing this node
                per = len(mine) // len(peers)
                if per >= 1:
                    i = peers.index(lr)
                    os.sched_setaffinity(0, mine[i * per:(i + 1) * per])
                    return "numa"
        per = len(allowed) // lw
        if per >= 1:
            os.sched_setaffinity(0, allowed[lr * per:(lr + 1) * per])
            return "slice"
    except OSError:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=0,
                    help="blocks per step (default 2048 x 65536 = 2^27 samples = 1 GiB of IQ per stream, SURVEY.md 8d; "
                         "AM: 8192 x 2048)")
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU")
    ap.add_argument("--mode", choices=["fm", "am"], default="fm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--multipath-stages", type=int, default=0,
                    help="configs[3]: FM stereo with the MultipathFilter equaliser (-E N); 0 = configs[1], the headline")
    ap.add_argument("--input-format", choices=["cf32", "s16", "u8"], default="cf32",
                    help="source sample format the front-end kernel reads (the headline metric is cf32)")
    ap.add_argument("--spinup-ms", type=float, default=60.0,
                    help="ordinary steps run for this long BEFORE the W warm-up steps (0: none).  The GPU's shader clock ramps over "
                         "tens of milliseconds of load after an idle gap (the oracle checks between set-up and warm-up are one), "
                         "and a 5 + 20-step region ends before the ramp does; a stream that runs continuously never sees it.  The "
                         "line reports the clock before the spin-up and on either side of the timed region (`clock`)")
    ap.add_argument("--no-region-events", action="store_true",
                    help="diagnostic: no HIP events inside the timed region (roofline from the instrumented step)")
    ap.add_argument("--api-mode", choices=["batch", "block"], default="batch",
                    help="block: one block per fmr_process() call through host buffers (the drop-in call, PCIe-inclusive; "
                         "never the headline)")
    ap.add_argument("--api-batch", type=int, default=1,
                    help="with --api-mode block: blocks held back and decoded per call (the facade's set_batch_blocks)")
    ap.add_argument("--no-pilot", action="store_true",
                    help="FM stereo decoder on a mono station (no 19 kHz pilot): the unlocked steady state, where the "
                         "PLL runs in its serial form (reported line, never the headline)")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="test hook: gloo backend, the per-rank step is the CPU oracle on a tiny sample -- exercises the "
                         "launch / barrier / aggregation contract without a GPU (tests/test_multi_process.py)")
    ap.add_argument("--resampler-class", choices=["fast", "r8b"], default="fast",
                    help="r8b: the IF resampler to the defaults of the reference's r8b::CDSPResampler24 (0.98 x Nyquist, stop "
                         "band from Nyquist, 180 dB; IfResampler.cpp:25-29) -- the reference-equivalent class, its own line, "
                         "never the headline")
    ap.add_argument("--if-filter", action="store_true",
                    help="FM: the IF filter 'medium' (main.cpp -f medium) on; the fused front end then stores IF samples")
    ap.add_argument("--sigma", type=float, default=1e-3,
                    help="noise per I / Q component of the synthetic FM signal (carrier amplitude 0.3): 1e-3 = 46.5 dB C/N (default), "
                         "1e-2 = 26.5 dB, 3e-2 = 17 dB -- what the PLL's Newton iteration needs on a noisier station (recurrences.pll_newton_rounds)")
    ap.add_argument("--no-r8b-leg", action="store_true",
                    help="default headline run only: skip the second, short region that times the reference-equivalent (R8B) resampler class")
    ap.add_argument("--all-configs", action="store_true",
                    help="after the headline line, print one line each for configs[2] (AM), configs[3] (-E 64), configs[4] "
                         "(32 streams per GPU) and the mono-station case, each with its own audio check and CPU baseline")
    args = ap.parse_args()

    global R8B, IF_FILTER
    R8B = args.resampler_class == "r8b"
    IF_FILTER = bool(args.if_filter)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)              # `python bench.py --gpus N` without a launcher: spawn the N ranks here

    import torch
    import torch.distributed as dist

    pinned = pin_rank_to_cores(use_gpu_topology=not args.cpu_dry_run)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        args.gpus = world
    am = args.mode == "am"
    blk, fs = (AM_BLK, AM_FS) if am else (BLK, FS)
    S = args.streams
    B = args.blocks or (8192 if am else 2048)
    n = B * blk

    if args.cpu_dry_run:
        return dry_run(args, rank, world, S, B, blk, pinned)

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    fmr = importlib.import_module("airspy-fmradion_amd")
    synth = synth_am_torch if am else (lambda n_, fs_, sid, d: synth_fm_stereo_torch(n_, fs_, sid, d, pilot=0.0 if args.no_pilot else 0.10, sigma=args.sigma))
    iq = torch.stack([synth(n, fs, rank * S + s, dev) for s in range(S)])  # (S, n, 2)
    max_au = int(n * (0.125 if am else 0.0048)) + 64
    audio = torch.zeros((S, (1 if am else 2) * max_au), dtype=torch.float64, device=dev)
    fmt = {"cf32": 0, "s16": 1, "u8": 2}[args.input_format]
    bps = {0: 8, 1: 4, 2: 2}[fmt]
    if fmt == 1:      # quantise the same stream to the FileSource default format (S16_LE)
        iq = torch.round(iq / 0.3 * 0.8 * 32767.0).to(torch.int16).contiguous()
    elif fmt == 2:    # RTL-SDR offset binary
        iq = (torch.round(iq / 0.3 * 0.8 * 127.0) + 128).to(torch.uint8).contiguous()
    if am:
        narrow = np.load(os.path.join(ROOT, "tests", "golden", "filters", "jj1bdx_am_48khz_narrow.npy"))
        ch = fmr.Chain(mode=fmr.MODE_AM, input_rate=fs, enable_resampler=True, filter_coeff=narrow, n_streams=S,
                       max_block_len=blk, max_blocks=B, device=local_rank, input_format=fmt)
    else:
        ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=fs, enable_resampler=True, stereo=True, n_streams=S,
                       max_block_len=blk, max_blocks=B, device=local_rank, input_format=fmt,
                       multipath_stages=args.multipath_stages,
                       **({"fmfilter_enable": True, "filter_coeff": np.load(os.path.join(
                           ROOT, "tests", "golden", "filters", "jj1bdx_fm_384kHz_medium.npy"))} if args.if_filter else {}),
                       resampler_class=fmr.RESAMPLER_R8B if args.resampler_class == "r8b" else fmr.RESAMPLER_FAST,
                       in_order=(args.api_mode == "block"))      # the host-buffer call synchronises every time
    block_len = [blk] * B
    torch.cuda.synchronize()

    if args.api_mode == "block":
        return block_api(args, ch, iq, blk, fs, rank, world, am)

    def step():
        return ch.process_blocks_device(iq.data_ptr(), n, block_len, audio.data_ptr(), audio.shape[1], sync=False)

    # Set-up, not warm-up: one call brings the chain from its cold state (initial AGC gain, PLL unlocked) into
    # lock -- the metric's configuration is "PilotPhaseLock on", i.e. the locked steady state.  The cold call is
    # reported separately (cold_first_call_ms); the W warm-up steps below are ordinary locked steps.
    t_c = time.perf_counter()
    alen0 = step()
    ch.synchronize()
    cold_ms = (time.perf_counter() - t_c) * 1e3
    # audio of the first call (cold start included) on the first blocks of stream 0: checked against the oracle below
    nchk = min(B, 4096 if am else 100)
    n_au_chk = int(alen0[:nchk].sum())
    audio_chk = audio[0, :n_au_chk].cpu().numpy().copy() if fmt == 0 else None      # every rank checks its own stream 0
    iq_chk = iq[0, :nchk * blk].cpu().numpy().view(np.complex64).reshape(-1).copy() if audio_chk is not None else None
    # Spin-up (round 6): the same step, untimed, until the shader clock has ramped.  What the ramp costs a short region:
    # profiles/r06_clock_ramp.txt (5 / 20 / 40 / 80 warm-up steps in front of 20 timed ones: 0.538 / 0.522 / 0.514 / 0.512 ms).
    mhz = [fmr.probe_shader_clock(local_rank)]
    spin_steps, t_sp = 0, time.perf_counter()
    while (time.perf_counter() - t_sp) * 1e3 < args.spinup_ms:
        step()
        spin_steps += 1
    for _ in range(args.warmup):
        step()
    ch.synchronize()
    mhz.append(fmr.probe_shader_clock(local_rank))
    # Timed region: only the stage kernels carry HIP events (two per kernel per step, on the chain's own
    # stream); the host never synchronises inside the region, so launches run ahead of the GPU.
    # (from 8 steps on the events go on every fourth step: the two markers around a stage kernel cost 7-10 us on the decoder
    # stream -- measured with and without them, 0.5787 / 0.5792 / 0.5807 against 0.5685 / 0.5737 ms per step)
    # round 6: the fused front end is timed on EVERY step with the start / stop events of its own dispatch (mode 5: no
    # markers on the stream for it); the other stage kernels (the configurations without the fused discriminator) keep
    # their marker pairs on every fourth step
    ev_mode = 0 if args.no_region_events else (5 if args.steps >= 8 else 2)
    ch.enable_kernel_timing(ev_mode)
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
    mhz.append(fmr.probe_shader_clock(local_rank))
    blocks_done = (1 + spin_steps + args.warmup + args.steps) * B     # stream blocks decoded so far (set-up call + spin-up + warm-up + timed steps)
    audio_last = audio[0, :int(alen.sum())].cpu().numpy().copy()      # stream 0 of this rank, the last TIMED step
    region = {}
    for name, ms in ch.kernel_times():
        region.setdefault(name, []).append(ms)
    if ev_mode == 2:
        assert all(len(v) == args.steps for v in region.values()), {k: len(v) for k, v in region.items()}
    elif ev_mode == 5:
        assert region and all(args.steps // 4 <= len(v) <= args.steps // 4 + 1 or (k == "ifr_fused" and len(v) == args.steps)
                              for k, v in region.items()), {k: len(v) for k, v in region.items()}
    if os.environ.get("FMR_BENCH_SERIES") and rank == 0:      # diagnostics: the stage kernels' launch-by-launch durations
        json.dump({k: [round(float(x), 5) for x in v] for k, v in region.items()}, open(os.environ["FMR_BENCH_SERIES"], "w"))
    if os.environ.get("FMR_FE_STAMPS") and rank == 0:          # diagnostics: where the last timed front-end launch spent its time, workgroup by workgroup
        raw = ch.debug_read(5)
        seq = int(raw[-1])
        NC = 32                                                 # calls in the ring of stream stamps (fmr_chain::kStampCalls)
        ring = raw[-(2 * NC * 16 + 1):-(NC * 16 + 1)].astype(np.int64).reshape(NC, 16)      # constant 100 MHz clock
        ring_khz = raw[-(NC * 16 + 1):-1].astype(np.int64).reshape(NC, 16)                   # shader clock [kHz] measured by the stamp in front of a front end
        before, after = int(ring[seq % NC, 0]), int(ring[seq % NC, 1])
        stp = raw[:-(2 * NC * 16 + 1)].reshape(-1, 3)
        # the timed region step by step: front-end start to start, the stages, and the shader clock in front of the front end
        ser = []
        for c in range(seq - min(args.steps, NC - 2) + 1, seq + 1):
            r0, r1 = ring[(c - 1) % NC], ring[c % NC]
            if not (r0[0] and r1[0]):
                continue
            ser.append("%d: %.0f us (fe %.0f, pll stage %.0f) %.0f MHz" % (c - (seq - args.steps), (r1[0] - r0[0]) * 0.01, (r0[1] - r0[0]) * 0.01,
                                                                           (r0[7] - r0[1]) * 0.01 if r0[7] else -1, ring_khz[(c - 1) % NC][0] * 1e-3))
        print("[fe stamps] steps of the timed region, start to start of the front ends: " + "; ".join(ser), file=sys.stderr)
        t_s, t_e = stp[:, 0].astype(np.int64), stp[:, 1].astype(np.int64)
        rel_s, rel_e, dur = (t_s - t_s.min()) * 0.01, (t_e - t_s.min()) * 0.01, (t_e - t_s) * 0.01      # us (100 MHz clock)
        xcc = (stp[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
        pc = lambda v: [round(float(np.percentile(v, q)), 1) for q in (0, 10, 50, 90, 99, 100)]
        print("[fe stamps] workgroups %d  start offset us (min p10 p50 p90 p99 max) %s  duration %s  end %s" %
              (len(stp), pc(rel_s), pc(dur), pc(rel_e)), file=sys.stderr)
        print("[fe stamps] stream stamp in front of the launch -> first workgroup %.1f us; last workgroup's end -> stream stamp behind the launch %.1f us" %
              ((int(t_s.min()) - before) * 0.01, (after - int(t_e.max())) * 0.01), file=sys.stderr)
        names = ["fe_before", "fe_after", "deemph_decim", "aud_poly", "pilotcut", "dc_pass1", "fm_out", "pll", "stats", "if_agc", "pll_commit", "pll_finish", "pll_shoot_jac"]
        t0s = int(t_s.min())
        for back in (1, 0):      # the stream stamps of the call before (whose tail runs into this front end) and of this call, relative to this front end's first workgroup
            row = ring[(seq - back) % NC]
            print("[fe stamps] call %d ends of kernels relative to the first workgroup of the last front end (us): " % (seq - back) +
                  ", ".join("%s %.1f" % (nm, (int(row[i]) - t0s) * 0.01) for i, nm in enumerate(names) if row[i]), file=sys.stderr)
        late = np.argsort(rel_e)[-8:]
        print("[fe stamps] last to end: " + ", ".join("wg %d xcc %d start %.1f dur %.1f end %.1f" % (w, xcc[w], rel_s[w], dur[w], rel_e[w]) for w in late), file=sys.stderr)
        print("[fe stamps] per XCC mean duration: " + ", ".join("%d: %.1f (n %d)" % (x, dur[xcc == x].mean(), int((xcc == x).sum())) for x in sorted(set(xcc.tolist()))), file=sys.stderr)
    st = ch.status(0)
    # The host's own cost per call.  Inside the timed loop the host runs ahead of the GPU until the chain's eight table
    # slots are taken and is then paced by the GPU: over a long loop t_enq / steps converges to ms_per_step whatever the
    # host spends.  Four calls enqueued into an idle queue (fewer than the slots) are not paced by anything.
    ch.enable_kernel_timing(0)
    t1 = time.perf_counter()
    for _ in range(3):                        # (the pipelined chain keeps three calls in flight: a fourth would wait for the first)
        step()
    host_unpaced_ms = (time.perf_counter() - t1) / 3 * 1e3
    ch.synchronize()
    torch.cuda.synchronize()
    # what this box delivers to a kernel that only reads the same input buffer (context for the roofline fraction)
    try:
        box_read = fmr.probe_read_bandwidth(local_rank, iq.data_ptr(), iq.numel() * iq.element_size())
    except Exception as e:      # a measurement aid must not take the line down with it
        print(f"[bench] read-bandwidth probe failed: {e}", file=sys.stderr)
        box_read = 0.0
    # one extra, untimed step with every kernel instrumented: the per-kernel table.  The instrumented step
    # serialises nothing, but its event pairs span the overlap of the chain's three HIP streams: the entries
    # sum to more than ms_per_step.
    ch.enable_kernel_timing(1)
    step()
    ktot = {}
    for name, ms in ch.kernel_times():
        a = ktot.setdefault(name, [0.0, 0])
        a[0] += ms
        a[1] += 1
    ch.enable_kernel_timing(0)
    per_rank_ms = [round(dt / args.steps * 1e3, 4)]
    if world > 1:
        mine = torch.tensor([dt], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 4) for t in allt]
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        dt = float(mine.item())
    total_samples = world * S * n * args.steps
    value = total_samples / dt / 1e6
    if not am and not args.no_pilot:
        assert st.stereo_detected == 1, "PLL did not lock: the timed work is not the stereo path"
    assert int(alen.sum()) > 0 and bool(torch.isfinite(audio[0, :int(alen.sum())]).all())

    # ---- audio of THIS rank's stream 0 against the oracle: the chain's first call (cold start and lock included) and the
    # last blocks of the last timed step.  Every rank checks its own stream; rank 0 reports all of them.
    audio_check = {"stereo_locked": int(st.stereo_detected), "pilot_level": round(st.pilot_level, 6)}
    my_errs = [-1.0, -1.0]
    if audio_chk is not None:
        ifr, dec = _oracle_chain(args.mode, args.multipath_stages)
        ref = np.concatenate([dec.process(ifr.process(iq_chk[i:i + blk])) for i in range(0, len(iq_chk), blk)])
        assert len(ref) == len(audio_chk), (len(ref), len(audio_chk))
        err = float(np.sqrt(np.mean((audio_chk - ref) ** 2)))
        my_errs[0] = err
        audio_check.update({"audio_rms_err_vs_oracle": float("%.3e" % err), "audio_rms": float("%.4g" % np.sqrt(np.mean(ref ** 2))),
                            "blocks_checked": nchk, "audio_samples_checked": len(ref), "tolerance": 1e-5,
                            "what": "stream 0 of the rank, first call of this chain (cold start and lock included)"})
        assert err < 1e-5, f"rank {rank}: audio RMS error {err} vs oracle exceeds the north-star tolerance"
    if fmt == 0 and not am and not args.multipath_stages and B >= 20:
        iq0 = iq[0].cpu().numpy().view(np.complex64).reshape(-1)
        tv = verify_timed_step(iq0, audio_last, alen, B, blk, blocks_done)
        del iq0
        if tv is not None:
            audio_check["timed_step"] = tv
            my_errs[1] = tv["rms_err_vs_oracle"]
            assert tv["rms_err_vs_oracle"] < 1e-5, f"rank {rank}, timed step: audio RMS error {tv} vs oracle exceeds the north-star tolerance"
    if world > 1:
        mine = torch.tensor(my_errs, dtype=torch.float64, device=dev)
        alle = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(alle, mine)
        audio_check["per_rank"] = [{"rank": r, "first_call_rms_err": float("%.3e" % e[0].item()), "timed_step_rms_err": float("%.3e" % e[1].item())}
                                   for r, e in enumerate(alle)]
        audio_check["max_over_ranks"] = float("%.3e" % max(float(e.max().item()) for e in alle))

    if rank == 0:
        kavg = {k: v[0] / v[1] for k, v in ktot.items()}
        ravg = {k: float(np.mean(v)) for k, v in region.items()}          # averages over the K timed steps
        dom_name = "ifr_fused" if "ifr_fused" in (ravg or kavg) else "ifr_decim"
        dec_ms = ravg.get(dom_name, kavg.get(dom_name, 0.0))
        stage_src = ravg if ravg else kavg
        stage_ms = sum(stage_src.get(k, 0.0) for k in STAGE_KERNELS)
        bytes_per_launch = float(bps) * S * n     # algorithmic: 8 B per cf32 input IQ sample; 4 / 2 B for s16 / u8
        achieved = bytes_per_launch / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        stage_achieved = bytes_per_launch / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
        if am:
            workload = "configs[2]: AM 384 kS/s complex-float IQ in HBM, IfResampler(48 k) + AmDecoder narrow filter -> f64 audio"
        elif args.no_pilot:
            workload = "FM stereo decoder on a mono station (no pilot): unlocked steady state, serial PLL"
        elif args.multipath_stages:
            workload = f"configs[3]: as configs[1] with the MultipathFilter equaliser -E {args.multipath_stages}"
        elif args.if_filter:
            workload = "as configs[1] with the IF filter on (main.cpp -f medium): resampler -> 127-tap FIR -> discriminator"
        elif R8B:
            workload = ("as configs[1] with the IF resampler in the R8B class (r8b::CDSPResampler24 defaults: 0.98 x Nyquist, "
                        "stop band from Nyquist, 180 dB) -- the reference-equivalent filter, stage B 3122 taps per phase on the fp16 matrix cores (three-product split)")
        elif S > 1:
            workload = (f"configs[4] shard: {S} independent FM stereo streams per GPU, 10 MS/s complex-float IQ in HBM, "
                        "PilotPhaseLock on, IfResampler+FmDecoder -> f64 stereo 48 kHz")
        else:
            workload = ("configs[1]: single FM stereo stream per GPU, 10 MS/s complex-float IQ in HBM, "
                        "PilotPhaseLock on, IfResampler+FmDecoder -> f64 stereo 48 kHz")
        if not am and args.sigma != 1e-3:
            workload += " -- noise sigma %g per component (C/N %.1f dB)" % (args.sigma, 10 * np.log10(0.09 / (2 * args.sigma ** 2)))
        pmc_traffic = committed_pmc_traffic(dom_name, B, S, args) if not R8B else (None, None)
        mfma_roofline = None
        if R8B and stage_src.get("ifr_poly", 0) > 0:
            info = ch.resampler_info()
            n_if = float(S) * n * info["LB"] / (info["MB"] * info["D"])                 # IF samples per launch
            flops = 2.0 * 2.0 * info["TB"] * n_if                                        # taps x (re, im) x multiply-add
            tf = flops / (stage_src["ifr_poly"] * 1e-3) / 1e12
            if True:
                # fp16 matrix cores, both operands split in two fp16 terms: THREE products (hh + hl + lh) per algorithmic one
                mfma_roofline = {"bound": "mfma", "kernel": "ifr_poly (k_ifr_poly5h: stage B, %d taps per phase, v_mfma_f32_16x16x32_f16, "
                                                             "two-term fp16 split of both operands)" % info["TB"],
                                 "achieved": round(3 * tf, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(3 * tf / 2500.0, 4), "traffic": None,
                                 "avg_launch_ms": round(stage_src["ifr_poly"], 5), "algorithmic_flops_per_launch": flops,
                                 "issued_flops_per_launch": 3 * flops, "algorithmic_tflops": round(tf, 2),
                                 "peak_source": "MI355X_MICROARCH.md: ~2.5 PFLOP/s dense fp16 / bf16 MFMA; `achieved` counts the three products "
                                                "issued per algorithmic one (fp32-class accuracy needs them), `algorithmic_tflops` the filter's own"}
        out = {
            "metric": ("IQ MS/s (AM, 384 kS/s in), whole job" if am else "IQ MS/s (FM stereo, 10 MS/s in), whole job"),
            "value": round(value, 3), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "ranks_seen": world, "per_rank_ms_per_step": per_rank_ms, "host_thread_pinning": pinned,
            "vs_baseline": None, "dtype": "f32 front end (stage A as a two-term fp16 split on the matrix cores, fp32 accumulate) / f64 after the discriminator", "data": "synthetic",
            "config": {"workload": workload,
                       "input_format": args.input_format, "streams_per_gpu": S, "blocks_per_step": B, "block_len": blk,
                       "samples_per_step_per_gpu": S * n, "per_gpu_msps": round(value / world, 3),
                       "resampler": ch.resampler_info()},
            "roofline": {"bound": "hbm", "kernel": f"{dom_name} (reads every IQ sample)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic[0], "traffic_source": pmc_traffic[1],
                         "avg_launch_ms": round(dec_ms, 5), "launches_timed": (len(region.get(dom_name, [])) or None),
                         "timed_with": ("HIP events inside the timed region, on the stream the kernel runs on: " +
                                        ("the start / stop events of the kernel's own dispatch (hipExtLaunchKernelGGL), every timed step"
                                         if dom_name == "ifr_fused" and not os.environ.get("FMR_EVT_MARKERS") else
                                         "an event marker before and after the launch")),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "box_streaming_read": None if box_read <= 0 else {"GB/s": round(box_read, 1), "frac_of_peak": round(box_read / HBM_PEAK_GBS, 4),
                                                "kernel_vs_box": round(achieved / box_read, 4) if box_read > 0 else None,
                                                "what": "a plain read-only kernel (16-byte loads, 8 workgroups per CU) over the same "
                                                        "input buffer on this box, best of 5: the roofline's peak is the data sheet's"},
                         "stage": {"what": "FIR + discriminator stage (north star): sum of the average launch durations of "
                                           + " + ".join(k for k in STAGE_KERNELS if k in stage_src),
                                   "ms": round(stage_ms, 5), "achieved": round(stage_achieved, 2),
                                   "frac": round(stage_achieved / HBM_PEAK_GBS, 4),
                                   "kernels_ms": {k: round(stage_src[k], 5) for k in STAGE_KERNELS if k in stage_src}}},
            **({"roofline_mfma": mfma_roofline} if mfma_roofline else {}),
            "kernel_ms_per_step": {k: round(v, 5) for k, v in kavg.items()},
            "kernel_ms_note": "from one extra instrumented step; kernels on the chain's three HIP streams overlap, so the entries sum to more than ms_per_step",
            "host_enqueue_ms_per_step": round(host_unpaced_ms, 4),
            "host_enqueue_note": "host time per call for three calls enqueued into an idle queue (the host's own cost); in the "
                                 "timed loop the host is paced by the GPU once it is three calls ahead: " +
                                 "%.4f ms per step there" % (t_enq / args.steps * 1e3),
            "cold_first_call_ms": round(cold_ms, 2),
            "clock": {"spinup_ms": args.spinup_ms, "spinup_steps": spin_steps,
                      "shader_mhz": {"before_spinup": round(mhz[0]), "before_timed_region": round(mhz[1]), "after_timed_region": round(mhz[2])},
                      "note": "untimed ordinary steps in front of the W warm-up steps: the shader clock ramps over tens of ms of load after "
                              "an idle gap, the decoder's recurrence kernels follow it (--spinup-ms 0: the region as rounds 1-5 timed it)"},
            "audio_check": audio_check,
            "recurrences": {"agc_newton_rounds": st.agc_iterations, "pll_newton_rounds": st.pll_iterations,
                            "pll_residuals": [float("%.3g" % v) for v in st.pll_residual_history[:st.pll_iterations]],
                            "pll_mismatches": [float("%.3g" % v) for v in st.pll_mismatch_history[:st.pll_iterations]],
                            "agc_residuals": [float("%.3g" % v) for v in st.agc_residual_history[:min(st.agc_iterations, 16)]],
                            "agc_serial_fallback": st.agc_fallback, "pll_serial_fallback": st.pll_fallback,
                            "af_tail_serial_fallback": st.af_agc_fallback},
        }
    ch.close()
    headline = (not am and fmt == 0 and S == 1 and not R8B and not IF_FILTER and not args.multipath_stages and not args.no_pilot
                and args.sigma == 1e-3)
    if rank == 0 and world == 1 and headline and not args.no_r8b_leg:
        out["r8b"] = r8b_leg(fmr, iq, audio, n, blk, B, local_rank, spinup_ms=args.spinup_ms)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # The CPU legs run on rank 0 once the GPU part of every rank is over (the other ranks have left: the host's
        # cores are free), on all the cores of the box again (the rank was pinned to its slice for the timed region).
        if not args.no_cpu_baseline:
            if hasattr(os, "sched_setaffinity"):
                try:
                    os.sched_setaffinity(0, range(os.cpu_count() or 1))
                except OSError:
                    pass
            out["cpu_baseline"] = cpu_baseline(args.mode, args.multipath_stages)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if args.all_configs and world == 1:
        run_other_configs(1)


def r8b_leg(fmr, iq, audio, n, blk, B, device, steps=40, warmup=3, spinup_ms=60.0):
    """The same workload through the REFERENCE-EQUIVALENT resampler class (r8b::CDSPResampler24's default specification:
    0.98 x Nyquist, stop band from Nyquist, 180 dB -- IfResampler.cpp:25-29), timed in a second, short region of the same
    process and put into the headline line as `r8b`: the headline's FAST class is this project's own filter design, this is
    what the reference builds.  Same input buffer, same step, audio of the first call against the r8brain-class oracle."""
    import torch
    global R8B
    ch = fmr.Chain(mode=fmr.MODE_FM, input_rate=FS, enable_resampler=True, stereo=True, n_streams=1, max_block_len=blk, max_blocks=B,
                   device=device, resampler_class=fmr.RESAMPLER_R8B)
    block_len = [blk] * B

    def step():
        return ch.process_blocks_device(iq.data_ptr(), n, block_len, audio.data_ptr(), audio.shape[1], sync=False)

    alen0 = step()
    ch.synchronize()
    nchk = min(B, 40)
    got = audio[0, :int(alen0[:nchk].sum())].cpu().numpy().copy()
    spin_steps, t_sp = 0, time.perf_counter()        # (the shader clock's ramp after the idle gap of the headline's oracle checks: see main)
    while (time.perf_counter() - t_sp) * 1e3 < spinup_ms:
        step()
        spin_steps += 1
    for _ in range(warmup):
        step()
    ch.synchronize()
    ch.enable_kernel_timing(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ch.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    region = {}
    for name, ms in ch.kernel_times():
        region.setdefault(name, []).append(ms)
    st = ch.status(0)
    ravg = {k: float(np.mean(v)) for k, v in region.items()}
    stage_ms = sum(ravg.get(k, 0.0) for k in STAGE_KERNELS)
    info = ch.resampler_info()
    ch.close()
    was, R8B = R8B, True
    try:
        ifr, dec = _oracle_chain("fm", 0)
    finally:
        R8B = was
    x = iq[0, :nchk * blk].cpu().numpy().view(np.complex64).reshape(-1)
    ref = np.concatenate([dec.process(ifr.process(x[i:i + blk])) for i in range(0, len(x), blk)])
    assert len(ref) == len(got), (len(ref), len(got))
    err = float(np.sqrt(np.mean((got - ref) ** 2)))
    assert err < 1e-5, f"R8B leg: audio RMS error {err} vs the r8brain-class oracle exceeds the north-star tolerance"
    bytes_per_launch = 8.0 * n
    return {"what": "the same step through the reference-equivalent resampler class (r8b::CDSPResampler24 defaults, IfResampler.cpp:25-29): "
                    "second region of this process, %d steps after %d spin-up and %d warm-up steps" % (steps, spin_steps, warmup),
            "value": round(n * steps / dt / 1e6, 3), "unit": "MS/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "resampler": info,
            "stage": {"ms": round(stage_ms, 5), "frac": round(bytes_per_launch / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if stage_ms > 0 else None,
                      "kernels_ms": {k: round(ravg[k], 5) for k in STAGE_KERNELS if k in ravg}},
            "pll_newton_rounds": st.pll_iterations, "pll_serial_fallback": st.pll_fallback,
            "audio_check": {"audio_rms_err_vs_r8brain_class_oracle": float("%.3e" % err), "blocks_checked": nchk,
                            "audio_samples_checked": len(ref), "tolerance": 1e-5}}


# the other configurations of BASELINE.json (and the reference-equivalent resampler class), one line each, same run, same
# box, never the headline
OTHER_CONFIGS = [["--mode", "am", "--steps", "20", "--warmup", "3"],                                            # configs[2]
                 ["--multipath-stages", "64", "--blocks", "64", "--steps", "5", "--warmup", "3"],               # configs[3]
                 ["--streams", "32", "--blocks", "128", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],  # configs[4] shard
                 ["--resampler-class", "r8b", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],            # r8b::CDSPResampler24's filter
                 ["--no-pilot", "--blocks", "256", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],        # mono station
                 ["--sigma", "1e-2", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"],                     # 26.5 dB C/N: what the PLL's iteration needs there
                 ["--sigma", "3e-2", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"]]                     # 17 dB C/N


def run_other_configs(n_gpus):
    import subprocess
    for fl in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__)] + (["--gpus", str(n_gpus)] if n_gpus > 1 else []) + fl
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(line[-1] if line else json.dumps({"config": {"workload": " ".join(fl)}, "error": r.stderr[-400:]}), flush=True)


def block_api(args, ch, iq, blk, fs, rank, world, am):
    """The drop-in call (FmDecode.h:74 / main.cpp:956): one block per fmr_process() through HOST buffers -- H2D copy,
    the whole launch set, D2H copy, synchronise, every call.  PCIe-inclusive; reported as its own line."""
    nb = iq.shape[1] // blk                      # whole buffer: its tones are periodic over exactly this length
    K = max(1, args.api_batch)
    x = iq[0, :nb * blk].cpu().numpy().view(np.complex64).reshape(-1)
    blocks = [np.ascontiguousarray(x[i * blk:(i + 1) * blk]) for i in range(nb)]
    nwarm = -(-120 // K) * K                     # cold start + lock; a multiple of K keeps the replay seamless
    assert nb % K == 0 and nwarm < nb
    for b in blocks[:nwarm]:
        ch.process(b) if K == 1 else ch.process_blocks(b[None, :], [blk])
    lat = []
    t0 = time.perf_counter()
    k = nwarm
    for _ in range(args.steps):
        if k + K > nb:
            k = 0                                # the buffer is periodic (tones snapped to its length): no seam
        t1 = time.perf_counter()
        if K == 1:
            ch.process(blocks[k])
        else:                                    # K blocks the caller handed over one by one, decoded in one call
            ch.process_blocks(x[None, k * blk:(k + K) * blk], [blk] * K)
        k += K
        lat.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    lat = np.array(lat) * 1e6
    out = {"metric": "IQ MS/s through the host-buffer call (PCIe-inclusive), %d block(s) per call" % K, "value": round(args.steps * K * blk / dt / 1e6, 3),
           "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": nwarm, "ms_per_step": round(dt / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 front end / f64 after the discriminator",
           "data": "synthetic",
           "config": {"workload": f"{K} x {blk}-sample block(s) per call through host buffers, {'AM' if am else 'FM stereo'}",
                      "block_len": blk, "blocks_per_call": K, "streams_per_gpu": 1},
           "latency_us": {"p50": round(float(np.percentile(lat, 50)), 1), "p90": round(float(np.percentile(lat, 90)), 1),
                          "p99": round(float(np.percentile(lat, 99)), 1), "min": round(float(lat.min()), 1)},
           "roofline": None,
           "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(args.mode, args.multipath_stages, 4.0)}
    if rank == 0:
        print(json.dumps(out))
    ch.close()


def dry_run(args, rank, world, S, B, blk, pinned=None):
    """Launch-contract check without a GPU (gloo): same rank/barrier/max-over-ranks/aggregate code shape as main(); the
    per-rank step is the CPU oracle on a tiny sample.  The line is marked as a dry run and is never a measurement."""
    import torch
    import torch.distributed as dist
    import siggen
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    nb = min(B, 2)
    xs = [siggen.fm_stereo_iq(nb * blk, FS, stream_id=rank * S + s) for s in range(S)]
    chains = [_oracle_chain("fm") for _ in range(S)]

    def step():
        tot = 0
        for x, (ifr, dec) in zip(xs, chains):
            for i in range(0, len(x), blk):
                tot += len(dec.process(ifr.process(x[i:i + blk])))
        return tot

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = [round(dt / args.steps * 1e3, 4)]
    if world > 1:
        mine = torch.tensor([dt], dtype=torch.float64)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 4) for t in allt]
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        dt = float(mine.item())
    total = world * S * nb * blk * args.steps
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # as in main(): the CPU legs on rank 0 once every rank's timed part is over, whatever the world size (bounded here)
        cpu = None if args.no_cpu_baseline else cpu_baseline("fm", 0, seconds=0.5, max_procs=2)
        print(json.dumps({"metric": "IQ MS/s (FM stereo, 10 MS/s in), whole job", "value": round(total / dt / 1e6, 3), "unit": "MS/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                          "ranks_seen": world, "per_rank_ms_per_step": per_rank_ms, "host_thread_pinning": pinned,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "cpu oracle", "data": "DRY RUN (CPU oracle, gloo) -- not a measurement",
                          "config": {"workload": "dry run of the launch contract", "streams_per_gpu": S, "blocks_per_step": nb,
                                     "samples_per_step_per_gpu": S * nb * blk},
                          "roofline": None, "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()
