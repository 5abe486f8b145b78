"""World-size-2 check of the multi-GPU launch contract on CPU (gloo): streams shard one per
rank with no data-path collective; the only collectives are the start/stop barrier and the
max-over-ranks time, exactly as bench.py does on N GPUs.  The per-rank work here is the CPU
oracle (the HIP path needs a GPU); what is tested is the sharding/aggregation logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_py as ora
import siggen
from conftest import load_filter


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blk, nblk = 65536, 4
    x = siggen.fm_stereo_iq(nblk * blk, 10e6, stream_id=rank)      # stream s -> rank s, nothing exchanged
    ifr = ora.IfResampler(10e6, 384e3)
    fm = ora.FmDecoder(False, np.array([0, 1, 0], dtype=np.float32), True, 50.0, False, 0,
                       load_filter("jj1bdx_48khz_fmaudio"))
    dist.barrier()
    n_audio = sum(len(fm.process(ifr.process(b))) for b in siggen.blocks(x, blk))
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)     # stand-in for the per-rank wall time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    total = torch.tensor([float(nblk * blk)], dtype=torch.float64)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    out[rank] = (n_audio, float(t.item()), float(total.item()), float(fm.get_if_rms()))
    dist.destroy_process_group()


def test_two_ranks_shard_streams_without_exchange():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    n0, t0, tot0, rms0 = out[0]
    n1, t1, tot1, rms1 = out[1]
    assert n0 == n1 > 0                       # identical block lengths -> identical counts on every rank
    assert t0 == t1 == 2.0                    # max over ranks
    assert tot0 == tot1 == 2 * 4 * 65536      # whole-job sample count = sum over ranks (weak scaling)
    assert rms0 == pytest.approx(rms1, rel=1e-3)


def test_bench_launch_contract_two_ranks_gloo():
    """bench.py under the driver's own launcher (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py
    --gpus 2`) with the --cpu-dry-run hook: RANK / WORLD_SIZE / MASTER_* from the env, barrier on both sides of the
    timed region, max over ranks, ONE JSON line from rank 0 with the whole-job aggregate.  (The HIP step itself needs
    a GPU; N > 1 GPUs are only available to the driver.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--streams", "2", "--cpu-dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    # whole-job aggregate: 2 ranks x 2 streams x 2 blocks x 65536 samples x 2 steps
    assert out["config"]["samples_per_step_per_gpu"] == 2 * 2 * 65536
    assert out["value"] == pytest.approx(2 * 2 * 2 * 65536 * 2 / (out["ms_per_step"] * 2 * 1e-3) / 1e6, rel=1e-2)
    assert "DRY RUN" in out["data"]
    # the CPU baseline is part of an N > 1 line too (rank 0, after the timed region)
    assert out["cpu_baseline"] is not None and out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] == 1
    assert out["cpu_baseline"]["all_cores"]["cores"] >= 1


def test_bench_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` with NO launcher in the environment must start two ranks itself (round 2: it silently ran
    one) and say how many ranks it saw."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "2",
           "--cpu-dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and len(out["per_rank_ms_per_step"]) == 2
    assert out["config"]["samples_per_step_per_gpu"] == 2 * 2 * 65536
    assert out["ms_per_step"] == pytest.approx(max(out["per_rank_ms_per_step"]), rel=1e-3)


def test_bench_config5_shape_eight_ranks_gloo():
    """BASELINE.json configs[4] as the driver will launch it on the 8-GPU node -- `bench.py --gpus 8 --streams 32`: 256
    independent streams, 32 per rank -- through the self-launcher on CPU (gloo, --cpu-dry-run): eight ranks are started
    and seen, every one contributes its 32 streams to the whole-job aggregate, and the ranks take disjoint slices of the
    host's cores (the GPU path pins by NUMA node; the slices are the fallback this box can exercise)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--streams", "32",
           "--blocks", "1", "--cpu-dry-run", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and len(out["per_rank_ms_per_step"]) == 8
    assert out["scaling"] == "weak" and out["config"]["streams_per_gpu"] == 32
    assert out["config"]["samples_per_step_per_gpu"] == 32 * 1 * 65536                    # 32 streams x 1 block per rank
    assert out["value"] == pytest.approx(8 * 32 * 65536 / (out["ms_per_step"] * 1e-3) / 1e6, rel=1e-2)   # 256 streams in the aggregate
    assert out["host_thread_pinning"] in ("slice", None)
