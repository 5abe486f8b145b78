#!/bin/bash
root=$(pwd); cd /tmp && export TMPDIR=/tmp; cd "$root"
mkdir -p gpurun_out/c34
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/c34/tr -o t -- python bench.py --api-mode block --steps 40 --blocks 400 --no-cpu-baseline > gpurun_out/c34/bench.log 2>&1
FMR_HOST_PROF=1 python bench.py --api-mode block --steps 40 --blocks 400 --no-cpu-baseline > gpurun_out/c34/bench_hp.log 2>&1
tail -2 gpurun_out/c34/bench.log | cut -c1-300
grep -i "host" gpurun_out/c34/bench_hp.log | tail -5
