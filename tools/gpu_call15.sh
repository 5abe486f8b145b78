#!/bin/bash
root=$(pwd); cd /tmp && export TMPDIR=/tmp; cd "$root"
mkdir -p gpurun_out/c21
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c21/tr -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c21/bench.log 2>&1
find gpurun_out/c21 -name "*kernel_trace.csv" | head
