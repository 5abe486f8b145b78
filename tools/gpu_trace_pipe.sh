#!/bin/bash
# kernel trace of the pipelined chain:  tools/gpu_trace_pipe.sh <tag> <blocks> "ENV=.. ENV=.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tag=$1; blocks=${2:-8192}; cfg=$3
d=gpurun_out/trace_$tag
rm -rf $d; mkdir -p $d
env $cfg timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python bench.py --blocks $blocks --steps 12 --warmup 3 --no-cpu-baseline > $d/bench.json 2> $d/bench.err
f=$(find $d -name '*kernel_trace.csv' | head -1)
python tools/trace_window.py $f 6 2 > $d/timeline.txt
cut -c1-160 $d/bench.json | tail -1
find $d -name '*.csv' -size +20M -delete
