#!/bin/bash
# kernel trace of a step four times the default size: the GPU step (3 ms) is then far longer than the host's traced
# enqueue (1 ms), so the gaps in the timeline are the GPU's own (launch dependencies), not the tracer's
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/big; mkdir -p gpurun_out/big
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/big -o big -- python bench.py --blocks ${1:-8192} --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/big/bench.json 2> gpurun_out/big/bench.err
f=$(find gpurun_out/big -name '*kernel_trace.csv' | head -1)
python tools/trace_timeline.py $f 3 > gpurun_out/big/timeline.txt
cut -c1-300 gpurun_out/big/bench.json | tail -1
find gpurun_out/big -name '*.csv' -size +20M -delete
