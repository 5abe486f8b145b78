#!/bin/bash
mkdir -p gpurun_out/c28
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c28/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c28/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c28/bench.json 2> gpurun_out/c28/bench.err
tail -4 gpurun_out/c28/tests.log; cut -c1-300 gpurun_out/c28/bench.json
