#!/bin/bash
# full GPU suite + default bench (+ the one-block host API line)
mkdir -p gpurun_out/suite
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/suite/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/suite/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/suite/bench.json 2> gpurun_out/suite/bench.err
timeout 300 python bench.py --api-mode block --steps 300 --blocks 400 --no-cpu-baseline > gpurun_out/suite/bench_block1.json 2> gpurun_out/suite/bench_block1.err
tail -4 gpurun_out/suite/tests.log; cut -c1-200 gpurun_out/suite/bench.json; cut -c1-200 gpurun_out/suite/bench_block1.json
