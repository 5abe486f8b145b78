#!/bin/bash
mkdir -p gpurun_out/c7
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c7/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c7/tests.log
timeout 300 python bench.py --api-mode block --steps 300 --blocks 400 --no-cpu-baseline > gpurun_out/c7/bench_block1.json 2> gpurun_out/c7/bench_block1.err
timeout 300 python bench.py --api-mode block --api-batch 16 --steps 100 --blocks 400 --no-cpu-baseline > gpurun_out/c7/bench_block16.json 2> gpurun_out/c7/bench_block16.err
timeout 300 python bench.py --api-mode block --api-batch 64 --steps 50 --blocks 400 --no-cpu-baseline > gpurun_out/c7/bench_block64.json 2> gpurun_out/c7/bench_block64.err
tail -4 gpurun_out/c7/tests.log
