#!/bin/bash
mkdir -p gpurun_out/c4
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c4/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c4/tests.log
timeout 300 python bench.py --steps 100 --warmup 10 > gpurun_out/c4/bench_c2.json 2> gpurun_out/c4/bench_c2.err
timeout 300 python bench.py --streams 32 --blocks 128 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/c4/bench_c5.json 2> gpurun_out/c4/bench_c5.err
tail -4 gpurun_out/c4/tests.log
