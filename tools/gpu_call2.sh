#!/bin/bash
mkdir -p gpurun_out/c2
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c2/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c2/tests.log
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/c2/bench_c2.json 2> gpurun_out/c2/bench_c2.err
FMR_NO_FUSED=1 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/c2/bench_c2_nofused.json 2> gpurun_out/c2/bench_c2_nofused.err
timeout 300 python bench.py --streams 32 --blocks 128 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/c2/bench_c5.json 2> gpurun_out/c2/bench_c5.err
tail -5 gpurun_out/c2/tests.log
