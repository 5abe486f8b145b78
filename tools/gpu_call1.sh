#!/bin/bash
# round-2 call 1: configs 3/4 parity + baseline bench lines of every config
mkdir -p gpurun_out/c1
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q > gpurun_out/c1/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c1/tests.log
timeout 300 python bench.py --steps 50 --warmup 5 > gpurun_out/c1/bench_c2.json 2> gpurun_out/c1/bench_c2.err
timeout 300 python bench.py --streams 32 --blocks 128 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/c1/bench_c5.json 2> gpurun_out/c1/bench_c5.err
timeout 300 python bench.py --multipath-stages 64 --blocks 64 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/c1/bench_c4.json 2> gpurun_out/c1/bench_c4.err
timeout 300 python bench.py --multipath-stages 64 --streams 32 --blocks 64 --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/c1/bench_c4_s32.json 2> gpurun_out/c1/bench_c4_s32.err
timeout 300 python bench.py --mode am --steps 6 --warmup 2 > gpurun_out/c1/bench_c3.json 2> gpurun_out/c1/bench_c3.err
timeout 300 python bench.py --mode am --streams 32 --blocks 1024 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/c1/bench_c3_s32.json 2> gpurun_out/c1/bench_c3_s32.err
tail -3 gpurun_out/c1/tests.log
