#!/bin/bash
mkdir -p gpurun_out/c6
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "multipath" > gpurun_out/c6/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c6/tests.log
timeout 300 python bench.py --multipath-stages 64 --blocks 64 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/c6/bench_c4.json 2> gpurun_out/c6/bench_c4.err
FMR_MPF_V1=1 timeout 300 python bench.py --multipath-stages 64 --blocks 64 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/c6/bench_c4_v1.json 2> gpurun_out/c6/bench_c4_v1.err
timeout 300 python bench.py --multipath-stages 64 --streams 32 --blocks 64 --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/c6/bench_c4_s32.json 2> gpurun_out/c6/bench_c4_s32.err
tail -5 gpurun_out/c6/tests.log
