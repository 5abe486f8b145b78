#!/bin/bash
mkdir -p gpurun_out/c3
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "config2 or config5 or random or cold or hard or multi_stream" > gpurun_out/c3/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c3/tests.log
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/c3/bench_c2.json 2> gpurun_out/c3/bench_c2.err
FMR_NO_FUSED=1 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/c3/bench_c2_nofused.json 2> gpurun_out/c3/bench_c2_nofused.err
tail -4 gpurun_out/c3/tests.log
