#!/bin/bash
mkdir -p gpurun_out/c15
timeout 300 tools/bench_fused.bin > gpurun_out/c15/fused.log 2>&1
echo "fused rc=$?" >> gpurun_out/c15/fused.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/c15/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c15/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c15/bench.json 2> gpurun_out/c15/bench.err
tail -3 gpurun_out/c15/tests.log; grep -v "^call\|^mid tail" gpurun_out/c15/fused.log | head -40; cut -c1-300 gpurun_out/c15/bench.json
