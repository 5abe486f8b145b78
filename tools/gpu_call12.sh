#!/bin/bash
mkdir -p gpurun_out/c16
timeout 900 python -m pytest tests/test_gpu_ppm.py -m gpu -x -q > gpurun_out/c16/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c16/tests.log
timeout 120 tools/bench_acc64.bin > gpurun_out/c16/acc64.log 2>&1
tail -30 gpurun_out/c16/tests.log
