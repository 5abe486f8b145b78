#!/bin/bash
mkdir -p gpurun_out/c35
timeout 900 python -m pytest tests/test_gpu_errors.py -m gpu -x -q > gpurun_out/c35/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c35/tests.log
tail -40 gpurun_out/c35/tests.log
