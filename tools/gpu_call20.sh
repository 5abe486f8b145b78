#!/bin/bash
mkdir -p gpurun_out/c30
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "dropout" > gpurun_out/c30/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c30/tests.log
tail -40 gpurun_out/c30/tests.log
