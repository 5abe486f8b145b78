#!/bin/bash
mkdir -p gpurun_out/c20
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_stream_loop.py -m gpu -x -q > gpurun_out/c20/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c20/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c20/bench.json 2> gpurun_out/c20/bench.err
tail -3 gpurun_out/c20/tests.log
