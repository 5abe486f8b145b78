#!/bin/bash
mkdir -p gpurun_out/c10
timeout 120 tools/bench_ldsread.bin 2>&1 | grep "stride -1" > gpurun_out/c10/ldsread.log
timeout 120 tools/bench_acc64.bin > gpurun_out/c10/acc64.log 2>&1
timeout 300 tools/bench_fused.bin > gpurun_out/c10/fused.log 2>&1
echo "fused rc=$?" >> gpurun_out/c10/fused.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/c10/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c10/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c10/bench.json 2> gpurun_out/c10/bench.err
tail -3 gpurun_out/c10/tests.log; cat gpurun_out/c10/ldsread.log gpurun_out/c10/acc64.log; grep -v "^call\|^mid tail" gpurun_out/c10/fused.log | head -32; cut -c1-300 gpurun_out/c10/bench.json
