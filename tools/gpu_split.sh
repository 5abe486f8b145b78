#!/bin/bash
# GPU suite, then the default bench under the ablation switches given as arguments
mkdir -p gpurun_out/sp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/sp/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/sp/tests.log
tail -4 gpurun_out/sp/tests.log
bash tools/gpu_try.sh "$@"
