#!/bin/bash
# run one pytest selection on the GPU box:  tools/gpu_one.sh "<pytest args>"
mkdir -p gpurun_out/one
timeout 1200 python -m pytest $1 -m gpu -x -q > gpurun_out/one/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/one/tests.log
tail -40 gpurun_out/one/tests.log
