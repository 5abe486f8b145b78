#!/bin/bash
# the lines collect_profiles.sh does not take: host-buffer API (1 / 16 / 64 blocks per call) and the 4x-size kernel trace
mkdir -p gpurun_out/extra
timeout 300 python bench.py --api-mode block --steps 300 --blocks 400 --no-cpu-baseline > gpurun_out/extra/host_api_1.json 2> gpurun_out/extra/host_api_1.err
timeout 300 python bench.py --api-mode block --api-batch 16 --steps 300 --blocks 400 --no-cpu-baseline > gpurun_out/extra/host_api_16.json 2> gpurun_out/extra/host_api_16.err
timeout 300 python bench.py --api-mode block --api-batch 64 --steps 300 --blocks 448 --no-cpu-baseline > gpurun_out/extra/host_api_64.json 2> gpurun_out/extra/host_api_64.err
for f in 1 16 64; do cut -c1-160 gpurun_out/extra/host_api_$f.json | tail -1; done
bash tools/gpu_trace_big.sh 8192
