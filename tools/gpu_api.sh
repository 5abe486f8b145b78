#!/bin/bash
mkdir -p gpurun_out/api
timeout 300 python bench.py --api-mode block --api-batch 64 --steps 50 --blocks 448 --no-cpu-baseline > gpurun_out/api/b64.json 2> gpurun_out/api/b64.err
cut -c1-220 gpurun_out/api/b64.json; tail -2 gpurun_out/api/b64.err
