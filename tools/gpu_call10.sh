#!/bin/bash
mkdir -p gpurun_out/c13
timeout 300 tools/bench_fused.bin > gpurun_out/c13/fused.log 2>&1
echo "fused rc=$?" >> gpurun_out/c13/fused.log
grep -v "^call\|^mid tail" gpurun_out/c13/fused.log | head -40
