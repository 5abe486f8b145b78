#!/bin/bash
# Runs every prebuilt harness variant tools/tmp/fused_<name>.bin on the GPU box, twice (box drift check).
#   gpurun -- 'bash tools/gpu_fused_variants.sh <outdir-tag>'
tag=${1:-fv}
out=gpurun_out/$tag
mkdir -p $out
for r in 1 2; do
for b in tools/tmp/fused_*.bin; do
  n=$(basename $b .bin)
  timeout 180 $b > $out/${n}_$r.log 2>&1
  echo "$n run $r rc=$?" >> $out/rc.log
done
done
grep -H -E "CORRECTNESS|CHAIN|lean|fused A\+B, 256|DMA only\)|no DMA \(A" $out/fused_*.log | sed 's/  */ /g'
