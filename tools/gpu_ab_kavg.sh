#!/bin/bash
# interleaved A/B of prebuilt libraries, per-kernel averages from the chain's trace:  tools/gpu_ab_kavg.sh "<kernel_avg args>" <rounds> <name> ...
args=$1; rounds=$2; shift; shift
cp airspy-fmradion_amd/libfmradion_amd.so /tmp/keep.so
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/tmp_$v.so airspy-fmradion_amd/libfmradion_amd.so
    timeout 200 python tools/kernel_avg.py $args --tag $v < /dev/null 2>&1 | grep -v amdgpu.ids | tail -2
  done
done
cp /tmp/keep.so airspy-fmradion_amd/libfmradion_amd.so
