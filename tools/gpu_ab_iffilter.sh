#!/bin/bash
# A/B of prebuilt libraries (tools/tmp_<name>.so) on bench.py --if-filter:  tools/gpu_ab_iffilter.sh base lin lin5
mkdir -p gpurun_out/ab
cp airspy-fmradion_amd/libfmradion_amd.so /tmp/keep.so
for r in 1 2; do
  for v in "$@"; do
    cp tools/tmp_$v.so airspy-fmradion_amd/libfmradion_amd.so
    timeout 300 python bench.py --if-filter --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err
    python - $v <<'PY'
import json,sys
v=sys.argv[1]
b=json.loads([l for l in open(f'gpurun_out/ab/{v}.json') if l.startswith('{')][-1]); k=b['kernel_ms_per_step']
print(v, b['value'], b['ms_per_step'], 'fused', k.get('ifr_fused'), 'fm_block', k.get('fm_block'), 'stage', b['roofline']['stage']['kernels_ms'])
PY
  done
done
cp /tmp/keep.so airspy-fmradion_amd/libfmradion_amd.so
