#!/bin/bash
# A/B of two prebuilt libraries (tools/tmp_<name>.so), interleaved:  tools/gpu_ab_lib.sh k3 k4 [rounds]
mkdir -p gpurun_out/ab
for r in $(seq 1 ${3:-3}); do
  for v in $1 $2; do
    cp tools/tmp_$v.so airspy-fmradion_amd/libfmradion_amd.so
    timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err
    python - $v <<'PY'
import json,sys
v=sys.argv[1]
b=json.loads([l for l in open(f'gpurun_out/ab/{v}.json') if l.startswith('{')][-1]); k=b['kernel_ms_per_step']
print(v, b['value'], b['ms_per_step'], 'fused', k.get('ifr_fused'), 'pll', k.get('pll'))
PY
  done
done
