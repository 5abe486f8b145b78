#!/bin/bash
# interleaved A/B of prebuilt libraries on bench.py with extra arguments:  tools/gpu_ab_args.sh <rounds> "<bench args>" <name> <name> ...
mkdir -p gpurun_out/ab
rounds=$1; shift; args=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/tmp_$v.so airspy-fmradion_amd/libfmradion_amd.so
    timeout 200 python bench.py $args --no-cpu-baseline --no-r8b-leg > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err < /dev/null
    timeout 20 python - $v <<'PY'
import json,sys
v=sys.argv[1]
b=json.loads([l for l in open(f'gpurun_out/ab/{v}.json') if l.startswith('{')][-1]); k=b['kernel_ms_per_step']
print(v, b['ms_per_step'], 'fused', b['roofline']['avg_launch_ms'], 'pll', k.get('pll'), 'agc', k.get('if_agc'), 'rounds', b['recurrences']['pll_newton_rounds'], b['recurrences']['agc_newton_rounds'])
PY
  done
done
