#!/bin/bash
# fixed cost of a timed region (pipeline fill + drain) against the per-step cost: total = a + b K
mkdir -p gpurun_out/fx
for rep in 1 2; do
for k in 20 40 100 200; do
  timeout 100 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-r8b-leg > gpurun_out/fx/k$k.json 2> gpurun_out/fx/k$k.err < /dev/null
  timeout 20 python - $k <<'PY'
import json,sys
k=int(sys.argv[1])
j=json.loads([l for l in open(f'gpurun_out/fx/k{k}.json') if l.startswith('{')][-1])
print('steps', k, 'ms_per_step', j['ms_per_step'], 'total_ms', round(j['ms_per_step']*k,3), 'fused', j['roofline']['avg_launch_ms'])
PY
done
done
