#!/bin/bash
# is the pipelined chain host-bound?  the step at shrinking call sizes, and the library's own host profile
mkdir -p gpurun_out/hb
for b in 2048 1024 512 128 32; do
  FMR_HOST_PROF=1 timeout 100 python bench.py --steps 200 --blocks $b --no-cpu-baseline --no-r8b-leg > gpurun_out/hb/b$b.json 2> gpurun_out/hb/b$b.err < /dev/null
  timeout 20 python - $b <<'PY'
import json,sys
b=sys.argv[1]
j=json.loads([l for l in open(f'gpurun_out/hb/b{b}.json') if l.startswith('{')][-1])
print('blocks', b, 'ms_per_step', j['ms_per_step'], 'fused', j['roofline']['avg_launch_ms'], 'pll', j['kernel_ms_per_step'].get('pll'))
PY
  grep "host prof" gpurun_out/hb/b$b.err | tail -2
done
