#!/bin/bash
# bench under one environment setting per argument ("NAME=VALUE" or "-"):  tools/gpu_try.sh FMR_PIPELINE=1 - ...
mkdir -p gpurun_out/try
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then env_cmd=""; else env_cmd="env $v"; fi
  timeout 300 $env_cmd python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/try/$i.json 2> gpurun_out/try/$i.err
  python - "$v" $i <<'PY'
import json,sys
v,i=sys.argv[1:3]
try:
    b=json.loads([l for l in open(f'gpurun_out/try/{i}.json') if l.startswith('{')][-1]); k=b['kernel_ms_per_step']
    print(v, b['value'], b['ms_per_step'], 'enq', b['host_enqueue_ms_per_step'], 'fused', k.get('ifr_fused'), 'pll', k.get('pll'), 'fin', k.get('pll_finish'), 'err', b['audio_check'].get('audio_rms_err_vs_oracle'))
except Exception as e:
    print(v, 'FAILED', e); print(open(f'gpurun_out/try/{i}.err').read()[-800:])
PY
done
