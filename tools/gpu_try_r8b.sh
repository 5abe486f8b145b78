#!/bin/bash
# the R8B-class bench under one environment setting per argument ("NAME=VALUE" or "-"), interleaved:  tools/gpu_try_r8b.sh <rounds> - FMR_NO_FUSED=1 ...
mkdir -p gpurun_out/tryr8b
rounds=$1; shift
for r in $(seq 1 $rounds); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    if [ "$v" = "-" ]; then env_cmd=""; else env_cmd="env $v"; fi
    timeout 300 $env_cmd python bench.py --resampler-class r8b --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/tryr8b/$i.json 2> gpurun_out/tryr8b/$i.err < /dev/null
    python - "$v" $i <<'PY'
import json,sys
v,i=sys.argv[1:3]
try:
    b=json.loads([l for l in open(f'gpurun_out/tryr8b/{i}.json') if l.startswith('{')][-1]); k=b['kernel_ms_per_step']
    print('%-16s %9.1f MS/s %.4f ms  decim %.4f poly %.4f disc %s pll %s  err %s' % (v, b['value'], b['ms_per_step'], k.get('ifr_decim',0), k.get('ifr_poly',0), k.get('disc'), k.get('pll'), b['audio_check'].get('audio_rms_err_vs_oracle')))
except Exception as e:
    print(v, 'FAILED', e); print(open(f'gpurun_out/tryr8b/{i}.err').read()[-800:])
PY
  done
done
