#!/bin/bash
# bench.py --if-filter under one environment setting per argument ("NAME=VALUE" or "-"), interleaved:  tools/gpu_try_fir.sh <rounds> - FMR_NO_FUSED=1 ...
mkdir -p gpurun_out/tryfir
rounds=$1; shift
for r in $(seq 1 $rounds); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    if [ "$v" = "-" ]; then env_cmd=""; else env_cmd="env $v"; fi
    timeout 300 $env_cmd python bench.py --if-filter --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/tryfir/$i.json 2> gpurun_out/tryfir/$i.err < /dev/null
    python - "$v" $i <<'PY'
import json,sys
v,i=sys.argv[1:3]
try:
    b=json.loads([l for l in open(f'gpurun_out/tryfir/{i}.json') if l.startswith('{')][-1]); st=b['roofline']['stage']
    print('%-16s %9.1f MS/s %.4f ms  stage %.4f frac %.4f %s  err %s' % (v, b['value'], b['ms_per_step'], st['ms'], st['frac'], st['kernels_ms'], b['audio_check'].get('audio_rms_err_vs_oracle')))
except Exception as e:
    print(v, 'FAILED', e); print(open(f'gpurun_out/tryfir/{i}.err').read()[-800:])
PY
  done
done
