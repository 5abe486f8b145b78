#!/bin/bash
# interleaved A/B of prebuilt libraries on bench.py --if-filter:  tools/gpu_ab_fir.sh <rounds> <name> ...
O=gpurun_out/abfir; mkdir -p $O
rounds=$1; shift
cp airspy-fmradion_amd/libfmradion_amd.so /tmp/keep.so
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/tmp_$v.so airspy-fmradion_amd/libfmradion_amd.so
    timeout 200 python bench.py --if-filter --steps 100 --warmup 10 --no-cpu-baseline < /dev/null > $O/$v.json 2> $O/$v.err
    python - $v $O <<'PY'
import json,sys
v,O=sys.argv[1:3]
try:
    b=json.loads([l for l in open(f'{O}/{v}.json') if l.startswith('{')][-1]); st=b['roofline']['stage']
    print('%-8s %8.1f GS/s %.4f ms  stage %.4f frac %.3f %s audio %s' % (v, b['value']/1e3, b['ms_per_step'], st['ms'], st['frac'], st['kernels_ms'], b['audio_check'].get('audio_rms_err_vs_oracle')))
except Exception as e:
    print(v, 'FAILED', e); print(open(f'{O}/{v}.err').read()[-800:])
PY
  done
done
cp /tmp/keep.so airspy-fmradion_amd/libfmradion_amd.so
