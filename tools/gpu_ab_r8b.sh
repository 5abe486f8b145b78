#!/bin/bash
# interleaved A/B of prebuilt libraries on the R8B class:  tools/gpu_ab_r8b.sh <rounds> <name> ...
O=gpurun_out/abr8b; mkdir -p $O
rounds=$1; shift
cp airspy-fmradion_amd/libfmradion_amd.so /tmp/keep.so
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/tmp_$v.so airspy-fmradion_amd/libfmradion_amd.so
    timeout 200 python bench.py --resampler-class r8b --steps 20 --warmup 3 --no-cpu-baseline < /dev/null > $O/$v.json 2> $O/$v.err
    python - $v $O <<'PY'
import json,sys
v,O=sys.argv[1:3]
try:
    b=json.loads([l for l in open(f'{O}/{v}.json') if l.startswith('{')][-1]); k=b['kernel_ms_per_step']; st=b['roofline']['stage']
    print('%-8s %8.1f GS/s %.4f ms  stage %.4f frac %.3f  decim %.4f poly %.4f disc %s  audio %s' % (v, b['value']/1e3, b['ms_per_step'], st['ms'], st['frac'], k.get('ifr_decim',0), k.get('ifr_poly',0), k.get('disc'), b['audio_check']))
except Exception as e:
    print(v, 'FAILED', e); print(open(f'{O}/{v}.err').read()[-800:])
PY
  done
done
cp /tmp/keep.so airspy-fmradion_amd/libfmradion_amd.so
