#!/bin/bash
# interleaved A/B of prebuilt libraries (tools/tmp_<name>.so) on the driver's form of the bench (20 steps), with the front end's
# per-workgroup stamps:  tools/gpu_ab_stamps.sh <rounds> <steps> <name> <name> ...
O=gpurun_out/abst; mkdir -p $O
rounds=$1; steps=$2; shift; shift
cp airspy-fmradion_amd/libfmradion_amd.so /tmp/keep.so
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/tmp_$v.so airspy-fmradion_amd/libfmradion_amd.so
    FMR_FE_STAMPS=1 timeout 200 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-r8b-leg > $O/$v.json 2> $O/$v.err < /dev/null
    timeout 20 python - $v $O <<'PY'
import json,sys
v,O=sys.argv[1:3]
try:
    b=json.loads([l for l in open(f'{O}/{v}.json') if l.startswith('{')][-1]); r=b['roofline']
    print(v, b['value'], 'ms/step', b['ms_per_step'], 'fused', r['avg_launch_ms'], 'n', r['launches_timed'], 'frac', r['frac'], 'kvb', (r.get('box_streaming_read') or {}).get('kernel_vs_box'), 'err', b['audio_check'].get('audio_rms_err_vs_oracle'))
except Exception as e:
    print(v, 'FAILED', e); print(open(f'{O}/{v}.err').read()[-600:])
PY
    grep "fe stamps" $O/$v.err | head -5
  done
done
cp /tmp/keep.so airspy-fmradion_amd/libfmradion_amd.so
