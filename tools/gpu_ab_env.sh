#!/bin/bash
# interleaved A/B on bench.py: variants are "name[@lib][:VAR=val,VAR=val...]" (lib = tools/tmp_<lib>.so, default the library in place)
#   tools/gpu_ab_env.sh <rounds> "<bench args>" base x:FMR_X_SPARE_ASIDE=1 y@lpl16:FMR_X_CPLL=80
O=gpurun_out/abenv; mkdir -p $O
rounds=$1; shift; args=$1; shift
cp airspy-fmradion_amd/libfmradion_amd.so /tmp/keep.so
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    nl=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
    name=${nl%%@*}; lib=""; [[ "$nl" == *@* ]] && lib=${nl#*@}
    if [ -n "$lib" ]; then cp tools/tmp_$lib.so airspy-fmradion_amd/libfmradion_amd.so; else cp /tmp/keep.so airspy-fmradion_amd/libfmradion_amd.so; fi
    env FMR_FE_STAMPS=1 $(echo $envs | tr ',' ' ') timeout 200 python bench.py $args --no-cpu-baseline --no-r8b-leg > $O/$name.json 2> $O/$name.err < /dev/null
    timeout 20 python - $name $O <<'PY'
import json,sys,re
v,O=sys.argv[1:3]
try:
    b=json.loads([l for l in open(f'{O}/{v}.json') if l.startswith('{')][-1]); r=b['roofline']; k=b['kernel_ms_per_step']; rc=b['recurrences']
    err=open(f'{O}/{v}.err').read()
    m=re.search(r'in front of the launch -> first workgroup ([\d.]+) us', err)
    print('%-10s %8.1f GS/s  %.4f ms  fused %.4f frac %.3f kvb %s  pll %s  start_delay %s  rounds %s mism %s err %.3g' % (v, b['value']/1e3, b['ms_per_step'], r['avg_launch_ms'], r['frac'], (r.get('box_streaming_read') or {}).get('kernel_vs_box'), k.get('pll'), m.group(1) if m else None, rc.get('pll_newton_rounds'), rc.get('pll_mismatches'), b['audio_check'].get('audio_rms_err_vs_oracle')))
except Exception as e:
    print(v, 'FAILED', e); print(open(f'{O}/{v}.err').read()[-800:])
PY
  done
done
cp /tmp/keep.so airspy-fmradion_amd/libfmradion_amd.so
