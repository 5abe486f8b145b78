#!/bin/bash
# round 6: where the other configurations stand (AM, IF filter, R8B, one-block host API, config 4)
O=gpurun_out/others6; mkdir -p $O
run() { name=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline < /dev/null > $O/$name.json 2> $O/$name.err; python - $O/$name.json <<'PY'
import json,sys
try:
    b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    st=(b.get('roofline') or {}).get('stage') or {}
    print(sys.argv[1].split('/')[-1], b['value'], b['ms_per_step'], 'stage', st.get('ms'), st.get('frac'), b.get('kernel_ms_per_step'), b.get('recurrences',{}).get('agc_newton_rounds'), b.get('recurrences',{}).get('agc_residuals'), b.get('latency_us'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
run am --mode am --steps 20 --warmup 3
run iffilter --if-filter --steps 20 --warmup 5
run r8b --resampler-class r8b --steps 20 --warmup 3
run block1 --api-mode block --steps 300 --blocks 400
run cfg4 --multipath-stages 64 --blocks 64 --steps 5 --warmup 3
