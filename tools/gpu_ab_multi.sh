#!/bin/bash
# A/B of prebuilt libraries (tools/tmp_<name>.so), interleaved:  tools/gpu_ab_multi.sh <rounds> <name> <name> ...
mkdir -p gpurun_out/ab
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/tmp_$v.so airspy-fmradion_amd/libfmradion_amd.so
    timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err < /dev/null
    timeout 20 python - $v <<'PY'
import json,sys
v=sys.argv[1]
b=json.loads([l for l in open(f'gpurun_out/ab/{v}.json') if l.startswith('{')][-1]); k=b['kernel_ms_per_step']
print(v, b['value'], b['ms_per_step'], 'host', b['host_enqueue_ms_per_step'], 'fused', k.get('ifr_fused'), 'pll', k.get('pll'), 'pll_finish', k.get('pll_finish'), 'fm_out', k.get('fm_out'), 'audio', b['audio_check'].get('rms_err'), b['audio_check'].get('timed_step'))
PY
  done
done
