#!/bin/bash
mkdir -p gpurun_out/am
timeout 300 python bench.py --mode am --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/am/am1.json 2>/dev/null
timeout 300 python bench.py --mode am --streams 32 --blocks 1024 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/am/am32.json 2>/dev/null
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/am/fm.json 2>/dev/null
python - <<'PY'
import json
for f in ['am1','am32','fm']:
    b=json.loads([l for l in open(f'gpurun_out/am/{f}.json') if l.startswith('{')][-1])
    print(f, b['value'], b['ms_per_step'], {k:v for k,v in b['kernel_ms_per_step'].items() if k in ('if_agc','pll','ifr_fused','am_tail','fm_block','ifr_poly')}, b['recurrences'].get('agc_newton_rounds'), b['recurrences'].get('agc_serial_fallback'), b['recurrences'].get('agc_residuals'))
PY
