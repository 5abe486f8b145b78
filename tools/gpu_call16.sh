#!/bin/bash
mkdir -p gpurun_out/c22
for c in 32 48 64 96 128; do
  FMR_C_PLL=$c timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/c22/bench_c$c.json 2> gpurun_out/c22/bench_c$c.err
done
python - <<'PY'
import json
for c in (32,48,64,96,128):
    try:
        b=json.loads([l for l in open(f'gpurun_out/c22/bench_c{c}.json') if l.startswith('{')][-1])
        k=b['kernel_ms_per_step']; r=b['recurrences']
        print(c, b['value'], b['ms_per_step'], 'pll', k.get('pll'), 'fused', k.get('ifr_fused'), 'rounds', r['pll_newton_rounds'], r['pll_mismatches'][:3], 'fb', r['pll_serial_fallback'], 'audio', b['audio_check'].get('audio_rms_err_vs_oracle'))
    except Exception as e: print(c, 'failed', e)
PY
