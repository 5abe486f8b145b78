#!/bin/bash
# PLL chunk length sweep on the headline workload: tools/gpu_cpll.sh 40 48 56 64
for r in 1 2; do
for c in "$@"; do
  FMR_C_PLL=$c timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('c_pll', $c, d['value'], d['ms_per_step'], 'fused', k.get('ifr_fused'), 'pll', k.get('pll'), 'audio_err', d.get('audio_check',{}).get('max_over_ranks'))"
done
done
