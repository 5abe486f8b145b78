#!/bin/bash
# GPU suite, then the default bench under a few enqueue orders / PLL round forms (ablation switches of INTEGRATION.md section 4)
mkdir -p gpurun_out/sp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/sp/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/sp/tests.log
tail -4 gpurun_out/sp/tests.log
for v in default pllv1 nosplit agcearly default2; do
  unset FMR_NO_SPLIT FMR_AGC_EARLY FMR_PLL_V1
  case $v in
    nosplit) export FMR_NO_SPLIT=1;;
    agcearly) export FMR_AGC_EARLY=1;;
    pllv1) export FMR_PLL_V1=1;;
  esac
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/sp/$v.json 2> gpurun_out/sp/$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
b=json.loads([l for l in open(f'gpurun_out/sp/{v}.json') if l.startswith('{')][-1]); k=b['kernel_ms_per_step']
print(v, b['value'], b['ms_per_step'], 'fused', k.get('ifr_fused'), 'pll', k.get('pll'), 'audio_err', b.get('audio_rms_err'))
PY
done
