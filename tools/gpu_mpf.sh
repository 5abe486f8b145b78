#!/bin/bash
mkdir -p gpurun_out/mpf
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -k "multipath or config4" > gpurun_out/mpf/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/mpf/tests.log
timeout 300 python bench.py --multipath-stages 64 --blocks 64 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/mpf/bench.json 2> gpurun_out/mpf/bench.err
tail -3 gpurun_out/mpf/tests.log
python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/mpf/bench.json') if l.startswith('{')][-1])
print(b['value'], b['ms_per_step'], b['kernel_ms_per_step'].get('mpf'))
PY
