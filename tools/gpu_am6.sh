#!/bin/bash
# round 6: AM (config 3) after a change -- its GPU tests, the bench line and the chain's own schedule trace
mkdir -p gpurun_out/am6
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "am or 384k or if_resampler or nbfm or ssb or usb or cw" > gpurun_out/am6/tests.log 2>&1 < /dev/null
echo "tests rc=$?" >> gpurun_out/am6/tests.log
tail -5 gpurun_out/am6/tests.log
timeout 300 python bench.py --mode am --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/am6/am1.json 2>/dev/null < /dev/null
timeout 300 python tools/step_timeline.py --am --out gpurun_out/am6/timeline.txt > /dev/null 2>&1 < /dev/null
python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/am6/am1.json') if l.startswith('{')][-1])
print('am1', b['value'], b['ms_per_step'], b['kernel_ms_per_step'], b['recurrences'].get('agc_newton_rounds'), b['audio_check'])
PY
head -70 gpurun_out/am6/timeline.txt
cat gpurun_out/parity_report.json | head -40
