#!/bin/bash
# round 6: the R8B class after a change -- its GPU tests, the bench line and the chain's own schedule trace
mkdir -p gpurun_out/r8b6
timeout 1200 python -m pytest tests/test_gpu_r8b.py -m gpu -x -q > gpurun_out/r8b6/tests.log 2>&1 < /dev/null
echo "tests rc=$?" >> gpurun_out/r8b6/tests.log
tail -25 gpurun_out/r8b6/tests.log
timeout 300 python bench.py --resampler-class r8b --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r8b6/r8b.json 2>gpurun_out/r8b6/r8b.err < /dev/null
FMR_NO_FUSED=1 timeout 300 python bench.py --resampler-class r8b --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r8b6/r8b_nofused.json 2>/dev/null < /dev/null
timeout 300 python tools/step_timeline.py --r8b --out gpurun_out/r8b6/timeline.txt > /dev/null 2>&1 < /dev/null
python - <<'PY'
import json
for f in ('r8b','r8b_nofused'):
    try:
        b=json.loads([l for l in open('gpurun_out/r8b6/%s.json'%f) if l.startswith('{')][-1])
        print(f, b['value'], b['ms_per_step'], b['roofline']['stage'], b['audio_check'].get('audio_rms_err_vs_oracle'), b['audio_check'])
    except Exception as e: print(f, 'failed', e)
PY
tail -5 gpurun_out/r8b6/r8b.err
head -40 gpurun_out/r8b6/timeline.txt
