#!/bin/bash
# round 6: FM with the IF filter (-f) after a change -- its GPU tests, the bench line, kernel averages
mkdir -p gpurun_out/fir6
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_gpu_parity.py -m gpu -x -q -k "if_filter or if_fir or fir or taps or partitions or mono_and_if" > gpurun_out/fir6/tests.log 2>&1 < /dev/null
echo "tests rc=$?" >> gpurun_out/fir6/tests.log
tail -25 gpurun_out/fir6/tests.log
timeout 300 python bench.py --if-filter --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/fir6/fir.json 2>gpurun_out/fir6/fir.err < /dev/null
FMR_NO_FUSED=1 timeout 300 python bench.py --if-filter --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/fir6/fir_nofused.json 2>/dev/null < /dev/null
python - <<'PY'
import json
for f in ('fir','fir_nofused'):
    try:
        b=json.loads([l for l in open('gpurun_out/fir6/%s.json'%f) if l.startswith('{')][-1])
        print(f, b['value'], b['ms_per_step'], b['roofline']['stage'], b['kernel_ms_per_step'], b['audio_check'].get('audio_rms_err_vs_oracle'), b['audio_check'].get('timed_step'))
    except Exception as e: print(f, 'failed', e)
PY
tail -5 gpurun_out/fir6/fir.err
