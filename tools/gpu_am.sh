#!/bin/bash
mkdir -p gpurun_out/am
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_stream_loop.py -m gpu -x -q -k "am or dsb or usb or lsb or cw or wspr or nbfm or medium or filter" > gpurun_out/am/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/am/tests.log
timeout 300 python bench.py --mode am --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/am/bench_am.json 2> gpurun_out/am/bench_am.err
FMR_FMBLOCK_V1=1 timeout 300 python bench.py --mode am --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/am/bench_am_v1.json 2> gpurun_out/am/bench_am_v1.err
timeout 300 python bench.py --mode am --streams 32 --blocks 1024 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/am/bench_am32.json 2> gpurun_out/am/bench_am32.err
tail -3 gpurun_out/am/tests.log
python - <<'PY'
import json
for n in ("bench_am","bench_am_v1","bench_am32"):
    try:
        b=json.loads([l for l in open(f'gpurun_out/am/{n}.json') if l.startswith('{')][-1])
        print(n, b['value'], b['ms_per_step'], b['kernel_ms_per_step'].get('fm_block'), b['audio_check'].get('audio_rms_err_vs_oracle'))
    except Exception as e: print(n,'failed',e)
PY
