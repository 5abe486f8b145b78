#!/bin/bash
mkdir -p gpurun_out/c26
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "multipath or config4" > gpurun_out/c26/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c26/tests.log
timeout 300 python bench.py --multipath-stages 64 --blocks 64 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/c26/bench_c4.json 2> gpurun_out/c26/bench_c4.err
FMR_MPF_V2=1 timeout 300 python bench.py --multipath-stages 64 --blocks 64 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/c26/bench_c4_v2.json 2> gpurun_out/c26/bench_c4_v2.err
timeout 300 python bench.py --multipath-stages 64 --streams 32 --blocks 64 --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/c26/bench_c4_s32.json 2> gpurun_out/c26/bench_c4_s32.err
tail -5 gpurun_out/c26/tests.log
python - <<'PY'
import json
for n in ("bench_c4","bench_c4_v2","bench_c4_s32"):
    try:
        b=json.loads([l for l in open(f'gpurun_out/c26/{n}.json') if l.startswith('{')][-1])
        print(n, b['value'], b['ms_per_step'], b['kernel_ms_per_step'].get('mpf'), b['audio_check'].get('audio_rms_err_vs_oracle'))
    except Exception as e: print(n,'failed',e)
PY
