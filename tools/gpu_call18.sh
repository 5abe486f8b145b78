#!/bin/bash
mkdir -p gpurun_out/c27
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "multipath or config4" > gpurun_out/c27/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c27/tests.log
for nw in 1 2 4; do
FMR_MPF_NW=$nw timeout 300 python bench.py --multipath-stages 64 --blocks 64 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/c27/bench_nw$nw.json 2> gpurun_out/c27/bench_nw$nw.err
done
tail -3 gpurun_out/c27/tests.log
python - <<'PY'
import json
for n in ("bench_nw1","bench_nw2","bench_nw4"):
    try:
        b=json.loads([l for l in open(f'gpurun_out/c27/{n}.json') if l.startswith('{')][-1])
        print(n, b['value'], b['ms_per_step'], b['kernel_ms_per_step'].get('mpf'), b['audio_check'].get('audio_rms_err_vs_oracle'))
    except Exception as e: print(n,'failed',e)
PY
