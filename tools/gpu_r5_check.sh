#!/bin/bash
# round-5 check: GPU tests, the default bench (with its r8b leg), the front end on all CUs, two noisier signals
out=gpurun_out/r5
mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null; tail -4 $out/pytest.log
timeout 200 python bench.py --steps 100 --warmup 10 > $out/bench_default.json 2> $out/bench_default.err < /dev/null
FMR_FE_CUS=256 timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-r8b-leg > $out/bench_fe256.json 2> $out/bench_fe256.err < /dev/null
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-r8b-leg > $out/bench_default2.json 2> $out/bench_default2.err < /dev/null
for sg in 1e-2 3e-2; do timeout 120 python bench.py --sigma $sg --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_sigma_$sg.json 2> $out/bench_sigma_$sg.err < /dev/null; done
timeout 60 python - <<'PY' < /dev/null
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/bench_*.json')):
    try:
        b=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], b['value'], b['ms_per_step'], 'fused', b['roofline']['avg_launch_ms'], b['roofline']['frac'], 'pll', b['kernel_ms_per_step'].get('pll'), 'rounds', b['recurrences']['pll_newton_rounds'], b['recurrences']['pll_mismatches'], 'fb', b['recurrences']['pll_serial_fallback'], 'err', b['audio_check'].get('audio_rms_err_vs_oracle'), 'r8b', (b.get('r8b') or {}).get('value'), (b.get('r8b') or {}).get('stage'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
