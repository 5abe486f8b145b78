#!/bin/bash
# Sweep of the pipelined chain's scheduling knobs on one box:  tools/gpu_pipe_sweep.sh [steps] ["ENV=.. ENV=.." ...]
# Every configuration is one bench.py process (same input, audio checked against the oracle inside the run); one summary
# line per configuration goes to gpurun_out/sweep/summary.txt, the full JSON lines to gpurun_out/sweep/<n>.json.
steps=${1:-100}; shift
mkdir -p gpurun_out/sweep
if [ $# -eq 0 ]; then
  set -- "FMR_PIPELINE=0" "FMR_PIPELINE=1" "FMR_FE_GATE=0" "FMR_FE_GATE=2" "FMR_PRIO=0" \
         "FMR_FE_CUS=248" "FMR_FE_CUS=240" "FMR_FE_CUS=224" "FMR_FE_CUS=208" "FMR_FE_CUS=192" \
         "FMR_FE_CUS=224 FMR_FE_GATE=0" "FMR_FE_CUS=224 FMR_FE_MASK=224" "FMR_PIPELINE=0"
fi
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 300 python bench.py --steps $steps --warmup 10 --no-cpu-baseline $BENCH_ARGS > gpurun_out/sweep/$i.json 2> gpurun_out/sweep/$i.err < /dev/null
  python - "$cfg" gpurun_out/sweep/$i.json <<'PY' | tee -a gpurun_out/sweep/summary.txt
import json, sys
cfg, fn = sys.argv[1], sys.argv[2]
try:
    b = json.loads([l for l in open(fn) if l.startswith('{')][-1])
    k = b['kernel_ms_per_step']; r = b['roofline']; a = b['audio_check']
    print("%-34s %9.1f MS/s  %.4f ms/step  fused %.4f ms frac %.3f  host %.3f  audio %s / %s  rounds %s" % (
        cfg, b['value'], b['ms_per_step'], r['avg_launch_ms'], r['frac'], b['host_enqueue_ms_per_step'],
        a.get('audio_rms_err_vs_oracle'), (a.get('timed_step') or {}).get('rms_err_vs_oracle'), b['recurrences']['pll_newton_rounds']))
except Exception as e:
    print("%-34s FAILED (%s): %s" % (cfg, e, open(fn.replace('.json', '.err')).read()[-600:]))
PY
done
