#!/bin/bash
# round 6: full GPU suite, then the driver's form of the bench with and without the spin-up, twice each, interleaved
O=gpurun_out/suite6; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q < /dev/null > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for i in 1 2; do
  for sp in 60 0; do
    timeout 300 python bench.py --steps 20 --warmup 5 --spinup-ms $sp --no-cpu-baseline < /dev/null > $O/b20_sp${sp}_$i.json 2> $O/b20_sp${sp}_$i.err
    python - $O/b20_sp${sp}_$i.json <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=b['roofline']
print(sys.argv[1], b['value'], b['ms_per_step'], 'fused', r['avg_launch_ms'], 'n', r['launches_timed'], 'frac', r['frac'], 'kvb', (r.get('box_streaming_read') or {}).get('kernel_vs_box'), b['clock']['spinup_steps'], b['clock']['shader_mhz'], 'r8b', (b.get('r8b') or {}).get('value'), 'err', b['audio_check'].get('audio_rms_err_vs_oracle'), b['audio_check'].get('timed_step'))
PY
  done
done
