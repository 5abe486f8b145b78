#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of the default bench at several step sizes
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for B in "$@"; do
  rm -rf gpurun_out/kb_$B; mkdir -p gpurun_out/kb_$B
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kb_$B -o kb -- python bench.py --blocks $B --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/kb_$B/bench.json 2> gpurun_out/kb_$B/bench.err
  f=$(find gpurun_out/kb_$B -name '*kernel_stats.csv' | head -1)
  echo "== blocks $B"; python - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('void ','').replace('fmr::','').split('(')[0]
    if float(r['AverageNs'])>3000: print(f"  {n[:40]:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
  find gpurun_out/kb_$B -name '*kernel_trace.csv' -delete
done
