#!/bin/bash
# PLL chunk length experiment (library built with -DFMR_C_PLL_MIN=16): rounds, mismatches and per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in "$@"; do
  rm -rf gpurun_out/cp_$c; mkdir -p gpurun_out/cp_$c
  FMR_C_PLL=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cp_$c -o kb -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/cp_$c/bench.json 2> gpurun_out/cp_$c/bench.err
  f=$(find gpurun_out/cp_$c -name '*kernel_stats.csv' | head -1)
  echo "== c_pll $c"; python - $f gpurun_out/cp_$c/bench.json <<'PY'
import csv,sys,json
b=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
print('  ms_per_step',b['ms_per_step'],'pll',b['kernel_ms_per_step'].get('pll'),b['recurrences'])
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('void ','').replace('fmr::','').split('(')[0]
    if n.startswith('k_pll') and 'fallback' not in n: print(f"  {n[:40]:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
  find gpurun_out/cp_$c -name '*kernel_trace.csv' -delete
done
