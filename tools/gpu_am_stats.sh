#!/bin/bash
# per-kernel averages of the AM chain (config 3), rocprofv3 --kernel-trace --stats
mkdir -p gpurun_out/amstats
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/amstats -o am -- python $GRAFT_REPO_ROOT/bench.py --mode am --steps 30 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/amstats/bench.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/amstats -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print("%-70s calls %6s avg_us %9.2f tot_ms %9.3f" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
