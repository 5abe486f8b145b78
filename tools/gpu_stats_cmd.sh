#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of one bench command line:  tools/gpu_stats_cmd.sh <tag> <bench args...>
tag=$1; shift
mkdir -p gpurun_out/stats_$tag
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/stats_$tag -o s -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/stats_$tag/bench.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/stats_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-86s calls %6s avg_us %9.2f tot_ms %9.3f" % (r['Name'][:86], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
