#!/bin/bash
# round 6, first look: the fp16-edge tests, marker events vs dispatch events, per-workgroup stamps of the front end in the chain
O=gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_levels.py -m gpu -q < /dev/null > $O/levels.log 2>&1
echo "rc=$?" >> $O/levels.log
cp gpurun_out/parity_report_levels.json $O/ 2>/dev/null
show() { python - "$1" <<'PY'
import json,sys
f=sys.argv[1]
try:
    b=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=b['roofline']
    print(f, b['value'], 'ms/step', b['ms_per_step'], 'fused', r['avg_launch_ms'], 'n', r['launches_timed'], 'frac', r['frac'], 'box', (r.get('box_streaming_read') or {}).get('GB/s'), 'kvb', (r.get('box_streaming_read') or {}).get('kernel_vs_box'), 'err', b['audio_check'].get('audio_rms_err_vs_oracle'))
except Exception as e:
    print(f, 'FAILED', e)
PY
}
for i in 1 2; do
  FMR_EVT_MARKERS=1 FMR_FE_STAMPS=1 FMR_BENCH_SERIES=$O/series_mk_$i.json timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-r8b-leg < /dev/null > $O/b20_mk_$i.json 2> $O/b20_mk_$i.err; show $O/b20_mk_$i.json; grep "fe stamps" $O/b20_mk_$i.err
  FMR_FE_STAMPS=1 FMR_BENCH_SERIES=$O/series_ex_$i.json timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-r8b-leg < /dev/null > $O/b20_ex_$i.json 2> $O/b20_ex_$i.err; show $O/b20_ex_$i.json; grep "fe stamps" $O/b20_ex_$i.err
done
FMR_FE_STAMPS=1 FMR_BENCH_SERIES=$O/series_200.json timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-r8b-leg < /dev/null > $O/b200.json 2> $O/b200.err; show $O/b200.json; grep "fe stamps" $O/b200.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-r8b-leg < /dev/null > $O/b200_plain.json 2> $O/b200_plain.err; show $O/b200_plain.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-r8b-leg < /dev/null > $GRAFT_REPO_ROOT/$O/b20_prof.json 2> $GRAFT_REPO_ROOT/$O/b20_prof.err
cd $GRAFT_REPO_ROOT; show $O/b20_prof.json
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r6a/prof/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:12]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
# the fused launches of the timed region from the kernel trace: the last 20 full-size ones
for f in glob.glob('gpurun_out/r6a/prof/**/*kernel_trace.csv', recursive=True):
    d=[(int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(f)) if 'k_ifr_fused' in r['Kernel_Name']]
    d.sort()
    big=[x[1] for x in d if x[1]>150000]
    print('fused launches', len(d), 'full-size', len(big), 'last 20 avg us', sum(big[-24:-4])/20/1000 if len(big)>=24 else None)
PY
rm -rf $O/prof/*/*.db 2>/dev/null; du -sh $O
