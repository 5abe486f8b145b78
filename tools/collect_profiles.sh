#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: collects everything profiles/ is built from.
#   tools/collect_profiles.sh <tag>        e.g. r01
# Outputs under gpurun_out/<tag>_*; tools/finish_profiles.sh copies the summaries into profiles/.
set -u
tag=${1:-r01}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
cd "$root"
mkdir -p gpurun_out
[ "${2:-}" = "quick" ] || timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1 < /dev/null
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1 < /dev/null
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o ${tag} -- python bench.py --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_fetch.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_write.log 2>&1 < /dev/null
# tools/collect_profiles.sh <tag> quick: stop here (bench line, stats and PMC passes of the main configuration only;
# the lines of the other configurations stay as the last full collection left them under gpurun_out/)
if [ "${2:-}" = "quick" ]; then cut -c1-400 gpurun_out/${tag}_bench.json; exit 0; fi
# the three-kernel front end (FMR_NO_FUSED=1) under the same counters, for the traffic comparison
FMR_NO_FUSED=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_nofused.json 2> gpurun_out/${tag}_bench_nofused.err < /dev/null
FMR_NO_FUSED=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_fetch_nofused -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_fetch_nofused.log 2>&1 < /dev/null
FMR_NO_FUSED=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_write_nofused -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_write_nofused.log 2>&1 < /dev/null
# the other configs of BASELINE.json (lines only)
timeout 400 python bench.py --streams 32 --blocks 128 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_config5_32streams.json 2>/dev/null < /dev/null
timeout 400 python bench.py --multipath-stages 64 --blocks 64 --steps 5 --warmup 3 > gpurun_out/${tag}_bench_config4_E64.json 2>/dev/null < /dev/null
timeout 400 python bench.py --multipath-stages 64 --streams 32 --blocks 64 --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_config4_E64_32streams.json 2>/dev/null < /dev/null
for s in 128 256; do timeout 250 python bench.py --multipath-stages 64 --streams $s --blocks 64 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_config4_E64_${s}streams.json 2>/dev/null < /dev/null; done
timeout 400 python bench.py --mode am --steps 20 --warmup 3 > gpurun_out/${tag}_bench_config3_am.json 2>/dev/null < /dev/null
timeout 400 python bench.py --mode am --streams 32 --blocks 1024 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_config3_am_32streams.json 2>/dev/null < /dev/null
timeout 400 python bench.py --no-pilot --steps 3 --warmup 1 --blocks 256 --no-cpu-baseline > gpurun_out/${tag}_bench_no_pilot.json 2>/dev/null < /dev/null
for s in 64 256; do timeout 250 python bench.py --no-pilot --streams $s --blocks 32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_no_pilot_${s}streams.json 2>/dev/null < /dev/null; done
timeout 300 python bench.py --resampler-class r8b --steps 20 --warmup 3 > gpurun_out/${tag}_bench_r8b.json 2>/dev/null < /dev/null
timeout 300 python bench.py --if-filter --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_if_filter.json 2>/dev/null < /dev/null
for sg in 1e-2 3e-2; do timeout 200 python bench.py --sigma $sg --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_sigma_${sg}.json 2>/dev/null < /dev/null; done
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_driver_form.json 2>/dev/null < /dev/null
# round 6: the same form without the spin-up (the region as rounds 1-5 timed it), and with the per-workgroup / per-step stamps
timeout 200 python bench.py --steps 20 --warmup 5 --spinup-ms 0 --no-cpu-baseline > gpurun_out/${tag}_bench_driver_form_no_spinup.json 2>/dev/null < /dev/null
FMR_FE_STAMPS=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-r8b-leg 2>&1 > /dev/null < /dev/null | grep "fe stamps" > gpurun_out/${tag}_fe_stamps.txt
FMR_FE_STAMPS=1 timeout 200 python bench.py --steps 20 --warmup 5 --spinup-ms 0 --no-cpu-baseline --no-r8b-leg 2>&1 > /dev/null < /dev/null | grep "fe stamps" | sed 's/^/[no spin-up] /' >> gpurun_out/${tag}_fe_stamps.txt
# (the redirect belongs to the python process: behind the pipe it replaced grep's input and left the files of the first round-6 collection empty)
timeout 280 python tools/am_tol_check.py 2048 4 2>/dev/null < /dev/null | grep "^call" > gpurun_out/${tag}_am_agc_rounds.txt
timeout 200 python tools/block1_trace.py 2>/dev/null < /dev/null | grep -v amdgpu.ids > gpurun_out/${tag}_block1_trace.txt
timeout 200 python tools/step_time.py --steps 200 2>/dev/null < /dev/null | grep ms_per_step > gpurun_out/${tag}_step_time.txt
# the drop-in call: one block per fmr_process() through host buffers (latency percentiles)
timeout 200 python bench.py --api-mode block --steps 300 --blocks 400 --no-cpu-baseline > gpurun_out/${tag}_bench_block1.json 2>/dev/null < /dev/null
[ -x tools/bench_fused.bin ] && timeout 120 tools/bench_fused.bin > gpurun_out/${tag}_fused_harness.log 2>&1 < /dev/null
# the chain's own schedule trace, the equaliser's cycle account and rates, the PLL's mismatch history
timeout 120 python tools/step_timeline.py --show 2 --out gpurun_out/${tag}_step_timeline.txt > /dev/null 2>&1
timeout 120 python tools/step_timeline.py --show 1 --r8b --out gpurun_out/${tag}_step_timeline_r8b.txt > /dev/null 2>&1
timeout 120 python tools/step_timeline.py --show 1 --if-filter --out gpurun_out/${tag}_step_timeline_if_filter.txt > /dev/null 2>&1
timeout 120 python tools/step_timeline.py --show 1 --sigma 1e-2 --out gpurun_out/${tag}_step_timeline_sigma_1e-2.txt > /dev/null 2>&1 < /dev/null
timeout 120 python tools/step_timeline.py --show 2 --am --out gpurun_out/${tag}_step_timeline_am.txt > /dev/null 2>&1 < /dev/null
# round 6 (second collection): rocprofv3 kernel averages of the lines whose kernels changed -- R8B class, IF filter, AM -- their
# previous forms (FMR_NO_FUSED=1) on the same box, and the PMC traffic of the R8B class's front end
for cfg in "r8b --resampler-class r8b --steps 30 --warmup 5" "if_filter --if-filter --steps 30 --warmup 5" "am --mode am --steps 30 --warmup 3"; do
  set -- $cfg; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_$name -o s -- python bench.py "$@" --no-cpu-baseline > gpurun_out/${tag}_stats_$name.log 2>&1 < /dev/null
  cp gpurun_out/${tag}_stats_$name/s_kernel_stats.csv gpurun_out/${tag}_kernel_stats_$name.csv 2>/dev/null
done
FMR_NO_FUSED=1 timeout 300 python bench.py --resampler-class r8b --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_r8b_previous_form.json 2>/dev/null < /dev/null
FMR_NO_FUSED=1 timeout 300 python bench.py --if-filter --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_if_filter_previous_form.json 2>/dev/null < /dev/null
FMR_NO_FUSED=1 timeout 300 python bench.py --mode am --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_config3_am_previous_form.json 2>/dev/null < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_fetch_r8b -o f -- python bench.py --resampler-class r8b --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_fetch_r8b.log 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_write_r8b -o w -- python bench.py --resampler-class r8b --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_write_r8b.log 2>&1 < /dev/null
{ timeout 120 python tools/mpf_rate.py < /dev/null; } 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${tag}_mpf_account.txt
timeout 120 python tools/pll_mismatch.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${tag}_pll_mismatch.txt
# SQ counters of the R8B class's stage B (three --pmc passes, kernel trace only)
bash tools/gpu_pmc_r8b.sh k_ifr_poly5h > gpurun_out/${tag}_pmc_r8b_stage_b.txt 2>&1 < /dev/null
rm -rf gpurun_out/pmcr_1 gpurun_out/pmcr_2 gpurun_out/pmcr_3
tail -3 gpurun_out/${tag}_pytest_gpu.log
cat gpurun_out/${tag}_smoke.log | tail -2
cut -c1-400 gpurun_out/${tag}_bench.json
find gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write -name '*.csv' | head -20
