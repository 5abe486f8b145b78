#!/bin/bash
# Diagnostics: clocks / power of the box before, during and after a long bench run (read-only rocm-smi queries)
rocm-smi --showperflevel --showclocks --showpower 2>&1 | grep -vE "^$|====" | head -40
FMR_BENCH_SERIES=gpurun_out/series.json timeout 200 python bench.py --steps 6000 --warmup 0 --no-cpu-baseline < /dev/null > gpurun_out/series_bench.json 2>/dev/null &
pid=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power" | tr '\n' ' '; echo; sleep 0.7; done
wait $pid
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power" | tr '\n' ' '; echo
