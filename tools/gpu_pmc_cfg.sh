#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes, kernel trace only) of one bench configuration:  tools/gpu_pmc_cfg.sh <name> <bench args...>
name=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r06_pmc_fetch_$name -o f -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r06_pmc_fetch_$name.log 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r06_pmc_write_$name -o w -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r06_pmc_write_$name.log 2>&1 < /dev/null
ls gpurun_out/r06_pmc_fetch_$name gpurun_out/r06_pmc_write_$name
