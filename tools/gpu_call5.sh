#!/bin/bash
mkdir -p gpurun_out/c5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_stream_loop.py tests/test_facade.py -m gpu -x -q -k "am or nbfm or stream_loop or facade or ssb or cw" > gpurun_out/c5/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c5/tests.log
timeout 300 python bench.py --mode am --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/c5/bench_c3.json 2> gpurun_out/c5/bench_c3.err
timeout 300 python bench.py --mode am --streams 32 --blocks 1024 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/c5/bench_c3_s32.json 2> gpurun_out/c5/bench_c3_s32.err
tail -12 gpurun_out/c5/tests.log
