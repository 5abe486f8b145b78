#!/bin/bash
mkdir -p gpurun_out/c33
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k dropout > gpurun_out/c33/diag.log 2>&1; python -c "import json;print(json.load(open(\"gpurun_out/parity_report_configs.json\"))[\"carrier_dropout_and_nan\"])" >> gpurun_out/c33/diag.log
tail -40 gpurun_out/c33/diag.log
