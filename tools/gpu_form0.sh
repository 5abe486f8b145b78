#!/bin/bash
# probe: 8-wave fused front end (-DFUSED_A_FORM=0, tools/tmp_form0.so) alone and with two chains side by side
cp airspy-fmradion_amd/libfmradion_amd.so /tmp/lib_default.so
cp tools/tmp_form0.so airspy-fmradion_amd/libfmradion_amd.so
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/form0.json 2> gpurun_out/form0.err
python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/form0.json') if l.startswith('{')][-1]); k=b['kernel_ms_per_step']
print('form0 single', b['value'], b['ms_per_step'], 'fused', k.get('ifr_fused'), 'err', b['audio_check'].get('audio_rms_err_vs_oracle'))
PY
timeout 400 python tools/bench_two_chains.py --chains 2 --steps 100 2>&1 | tail -2
cp /tmp/lib_default.so airspy-fmradion_amd/libfmradion_amd.so
