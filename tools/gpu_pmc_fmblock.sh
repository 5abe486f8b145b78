#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcf_$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcf_$i -o p -- python bench.py --if-filter --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcf_$i.log 2>&1
  f=$(find gpurun_out/pmcf_$i -name '*counter_collection.csv' | head -1)
  python tools/pmc_table.py $f --filter "k_fm_block3"
  find gpurun_out/pmcf_$i -name '*.csv' -size +8M -delete
done
