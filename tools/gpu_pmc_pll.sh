#!/bin/bash
# SQ / TA counters of the PLL and tail kernels (separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
           "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcp_$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcp_$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcp_$i.log 2>&1
  f=$(find gpurun_out/pmcp_$i -name '*counter_collection.csv' | head -1)
  python tools/pmc_table.py $f --filter "${1:-k_pll_shoot}" 
  find gpurun_out/pmcp_$i -name '*.csv' -size +8M -delete
done
