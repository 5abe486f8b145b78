#!/bin/bash
# SQ counters of the R8B class's stage kernels (separate --pmc passes, kernel trace only):  tools/gpu_pmc_r8b.sh [kernel substring]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcr_$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcr_$i -o p -- python bench.py --resampler-class r8b --steps 2 --warmup 1 --spinup-ms 0 --no-cpu-baseline > gpurun_out/pmcr_$i.log 2>&1
  f=$(find gpurun_out/pmcr_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_table.py $f --filter "${1:-k_ifr_poly5h}" || tail -5 gpurun_out/pmcr_$i.log
  find gpurun_out/pmcr_$i -name '*.csv' -size +8M -delete
done
