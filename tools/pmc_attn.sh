#!/bin/bash
# SQ counters of the split attention kernels on the configs[3] stage shapes, three separate --pmc passes (no trace domains).
# Usage (on the GPU box): bash tools/pmc_attn.sh <kernel-name filter>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  i=$((i + 1))
  rm -rf $R/gpurun_out/pmca_$i
  rocprofv3 --kernel-trace --pmc $pass -d $R/gpurun_out/pmca_$i --output-format csv -- python3 $R/tools/attn_microbench.py 2 > $R/gpurun_out/pmca_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmca_$i.log; exit 1; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmca_$i $1
done
