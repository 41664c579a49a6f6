#!/bin/bash
# SQ counters of the pc_step launch per plan (tile 32 / chain 128 / chain 129) at 32000 rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3pmc; mkdir -p $O
for t in "$@"; do
  timeout 200 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/pmc_t$t -o run -- python scratch/pc_plan_run.py $t > $O/pmc_t$t.log 2>&1
  python scratch/pmc_summary.py $(find /tmp/pmc_t$t -name "*.db" | head -1) pc_step > $O/pmc_t$t.txt 2>&1; cat $O/pmc_t$t.txt
done
