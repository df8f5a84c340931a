#!/bin/bash
# Two PMC passes (L2-side traffic) for a probe command.  Usage: tools/pmc_quick.sh <tag> <cmd...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for grp in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp -d gpurun_out/$tag/p$i -o pmc --output-format csv -- "$@" > gpurun_out/$tag/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/$tag/p$i.log; exit 1; }
done
echo done
