#!/bin/bash
# Collect PMC counters for the 3-D forward kernel, one pass per counter group (rocprofv3 does not
# split counters over passes).  Usage: tools/pmc_3d.sh <n> <order> <tag>
n=${1:-512}; so=${2:-16}; tag=${3:-pmc3d}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp -d gpurun_out/$tag/p$i -o pmc --output-format csv -- python3 tools/probe_3d.py $n $so > gpurun_out/$tag/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/$tag/p$i.log; exit 1; }
done
ls -R gpurun_out/$tag | head -30
