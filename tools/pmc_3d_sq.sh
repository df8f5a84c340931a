#!/bin/bash
# SQ counters of the 3-D forward kernel (two passes).  Usage: tools/pmc_3d_sq.sh <order> <tag>
so=${1:-16}; tag=${2:-pmc3dsq}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp -d gpurun_out/$tag/p$i -o pmc --output-format csv -- python3 tools/probe_3d_fwd.py $so > gpurun_out/$tag/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/$tag/p$i.log; }
done
python3 tools/pmc_summary.py gpurun_out/$tag step3d
