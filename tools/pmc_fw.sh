#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes for a probe command.  Usage: tools/pmc_fw.sh <tag> <cmd...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp -d gpurun_out/$tag/p$i -o pmc --output-format csv -- "$@" > gpurun_out/$tag/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/$tag/p$i.log; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/$tag step2d > gpurun_out/${tag}_summary.txt
rm -rf gpurun_out/$tag
cat gpurun_out/${tag}_summary.txt
