#!/bin/bash
# rocprofv3 runs of the benchmark command for profiles/: kernel trace + stats, then two PMC passes
# (HBM-side read and write bytes).  Usage: tools/profile_bench.sh <tag>
tag=${1:-bench}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/trace -o bench -- $CMD > gpurun_out/$tag/trace.log 2>&1 || { echo trace failed; tail -5 gpurun_out/$tag/trace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/$tag/p1 -o pmc --output-format csv -- $CMD > gpurun_out/$tag/p1.log 2>&1 || { echo pmc1 failed; tail -5 gpurun_out/$tag/p1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/$tag/p2 -o pmc --output-format csv -- $CMD > gpurun_out/$tag/p2.log 2>&1 || { echo pmc2 failed; tail -5 gpurun_out/$tag/p2.log; exit 1; }
grep "^{\"metric\"" gpurun_out/$tag/trace.log > gpurun_out/${tag}_bench_line_under_trace.json
echo done
