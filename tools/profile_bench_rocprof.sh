#!/bin/bash
# rocprofv3 runs of the benchmark command for profiles/: kernel trace, then two PMC passes (fabric-side read and
# write bytes), each in a run of its own.  Usage (on the GPU box): tools/profile_bench_rocprof.sh <tag>
# Leaves gpurun_out/<tag>_kernel_stats.txt, gpurun_out/<tag>_pmc.txt, gpurun_out/<tag>_traffic.json and the bench line.
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
# the 3-D kernels are profiled sweeping the whole grid (by default their loops skip what the wave has not reached,
# and a mean over such launches describes no launch)
export MIFWI_3D_BOX=0
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batches --no-strong"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/trace -o bench -- $CMD > gpurun_out/$tag/trace.log 2>&1 || { echo trace failed; tail -5 gpurun_out/$tag/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/$tag/p1 -o pmc --output-format csv -- $CMD > gpurun_out/$tag/p1.log 2>&1 || { echo pmc1 failed; tail -5 gpurun_out/$tag/p1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/$tag/p2 -o pmc --output-format csv -- $CMD > gpurun_out/$tag/p2.log 2>&1 || { echo pmc2 failed; tail -5 gpurun_out/$tag/p2.log; exit 1; }
grep "^{\"metric\"" gpurun_out/$tag/trace.log > gpurun_out/${tag}_bench_line_under_trace.json
python3 tools/profile_bench_summary.py gpurun_out/$tag gpurun_out/${tag} || exit 1
rm -rf gpurun_out/$tag/trace gpurun_out/$tag/p1 gpurun_out/$tag/p2
echo done
