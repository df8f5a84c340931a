#!/bin/bash
# Bytes the L2s request from the fabric per 3-D forward step: FETCH_SIZE and WRITE_SIZE passes of tools/probe_3d_fwd.py.
# Usage: tools/pmc_3d_fetch.sh <order> <tag>
so=${1:-16}; tag=${2:-pmc3dfetch}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/$tag/p$i -o pmc --output-format csv -- python3 tools/probe_3d_fwd.py $so > gpurun_out/$tag/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/$tag/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for i, name in ((1, "FETCH_SIZE"), (2, "WRITE_SIZE")):
    f = glob.glob("gpurun_out/$tag/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not f: print("no csv for", name); continue
    tot = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == name and "step3d" in r["Kernel_Name"]:
            t = tot[r["Kernel_Name"][:60]]; t[0] += float(r["Counter_Value"]); t[1] += 1
    for k, (v, n) in tot.items(): print(name, k, "launches", n, "per launch KB %.4g" % (v / n))
PY
