#!/bin/bash
# Two PMC passes over a probe command: L2 fabric reads + L2 hits, then L2 misses + fabric writes + wave cycle buckets.
# usage: tools/pmc_probe.sh <tag> <kernel name fragment> <command...>
tag=$1; frag=$2; shift 2
mkdir -p gpurun_out/$tag
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD -d gpurun_out/$tag/p1 -o pmc --output-format csv -- "$@" > gpurun_out/$tag/p1.log 2>&1 || { echo pass 1 failed; tail -5 gpurun_out/$tag/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d gpurun_out/$tag/p2 -o pmc --output-format csv -- "$@" > gpurun_out/$tag/p2.log 2>&1 || { echo pass 2 failed; tail -5 gpurun_out/$tag/p2.log; exit 1; }
python3 - "$tag" "$frag" <<'PY'
import csv, glob, sys, collections
tag, frag = sys.argv[1:3]
for p in ("p1", "p2"):
    f = glob.glob("gpurun_out/%s/%s/**/*counter_collection.csv" % (tag, p), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if frag in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][-60:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, d in acc.items():
        print(k, "launches", len(n[k]), " ".join("%s=%.4g" % (c, v / len(n[k])) for c, v in sorted(d.items())))
PY
