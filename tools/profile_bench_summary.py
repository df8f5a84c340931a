#!/usr/bin/env python
"""Turn the rocprofv3 outputs of tools/profile_bench_rocprof.sh into the summaries kept under profiles/:
  <out>_kernel_stats.txt  per (kernel, grid size): launches, mean / min / max duration from the kernel trace
  <out>_pmc.txt           per (kernel, grid size): mean FETCH_SIZE and WRITE_SIZE (KiB) per launch
  <out>_traffic.json      what bench.py reads: bytes and mean duration per launch of the benchmark's kernels
Dispatches of one kernel are kept apart by grid size: the same step kernel runs with interpolation blocks in the
forward loops and without them in the gradient loops, and the 4- and 32-shot legs of the bench use other grids.
Usage: profile_bench_summary.py <dir with trace/ p1/ p2/> <output prefix>"""
import csv, glob, json, os, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]


def short(name):
    return name.split("(")[0].replace("void ", "").replace("mifwi::", "")


# ---- kernel trace --------------------------------------------------------------------------------------
dur = defaultdict(list)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[(short(row["Kernel_Name"]), int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"]))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
with open(out + "_kernel_stats.txt", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace of: python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batches --no-strong\n")
    fo.write("# durations in microseconds, per (kernel, grid size in threads)\n")
    fo.write("%-62s %10s %8s %12s %10s %10s %10s\n" % ("kernel", "grid", "calls", "total_us", "avg_us", "min_us", "max_us"))
    for (k, g), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        fo.write("%-62s %10d %8d %12.1f %10.3f %10.3f %10.3f\n" % (k[:62], g, len(v), sum(v), sum(v) / len(v), min(v), max(v)))

# ---- counters --------------------------------------------------------------------------------------------
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        pmc[(short(row["Kernel_Name"]), int(row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "_pmc.txt", "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate runs) of the same command; means per launch, KiB\n")
    fo.write("# FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950: double it before comparing with a byte count\n")
    for (k, g), cs in sorted(pmc.items()):
        if "step" not in k and "tti" not in k and "persist" not in k:
            continue
        fo.write("%-62s grid=%-9d %s\n" % (k[:62], g, "  ".join("%s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v))
                                                               for c, v in sorted(cs.items()))))


def pick(table, frag, grid):
    hits = [(k, v) for k, v in table.items() if frag in k[0] and k[1] == grid]
    return hits[0][1] if hits else None


def entry(parts):
    """parts: [(kernel fragment, grid)] of the launches that make up one loop; means weighted by launch count."""
    n = us = fetch = write = 0.0
    for frag, grid in parts:
        d, c = pick(dur, frag, grid), pick(pmc, frag, grid)
        if d is None or c is None or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            return None
        n += len(d)
        us += sum(d)
        fetch += len(d) * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        write += len(d) * sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    return {"avg_us": us / n, "fetch_kib": fetch / n, "write_kib": write / n,
            "bytes": int((2 * fetch / n + write / n) * 1024), "launches_in_trace": int(n)}


# grids of the benchmark: 8 shots x 98 tiles of 32 x 32 (+ 16 interpolation blocks in forward loops), 256 threads;
# 592^3: 370 tiles x 6 chunks (+ 16 interpolation blocks) forward, 370 x 4 chunks in the gradient loop
traffic = {"_comment": "Per launch, from rocprofv3 of `MIFWI_3D_BOX=0 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batches --no-strong` (3-D loops sweeping the whole grid): mean "
                       "duration from the kernel trace, fabric-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE "
                       "tallies 128-byte requests at 64 bytes on gfx950). For the 2-D kernels the working set lives in "
                       "the 256 MB last-level cache: these are bytes leaving the L2s, an upper bound on HBM traffic. "
                       "Gradient loops alternate a plain step and a step that adds the gradient terms of both: their "
                       "entries are means over the launches of the loop. persist2d_kernel entries: one launch runs a "
                       "whole time loop (nt - 2 = 1499 steps of 8 shots); the plain variant's launches mix saving "
                       "forward loops, forward-only loops and none else."}
spec = {
    "step2d_kernel": [("step2d_kernel<8, 32, 32, 0, 4>", 204800)],
    "step2d_kernel<GRAD>": [("step2d_kernel<8, 32, 32, 0, 4>", 200704), ("step2d_kernel<8, 32, 32, 4, 4>", 200704)],
}
for name, parts in spec.items():
    e = entry(parts)
    if e:
        traffic[name] = e
# whole-loop launches of the 2-D time loops (one launch = nt - 2 steps of 8 shots): every grid of the kernel
for name, frag in (("persist2d_kernel", "persist2d_kernel<8, 32, 32, 0,"), ("persist2d_kernel<GRAD>", "persist2d_kernel<8, 32, 32, 1,")):
    parts = [(k[0], k[1]) for k in dur if frag in k[0]]
    e = entry(parts) if parts else None
    if e:
        traffic[name] = e
# whole-loop launches of the 2-D TTI time loops (template arguments: radius, variant 0 forward / 1 adjoint / 2 gradient)
for mode, name in ((0, "persist_tti2d_kernel"), (1, "persist_tti2d_kernel<ADJ>"), (2, "persist_tti2d_kernel<GRAD>")):
    parts = [(k[0], k[1]) for k in dur if k[0].startswith("persist_tti2d_kernel<") and k[0].rstrip().endswith(", %d>" % mode)]
    e = entry(parts) if parts else None
    if e:
        traffic[name] = e
# 3-D: take whatever grids the trace holds for the order-16 forward step and the order-8 gradient loop
for name, frags in (("step3d_kernel", ["step3d_kernel<8, 16, 64, 4, 0, true>"]),
                    ("step3d_kernel<GRAD>", ["step3d_kernel<4, 16, 64, 4, 4, true>"])):
    parts = [(k[0], k[1]) for k in dur if any(f in k[0] for f in frags)]
    parts = [max(parts, key=lambda p: len(dur[p]))] if parts else []
    e = entry(parts) if parts else None
    if e:
        e["grid"] = parts[0][1]
        traffic[name] = e
json.dump(traffic, open(out + "_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
