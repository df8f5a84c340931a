"""Per-kernel totals from a rocprofv3 --kernel-trace csv: python tools/kstats.py <dir> [name fragment]."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frag = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if frag in n:
        tot[n][0] += 1
        tot[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%8d calls  %10.1f us avg  %9.2f ms total  %s" % (c, t / c / 1e3, t / 1e6, n[:110]))
