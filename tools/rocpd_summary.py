#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (`--kernel-trace --stats`) into a per-kernel table:
calls, total/avg/min/max duration.  Usage: rocpd_summary.py results.db [> profiles/x.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print("%-90s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) <= 90 else name[:87] + "..."
        print("%-90s %8d %12.1f %10.3f %10.3f %10.3f %6.1f" % (short, a[0], a[1] / 1e3, a[1] / a[0] / 1e3,
                                                          a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))


if __name__ == "__main__":
    main(sys.argv[1])
