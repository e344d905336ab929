#!/usr/bin/env python3
"""Short per-kernel table from a rocprofv3 --kernel-trace --stats run (CSV `*_kernel_stats.csv` or the rocpd
`*_results.db`): python tools/kernel_stats.py FILE [out.csv]   (kernel names are cut to their identifier)."""
import csv
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(?:\(anonymous namespace\)::)?([A-Za-z_][A-Za-z0-9_]*)(<[^>(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


rows = []
if sys.argv[1].endswith(".db"):
    cur = sqlite3.connect(sys.argv[1]).cursor()
    for n, c, tot, mn, mx in cur.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) "
                                         "from kernels group by name"):
        rows.append((short(n), c, tot, tot / c, mn, mx))
else:
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((short(r["Name"]), int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]),
                     int(r["MinNs"]), int(r["MaxNs"])))
rows.sort(key=lambda r: -r[2])
total = sum(r[2] for r in rows)
out = ["kernel,calls,total_ms,avg_us,min_us,max_us,percent"]
for n, c, tot, avg, mn, mx in rows:
    out.append(f"{n},{c},{tot / 1e6:.3f},{avg / 1e3:.1f},{mn / 1e3:.1f},{mx / 1e3:.1f},{100 * tot / total:.2f}")
text = "\n".join(out) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text if len(rows) < 40 else "\n".join(out[:40]))
