#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_table.py FILE [kernel substring ...]"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)(<[^>]*>)?\(", r["Kernel_Name"])
    k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:50]
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r:
        dur[(k, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, c in rows.items():
    ds = [v for (kk, _), v in dur.items() if kk == k]
    print(f"{k}: {len(ds)} dispatches, avg {sum(ds) / max(len(ds), 1):.1f} us")
    for name, v in sorted(c.items()):
        print(f"    {name:28s} {sum(v) / len(v):16.0f}")
