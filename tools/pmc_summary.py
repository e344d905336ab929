#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>/pmc_hbm_traffic.csv
and profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/write -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc profiles/r1 --videos 8

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counter values are
KiB per dispatch; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (128-B
requests tallied at 64 B) — confirmed on this repo's transpose kernel, which reads exactly
307 200 KiB per launch and reports 153 615 — so reads are doubled; WRITE_SIZE is exact.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name):
    m = re.search(r"(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--videos", type=int, default=16)
    ap.add_argument("--workload", default="cfg2")
    args = ap.parse_args()
    rows, per_kernel = [], collections.defaultdict(dict)
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(args.src, sub, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            raise SystemExit(f"no counter_collection.csv under {args.src}/{sub}")
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):  # newest pass
            if r["Counter_Name"] != ctr:
                continue
            agg[short(r["Kernel_Name"])].append(
                (float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        for k, v in sorted(agg.items()):
            n = len(v)
            val, dur = sum(x[0] for x in v) / n, sum(x[1] for x in v) / n
            rows.append((ctr, k, n, val, dur))
            per_kernel[k][ctr] = val * 1024.0
    os.makedirs(args.dst, exist_ok=True)
    csv_name = "pmc_hbm_traffic.csv" if args.workload == "cfg2" else f"pmc_hbm_traffic_{args.workload}.csv"
    with open(os.path.join(args.dst, csv_name), "w") as fh:
        fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py "
                 f"--workload {args.workload} --videos {args.videos}; KiB per dispatch as reported (reads x2 on gfx950, see tools/pmc_summary.py)\n")
        fh.write("counter,kernel,dispatches,avg_value_KiB,avg_duration_us\n")
        for r in rows:
            fh.write("%s,%s,%d,%.1f,%.1f\n" % r)
    out = {"videos_per_launch": args.videos, "workload": args.workload,
           "source": os.path.join(args.dst, csv_name), "kernels": {}}
    for k, d in per_kernel.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            out["kernels"][k] = {"fetch_bytes_corrected": 2.0 * d["FETCH_SIZE"], "write_bytes": d["WRITE_SIZE"],
                                 "hbm_bytes": 2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]}
    # profiles/pmc_traffic.json holds one measurement set per "<workload>:<videos per launch>"
    path = os.path.join(os.path.dirname(args.dst.rstrip("/")), "pmc_traffic.json")
    try:
        allsets = json.load(open(path))
        if "sets" not in allsets:
            allsets = {"sets": {}}
    except (OSError, ValueError):
        allsets = {"sets": {}}
    key = f"{args.workload}:{args.videos}"
    for k, d in allsets["sets"].get(key, {}).get("kernels", {}).items():
        # kernels of an earlier pass that this one did not launch (another --conv): keep, with their source
        out["kernels"].setdefault(k, dict(d, source=d.get("source", allsets["sets"][key].get("source"))))
    allsets["sets"][key] = out
    with open(path, "w") as fh:
        json.dump(allsets, fh, indent=1, sort_keys=True)
    print(json.dumps(out["kernels"], indent=1)[:1500])


if __name__ == "__main__":
    main()
