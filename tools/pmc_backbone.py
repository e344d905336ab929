#!/usr/bin/env python3
"""Fabric traffic of the ResNet-101-C4 backbone per 720p frame from rocprofv3 PMC passes of tools/bench_backbone.py:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/bb_pmc/fetch -- python3 tools/bench_backbone.py \
        --frames 64 --chunk 8 --bf16 --iters 3
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/bb_pmc/write -- python3 tools/bench_backbone.py ...
    python3 tools/pmc_backbone.py gpurun_out/bb_pmc profiles/r3 --frames 64 --passes 4

The backbone is many launches of a few kernels, so the figure is the SUM over the dispatches of the tool's forward
passes (warm-up + iters) divided by passes x frames; the input generator and the one-off weight folding are left out.
Units and the gfx950 read correction as in tools/pmc_summary.py (KiB per dispatch; reads x2).  The result goes to
profiles/<round>/pmc_hbm_traffic_backbone.csv and, as set "cfg5:1", into profiles/pmc_traffic.json, where
bench.py --workload cfg5 reads `backbone_bytes_per_frame`."""
import argparse
import collections
import csv
import glob
import json
import os

from pmc_summary import short

BACKBONE_KERNELS = ("bottleneck_bf16_kernel", "bottleneck_block_bf16_kernel", "tail_io_bf16_kernel", "bottleneck_pipe_bf16_kernel", "conv2d_nhwc_bf16_kernel", "stem_conv_bf16_kernel", "stem_pool_bf16_kernel", "stem_s2d_bf16_kernel",
                    "max_pool_nhwc_bf16_kernel", "CatArrayBatchedCopy")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--passes", type=int, default=4, help="forward passes the tool ran (1 warm-up + --iters)")
    ap.add_argument("--name", default="pmc_hbm_traffic_backbone.csv", help="output file name under DST")
    ap.add_argument("--no-json", action="store_true", help="do not touch profiles/pmc_traffic.json")
    ap.add_argument("--note", default="--chunk 8")
    args = ap.parse_args()
    tot = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
    ours = set()
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(args.src, sub, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            raise SystemExit(f"no counter_collection.csv under {args.src}/{sub}")
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            if r["Counter_Name"] != ctr:
                continue
            k = short(r["Kernel_Name"])
            if r["Kernel_Name"].replace("void ", "", 1).startswith("(anonymous namespace)::"):
                ours.add(k)                 # a kernel of this library (every csrc kernel lives in an anonymous namespace)
            tot[k][ctr] += float(r["Counter_Value"]) * 1024.0
            if ctr == "FETCH_SIZE":
                tot[k]["n"] += 1
    per = args.passes * args.frames
    # Round 5 published 0.77 GB per frame for a backbone that moved 1.15: the table selected kernels by NAME and did not know
    # the new one-launch block kernels.  Every kernel of this library that is not a one-off (weight packing) has to be listed.
    unknown = sorted(k for k in ours if not any(b in k for b in BACKBONE_KERNELS) and not k.startswith("pack_") and "cast" not in k)
    if unknown:
        raise SystemExit(f"pmc_backbone: kernels of the library missing from BACKBONE_KERNELS: {unknown}")
    rows, total = [], 0.0
    for k, d in sorted(tot.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
        if not any(b in k for b in BACKBONE_KERNELS):
            continue
        by = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) / per
        rows.append((k, d["n"] / args.passes, 2.0 * d["FETCH_SIZE"] / per, d["WRITE_SIZE"] / per, by))
        total += by
    os.makedirs(args.dst, exist_ok=True)
    name = os.path.join(args.dst, args.name)
    with open(name, "w") as fh:
        fh.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/bench_backbone.py --frames "
                 f"{args.frames} {args.note} --bf16; bytes per 720p frame = sum over the dispatches of {args.passes} forward "
                 "passes / (passes x frames); reads x2 (gfx950, tools/pmc_summary.py)\n")
        fh.write("kernel,launches_per_pass,read_bytes_per_frame,write_bytes_per_frame,bytes_per_frame\n")
        for r in rows:
            fh.write("%s,%.1f,%.0f,%.0f,%.0f\n" % r)
        fh.write("total,,,,%.0f\n" % total)
    if args.no_json:
        print(open(name).read())
        return
    path = os.path.join(os.path.dirname(args.dst.rstrip("/")), "pmc_traffic.json")
    allsets = json.load(open(path))
    allsets["sets"]["cfg5:1"] = {"workload": "cfg5", "videos_per_launch": 1, "source": name,
                                 "backbone_bytes_per_frame": total,
                                 "kernels": {r[0]: {"hbm_bytes_per_frame": r[4], "launches_per_pass": r[1]} for r in rows}}
    with open(path, "w") as fh:
        json.dump(allsets, fh, indent=1, sort_keys=True)
    print(open(name).read())


if __name__ == "__main__":
    main()
