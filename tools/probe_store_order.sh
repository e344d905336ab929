#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the fused bottleneck tail at the backbone's shapes (8 frames), one PMC pass each, and its
# timing: how many bytes a launch really writes against the size of its output (bottleneck_bf16_kernel, epilogue
# group order).  Run on the GPU box: bash tools/probe_store_order.sh [variant.so]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/store_order; mkdir -p $O
[ -n "$1" ] && export TSPN_LIB_PATH=$R/$1
cd $R
python tools/time_bt.py 16 2>&1 | grep CM=
python tools/time_bt.py 8 2>&1 | grep CM=
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/time_bt.py 8 > $O/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/time_bt.py 8 > $O/fetch.log 2>&1
cd $R/tools && python3 - <<PY
import csv, glob, collections
from pmc_summary import short
for sub, ctr, mul in (("write", "WRITE_SIZE", 1.0), ("fetch", "FETCH_SIZE", 2.0)):
    f = max(glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True), key=lambda p: __import__("os").path.getmtime(p))
    c = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if r["Counter_Name"] == ctr and ("bottleneck" in k or "conv2d_nhwc" in k):
            c[k].append(float(r["Counter_Value"]) * 1024 * mul / 1e6)
    for k, v in sorted(c.items()):
        print("%s %s: %d launches, %.1f MB avg (min %.1f, max %.1f)" % (ctr, k, len(v), sum(v) / len(v), min(v), max(v)))
PY
