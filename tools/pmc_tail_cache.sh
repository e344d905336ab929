#!/bin/bash
# L1 (TCP) / L2 (TCC) request counters of the fused tail at the backbone's shapes: how much of the weight stream hits L1.
# bash tools/pmc_tail_cache.sh [frames] [--with-ta]
# Every pass runs under `timeout` (as the PARENT of rocprofv3).  The TA_* pass hangs rocprofv3 on this pool
# (tools/README.md): it only runs with --with-ta.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/tail_cache; rm -rf $O; mkdir -p $O
F=${1:-9}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" $O/avail.txt | sort -u > $O/names.txt
run() { n=$1; shift; timeout -k 10 600 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/time_bt.py $F > $O/$n.log 2>&1 || echo "pass $n: rc $?"; }
run a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE
run b TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
run c TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
[ "$2" = "--with-ta" ] && run d TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
cd $R/tools && python3 - <<PY
import csv, glob, collections
from pmc_summary import short
for sub in "abcd":
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs:
        import os
        log = "$O/%s.log" % sub
        print(sub, "no csv:", open(log).read()[-400:] if os.path.exists(log) else "(pass not run)"); continue
    c = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if "bottleneck" in k or "conv2d" in k:
            c[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(c):
        print(sub, k, {n: round(sum(v) / len(v)) for n, v in sorted(c[k].items())})
PY
