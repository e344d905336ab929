#!/bin/bash
# One rocprofv3 --pmc pass (no tracing flags besides the implied kernel dispatch records):
#   tools/pmc_quick.sh OUTDIR "CTR1 CTR2 ..." -- script.py args...
# prints the per-kernel average of every counter.  Run from the repo root on the GPU box.
set -e
OUT=$1; CTRS=$2; shift 3
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $ROOT/$OUT; mkdir -p $ROOT/$OUT
( cd $ROOT && rocprofv3 --pmc $CTRS --output-format csv -d $ROOT/$OUT -- python3 "$@" > $ROOT/$OUT/run.log 2>&1 )
python3 - "$ROOT/$OUT" <<'PY'
import collections, csv, glob, os, re, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"([A-Za-z0-9_]+)(<[^>]*>)?\(", r["Kernel_Name"])
    k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:50]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in sorted(d.items())})
PY
