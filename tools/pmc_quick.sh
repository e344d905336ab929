#!/bin/bash
# usage: tools/pmc_quick.sh <outdir> <counter-set>... -- <python args>   (each set is one rocprofv3 --pmc pass)
export TMPDIR=/tmp
out=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
i=0
for s in "${sets[@]}"; do
  d=$out/p$i; mkdir -p $d
  rocprofv3 --pmc $s --output-format csv -d $d -- python3 "$@" > $d/log.txt 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"([A-Za-z0-9_]+)(<[^>]*>)?\(", r["Kernel_Name"])
    k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
    agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if ("conv3" in k or "heads" in k) and "pack" not in k:
        print(f"{k:45s} {c:38s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
  i=$((i+1))
done
