#!/bin/bash
# SQ counters of the fused tail's phase 2 alone (probe build -DTSPN_BT_ABL_NOP3) at the backbone's shapes.
# bash tools/pmc_tail_phase2.sh variants/libtspn_bt_nop3.so [frames]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/tail_p2; rm -rf $O; mkdir -p $O
[ -n "$1" ] && export TSPN_LIB_PATH=$R/$1
F=${2:-16}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/a -- python3 $R/tools/time_bt.py $F > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 $R/tools/time_bt.py $F > $O/b.log 2>&1
cd $R/tools && python3 - <<PY
import csv, glob, collections
from pmc_summary import short
for sub in ("a", "b"):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True)[0]
    c = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "bottleneck" in k:
            c[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(c):
        print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c[k].items())})
PY
