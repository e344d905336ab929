#!/bin/bash
# SQ counters of the two res4 tail kernels (one-role `bottleneck_bf16_kernel<256>` and role-split `tail_io_bf16_kernel`) at the
# backbone's shape, two passes (rocprofv3 --pmc with --kernel-trace only; no TCP / TCC / TA counters: those hang on this pool).
#   bash tools/pmc_tail_io.sh [frames]        -> gpurun_out/tail_io_pmc/summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/tail_io_pmc; rm -rf $O; mkdir -p $O
F=${1:-18}
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/a -- python3 $R/tools/time_tail_io.py $F > $O/a.log 2>&1
echo "pass a done"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 $R/tools/time_tail_io.py $F > $O/b.log 2>&1
echo "pass b done"
cd $R/tools && python3 - > $O/summary.txt <<PY
import csv, glob, collections
from pmc_summary import short
print("SQ counters per launch (mean over the launches of tools/time_tail_io.py $F), summed over the chip as rocprofv3 reports them")
for sub in ("a", "b"):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True)[0]
    c = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "bottleneck_bf16_kernel" in k or "tail_io" in k:
            c[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(c):
        m = {n: sum(v) / len(v) for n, v in sorted(c[k].items())}
        print(k, {n: round(v) for n, v in m.items()})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("SQ_BUSY_CYCLES"):
            print("   MFMA busy / SQ busy cycles = %.3f;  waves waiting (any) / wave cycles = %.3f;  issue-stalled (SQ_WAIT_INST_ANY) / wave cycles = %.3f" % (
                m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
cat $O/summary.txt
