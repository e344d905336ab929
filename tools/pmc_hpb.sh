#!/bin/bash
# SQ counters of the bf16 pair stage alone (tools/time_hpb.py 4 videos): bash tools/pmc_hpb.sh [variant.so]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/hpb_pmc; rm -rf $O; mkdir -p $O
[ -n "$1" ] && export TSPN_LIB_PATH=$R/$1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -- python3 $R/tools/time_hpb.py 4 3 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 $R/tools/time_hpb.py 4 3 > $O/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c -- python3 $R/tools/time_hpb.py 4 3 > $O/c.log 2>&1
cd $R/tools && for sub in a b c; do python3 pmc_table.py $(ls $O/$sub/*/*counter_collection.csv | head -1) heads_pairgrid_bf16; done
