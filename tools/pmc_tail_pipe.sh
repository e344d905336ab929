#!/bin/bash
# SQ counters of the persistent pipelined tail and of the one-launch-per-tile tail (72 frames of the res4 shape):
# bash tools/pmc_tail_pipe.sh [lib relative to the repo]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${2:-default}
O=$R/gpurun_out/tail_pipe_$TAG; rm -rf $O; mkdir -p $O
[ -n "$1" ] && export TSPN_LIB_PATH=$R/$1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -- python3 $R/tools/time_bt_pipe.py 72 > $O/a.log 2>&1
cd $R/tools && python3 pmc_table.py $(find $O/a -name "*counter_collection.csv" | head -1) bottleneck
