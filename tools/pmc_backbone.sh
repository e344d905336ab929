#!/bin/bash
# PMC traffic passes of the backbone alone (GPU box): bash tools/pmc_backbone.sh ; then python3 tools/pmc_backbone.py gpurun_out/bb_pmc profiles/rN
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/bb_pmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/bench_backbone.py --frames 64 --chunk 8 --bf16 --iters 3 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/bench_backbone.py --frames 64 --chunk 8 --bf16 --iters 3 > $O/write.log 2>&1
echo done
