#!/bin/bash
# Round profile on the GPU box (run from the repo root): bench line, rocprofv3 kernel stats of the same
# command, PMC HBM-traffic passes (FETCH_SIZE / WRITE_SIZE separately) and the MFMA-busy pass.
# Outputs under gpurun_out/prof/; tools/pmc_summary.py + copies into profiles/ are done afterwards.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench_latest.json 2> $OUT/bench_latest.err
echo "bench done"; cut -c1-300 $OUT/bench_latest.json
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 )
echo "stats done"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 )
echo "pmc traffic done"
( cd $ROOT && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc/mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfma.log 2>&1 )
echo "pmc mfma done"
find $OUT -name "*.csv" | head -20
