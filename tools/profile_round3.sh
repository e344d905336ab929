#!/bin/bash
# Round-3 evidence on the GPU box (run from the repo root): bench lines of every workload, rocprofv3 kernel stats of the
# same commands, PMC passes (FETCH_SIZE / WRITE_SIZE separately; SQ sets) for cfg2 and cfg3.  Outputs under
# gpurun_out/prof3/; tools/pmc_summary.py / tools/pmc_table.py + copies into profiles/r3/ are done afterwards.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof3
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench_latest.json 2> $OUT/bench_latest.err; echo "cfg2 done"; cut -c1-200 $OUT/bench_latest.json
python3 bench.py --ops-level --no-cpu-baseline > $OUT/bench_ops_level.json 2>> $OUT/bench_latest.err
python3 bench.py --workload cfg4 --no-cpu-baseline > $OUT/bench_cfg4_shard.json 2>> $OUT/bench_latest.err; echo "cfg4 done"
python3 bench.py --force-collective --no-cpu-baseline > $OUT/bench_force_collective.json 2>> $OUT/bench_latest.err; echo "collective done"
python3 bench.py --workload cfg3 > $OUT/bench_cfg3.json 2>> $OUT/bench_latest.err; echo "cfg3 done"
python3 bench.py --workload cfg5 > $OUT/bench_cfg5.json 2>> $OUT/bench_latest.err; echo "cfg5 done"; cut -c1-200 $OUT/bench_cfg5.json
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_cfg2 -- python3 bench.py --no-cpu-baseline > $OUT/rp_stats_cfg2.log 2>&1 ); echo "stats cfg2"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_cfg3 -- python3 bench.py --workload cfg3 --no-cpu-baseline > $OUT/rp_stats_cfg3.log 2>&1 ); echo "stats cfg3"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_cfg5 -- python3 bench.py --workload cfg5 --no-cpu-baseline --steps 2 > $OUT/rp_stats_cfg5.log 2>&1 ); echo "stats cfg5"
S="--steps 3 --warmup 1 --no-cpu-baseline"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/pmc_cfg2/fetch -- python3 bench.py $S > $OUT/rp_fetch2.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/pmc_cfg2/write -- python3 bench.py $S > $OUT/rp_write2.log 2>&1 ); echo "traffic cfg2"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/pmc_cfg3/fetch -- python3 bench.py --workload cfg3 $S > $OUT/rp_fetch3.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/pmc_cfg3/write -- python3 bench.py --workload cfg3 $S > $OUT/rp_write3.log 2>&1 ); echo "traffic cfg3"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU $P -d $OUT/sq_cfg2 -- python3 bench.py $S > $OUT/rp_sq2.log 2>&1 ); echo "sq cfg2"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU $P -d $OUT/sq_cfg3 -- python3 bench.py --workload cfg3 $S > $OUT/rp_sq3.log 2>&1 ); echo "sq cfg3"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE $P -d $OUT/sq_cfg3_lds -- python3 bench.py --workload cfg3 $S > $OUT/rp_sq3l.log 2>&1 ); echo "sq cfg3 lds"
find $OUT -name "*.csv" | wc -l
