#!/bin/bash
# Round-5 evidence on the GPU box (run from the repo root): bench lines of every workload, rocprofv3 kernel stats of the
# same commands, PMC passes (FETCH_SIZE / WRITE_SIZE separately; SQ set) for cfg2 and the backbone alone.
# Outputs under gpurun_out/prof5/ (progress lines on stdout); summaries are copied into profiles/r5/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof5
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench_latest.json 2> $OUT/bench_latest.err; echo "cfg2 done"; cut -c1-200 $OUT/bench_latest.json
python3 bench.py --ops-level --no-cpu-baseline > $OUT/bench_ops_level.json 2>> $OUT/bench_latest.err
python3 bench.py --workload cfg4 --no-cpu-baseline > $OUT/bench_cfg4_shard.json 2>> $OUT/bench_latest.err; echo "cfg4 done"
python3 bench.py --force-collective --no-cpu-baseline > $OUT/bench_force_collective.json 2>> $OUT/bench_latest.err; echo "collective done"
python3 bench.py --workload cfg3 > $OUT/bench_cfg3.json 2>> $OUT/bench_latest.err; echo "cfg3 done"
python3 bench.py --workload cfg5 > $OUT/bench_cfg5.json 2>> $OUT/bench_latest.err; echo "cfg5 done"; cut -c1-200 $OUT/bench_cfg5.json
python3 bench.py --workload cfg5 --no-cpu-baseline --tail-io-waves off > $OUT/bench_cfg5_tail_io_waves_off.json 2>> $OUT/bench_latest.err; echo "cfg5 (one-role tails) done"
python3 bench.py --workload cfg5 --no-cpu-baseline --tail-io-waves off --fused-block off > $OUT/bench_cfg5_fused_block_off.json 2>> $OUT/bench_latest.err; echo "cfg5 (chain) done"
python3 tools/bench_backbone.py --frames 144 --bf16 --iters 5 > $OUT/backbone.txt 2>> $OUT/bench_latest.err
python3 tools/bench_backbone.py --frames 144 --bf16 --iters 5 --no-io-waves >> $OUT/backbone.txt 2>> $OUT/bench_latest.err
python3 tools/bench_backbone.py --frames 144 --bf16 --iters 5 --no-io-waves --no-proj >> $OUT/backbone.txt 2>> $OUT/bench_latest.err
python3 tools/bench_backbone.py --frames 144 --bf16 --iters 5 --no-io-waves --no-block >> $OUT/backbone.txt 2>> $OUT/bench_latest.err
python3 tools/time_block.py 9 > $OUT/time_block.txt 2>> $OUT/bench_latest.err
python3 tools/time_block.py 18 >> $OUT/time_block.txt 2>> $OUT/bench_latest.err
python3 tools/time_tail_io.py 9 18 36 > $OUT/time_tail_io.txt 2>> $OUT/bench_latest.err
python3 tools/time_conv1x1.py 9 18 36 > $OUT/time_conv1x1.txt 2>> $OUT/bench_latest.err; echo "backbone timings done"
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_cfg2 -- python3 bench.py --no-cpu-baseline > $OUT/rp_stats_cfg2.log 2>&1 ); echo "stats cfg2"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_cfg3 -- python3 bench.py --workload cfg3 --no-cpu-baseline > $OUT/rp_stats_cfg3.log 2>&1 ); echo "stats cfg3"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_cfg5 -- python3 bench.py --workload cfg5 --no-cpu-baseline --steps 2 > $OUT/rp_stats_cfg5.log 2>&1 ); echo "stats cfg5"
BB="--frames 72 --chunk 18 --bf16 --iters 3"      # (kernel-level tables stay at 18 frames per launch, as in rounds 3 - 5; the product default is 36)
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_backbone -- python3 tools/bench_backbone.py $BB --streams 1 > $OUT/rp_stats_bb.log 2>&1 ); echo "stats backbone"
S="--steps 3 --warmup 1 --no-cpu-baseline"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/pmc_cfg2/fetch -- python3 bench.py $S > $OUT/rp_fetch2.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/pmc_cfg2/write -- python3 bench.py $S > $OUT/rp_write2.log 2>&1 ); echo "traffic cfg2"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/bb_pmc/fetch -- python3 tools/bench_backbone.py $BB > $OUT/rp_bbf.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/bb_pmc/write -- python3 tools/bench_backbone.py $BB > $OUT/rp_bbw.log 2>&1 ); echo "traffic backbone"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/bb_pmc_chain/fetch -- python3 tools/bench_backbone.py $BB --no-block --no-io-waves > $OUT/rp_bbcf.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/bb_pmc_chain/write -- python3 tools/bench_backbone.py $BB --no-block --no-io-waves > $OUT/rp_bbcw.log 2>&1 ); echo "traffic backbone, round-4 chain"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU $P -d $OUT/sq_cfg2 -- python3 bench.py $S > $OUT/rp_sq2.log 2>&1 ); echo "sq cfg2"
find $OUT -name "*.csv" | wc -l
