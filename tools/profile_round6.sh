#!/bin/bash
# Round-6 evidence on the GPU box (run from the repo root): bench lines (the default one carries the secondary legs), the
# frames -> video-level relations pipeline, rocprofv3 kernel stats of the same commands, PMC passes (FETCH_SIZE / WRITE_SIZE
# separately; SQ sets) for cfg2, cfg3, the backbone and the res4 tails, the materialising pair builder, the association.
# Outputs under gpurun_out/prof6/ (progress lines on stdout); tools/collect_round6.sh copies the summaries into profiles/r6/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof6
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench_latest.json 2> $OUT/bench_latest.err; echo "cfg2 (+ secondary legs) done"; cut -c1-200 $OUT/bench_latest.json
python3 bench.py --workload cfg3 > $OUT/bench_cfg3.json 2>> $OUT/bench_latest.err; echo "cfg3 done"
python3 bench.py --workload cfg5 > $OUT/bench_cfg5.json 2>> $OUT/bench_latest.err; echo "cfg5 done"; cut -c1-200 $OUT/bench_cfg5.json
python3 bench.py --workload cfg5 --no-cpu-baseline --frame-chunk 36 > $OUT/bench_cfg5_chunk36.json 2>> $OUT/bench_latest.err; echo "cfg5 (round-5 chunk) done"
python3 bench.py --workload cfg5 --associate process --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench_cfg5_associate.json 2>> $OUT/bench_latest.err; echo "cfg5 + association (process) done"
python3 bench.py --workload cfg5 --associate thread --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_cfg5_associate_thread.json 2>> $OUT/bench_latest.err; echo "cfg5 + association (thread) done"
python3 tools/bench_association.py --skip-reference > $OUT/association_bench.json 2>> $OUT/bench_latest.err; echo "association done"
python3 tools/bench_pair_builder.py > $OUT/pair_builder.txt 2>> $OUT/bench_latest.err
[ -f $ROOT/probe_builds/libtspn_tio_r5.so ] && bash tools/ab_tail_io.sh $ROOT/probe_builds/libtspn_tio_r5.so > $OUT/ab_tail_io.txt 2>&1
python3 tools/time_tail_io.py 18 36 90 > $OUT/time_tail_io.txt 2>> $OUT/bench_latest.err; echo "tail timings done"
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_cfg2 -- python3 bench.py --no-cpu-baseline --no-secondary > $OUT/rp_stats_cfg2.log 2>&1 ); echo "stats cfg2"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_cfg3 -- python3 bench.py --workload cfg3 --no-cpu-baseline > $OUT/rp_stats_cfg3.log 2>&1 ); echo "stats cfg3"
BB="--frames 72 --chunk 18 --bf16 --iters 3"      # (kernel-level tables stay at 18 frames per launch, as in rounds 3 - 5; the product default is 90)
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_backbone -- python3 tools/bench_backbone.py $BB --streams 1 > $OUT/rp_stats_bb.log 2>&1 ); echo "stats backbone"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_pb -- python3 tools/bench_pair_builder.py 5 > $OUT/rp_stats_pb.log 2>&1 ); echo "stats pair builder"
S="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/pmc_cfg2/fetch -- python3 bench.py $S > $OUT/rp_fetch2.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/pmc_cfg2/write -- python3 bench.py $S > $OUT/rp_write2.log 2>&1 ); echo "traffic cfg2"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/bb_pmc/fetch -- python3 tools/bench_backbone.py $BB > $OUT/rp_bbf.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/bb_pmc/write -- python3 tools/bench_backbone.py $BB > $OUT/rp_bbw.log 2>&1 ); echo "traffic backbone"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/pb_pmc/fetch -- python3 tools/bench_pair_builder.py 3 > $OUT/rp_pbf.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/pb_pmc/write -- python3 tools/bench_pair_builder.py 3 > $OUT/rp_pbw.log 2>&1 ); echo "traffic pair builder"
# issue / wait attribution of the conv and the pair stages (passes of eight counters)
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE $P -d $OUT/sq_cfg2_a -- python3 bench.py $S > $OUT/rp_sq2a.log 2>&1 ); echo "sq cfg2 a"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY $P -d $OUT/sq_cfg2_b -- python3 bench.py $S > $OUT/rp_sq2b.log 2>&1 ); echo "sq cfg2 b"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE $P -d $OUT/sq_cfg2_c -- python3 bench.py $S > $OUT/rp_sq2c.log 2>&1 ); echo "sq cfg2 c"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE $P -d $OUT/sq_cfg3_a -- python3 bench.py --workload cfg3 $S > $OUT/rp_sq3a.log 2>&1 ); echo "sq cfg3 a"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY $P -d $OUT/sq_cfg3_b -- python3 bench.py --workload cfg3 $S > $OUT/rp_sq3b.log 2>&1 ); echo "sq cfg3 b"
cd $ROOT && bash tools/pmc_tail_io.sh 18 > $OUT/pmc_tail_io.log 2>&1; cp gpurun_out/tail_io_pmc/summary.txt $OUT/tail_io_sq_counters.txt; echo "sq tails"
find $OUT -name "*.csv" | wc -l
