ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof5
cd $ROOT
rm -rf $OUT/stats_backbone $OUT/bb_pmc $OUT/bb_pmc_chain
python3 tools/bench_backbone.py --frames 72 --bf16 --iters 5 > $OUT/backbone.txt 2>> $OUT/bench_latest.err
python3 tools/bench_backbone.py --frames 72 --bf16 --iters 5 --no-proj >> $OUT/backbone.txt 2>> $OUT/bench_latest.err
python3 tools/bench_backbone.py --frames 72 --bf16 --iters 5 --no-block >> $OUT/backbone.txt 2>> $OUT/bench_latest.err
python3 tools/bench_backbone.py --frames 72 --bf16 --iters 5 --chunk 9 >> $OUT/backbone.txt 2>> $OUT/bench_latest.err
cat $OUT/backbone.txt | cut -c1-120
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
BB="--frames 72 --bf16 --iters 3"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_backbone -- python3 tools/bench_backbone.py $BB --streams 1 > $OUT/rp_stats_bb.log 2>&1 ); echo "stats backbone"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/bb_pmc/fetch -- python3 tools/bench_backbone.py $BB > $OUT/rp_bbf.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/bb_pmc/write -- python3 tools/bench_backbone.py $BB > $OUT/rp_bbw.log 2>&1 ); echo "traffic backbone"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/bb_pmc_chain/fetch -- python3 tools/bench_backbone.py $BB --no-block > $OUT/rp_bbcf.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/bb_pmc_chain/write -- python3 tools/bench_backbone.py $BB --no-block > $OUT/rp_bbcw.log 2>&1 ); echo "traffic backbone, round-4 chain"
