OUT=$GRAFT_REPO_ROOT/gpurun_out/prof6b; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python3 tools/bench_pair_builder.py > $OUT/pair_builder.txt 2>/dev/null; cat $OUT/pair_builder.txt
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_pb -- python3 tools/bench_pair_builder.py 5 > $OUT/rp_stats_pb.log 2>&1 ); echo "stats pair builder"
( cd $GRAFT_REPO_ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/pb_pmc/fetch -- python3 tools/bench_pair_builder.py 3 > $OUT/rp_pbf.log 2>&1 )
( cd $GRAFT_REPO_ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/pb_pmc/write -- python3 tools/bench_pair_builder.py 3 > $OUT/rp_pbw.log 2>&1 ); echo "traffic pair builder"
