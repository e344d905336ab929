#!/bin/bash
# Round-6 evidence, part 1 (run from the repo root on the GPU box): A/B of the res4 tail builds, the bench line with its
# secondary legs + kernel stats, SQ counters of the two pair stages and of the cfg2 conv, the materialising pair builder.
# Outputs under gpurun_out/prof6/ (progress lines on stdout).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof6
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
bash tools/ab_tail_io.sh $ROOT/probe_builds/libtspn_tio_r5.so $ROOT/probe_builds/libtspn_tio_masks.so > $OUT/ab_tail_io.txt 2>&1; echo "tail A/B done"; cat $OUT/ab_tail_io.txt | grep -v amdgpu.ids
python3 bench.py > $OUT/bench_latest.json 2> $OUT/bench_latest.err; echo "bench done"; cut -c1-200 $OUT/bench_latest.json
python3 tools/bench_pair_builder.py > $OUT/pair_builder.txt 2>&1; cat $OUT/pair_builder.txt | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
S="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_cfg2 -- python3 bench.py --no-cpu-baseline --no-secondary > $OUT/rp_stats_cfg2.log 2>&1 ); echo "stats cfg2"
( cd $ROOT && rocprofv3 --kernel-trace --stats $P -d $OUT/stats_pb -- python3 tools/bench_pair_builder.py 5 > $OUT/rp_stats_pb.log 2>&1 ); echo "stats pair builder"
( cd $ROOT && rocprofv3 --pmc FETCH_SIZE $P -d $OUT/pb_pmc/fetch -- python3 tools/bench_pair_builder.py 3 > $OUT/rp_pbf.log 2>&1 )
( cd $ROOT && rocprofv3 --pmc WRITE_SIZE $P -d $OUT/pb_pmc/write -- python3 tools/bench_pair_builder.py 3 > $OUT/rp_pbw.log 2>&1 ); echo "traffic pair builder"
# pair stages: issue / wait attribution (two passes of eight counters each, per workload)
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE $P -d $OUT/sq_cfg2_a -- python3 bench.py $S > $OUT/rp_sq2a.log 2>&1 ); echo "sq cfg2 a"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY $P -d $OUT/sq_cfg2_b -- python3 bench.py $S > $OUT/rp_sq2b.log 2>&1 ); echo "sq cfg2 b"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE $P -d $OUT/sq_cfg2_c -- python3 bench.py $S > $OUT/rp_sq2c.log 2>&1 ); echo "sq cfg2 c"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE $P -d $OUT/sq_cfg3_a -- python3 bench.py --workload cfg3 $S > $OUT/rp_sq3a.log 2>&1 ); echo "sq cfg3 a"
( cd $ROOT && rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY $P -d $OUT/sq_cfg3_b -- python3 bench.py --workload cfg3 $S > $OUT/rp_sq3b.log 2>&1 ); echo "sq cfg3 b"
find $OUT -name "*.csv" | wc -l
