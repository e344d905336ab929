mkdir -p gpurun_out/r6k && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "pair_gather or dense or geometry" > gpurun_out/r6k/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6k/pytest.log
python tools/bench_pair_builder.py 2>&1 | grep -v amdgpu | tee gpurun_out/r6k/pair_builder.txt
cd /tmp && export TMPDIR=/tmp
P="--output-format csv"
( cd $GRAFT_REPO_ROOT && rocprofv3 --pmc FETCH_SIZE $P -d gpurun_out/r6k/pb_pmc/fetch -- python3 tools/bench_pair_builder.py 3 > gpurun_out/r6k/rp_pbf.log 2>&1 )
( cd $GRAFT_REPO_ROOT && rocprofv3 --pmc WRITE_SIZE $P -d gpurun_out/r6k/pb_pmc/write -- python3 tools/bench_pair_builder.py 3 > gpurun_out/r6k/rp_pbw.log 2>&1 )
cd $GRAFT_REPO_ROOT && python tools/pmc_table.py $(find gpurun_out/r6k/pb_pmc/fetch -name "*counter_collection.csv") transpose_gather; python tools/pmc_table.py $(find gpurun_out/r6k/pb_pmc/write -name "*counter_collection.csv") transpose_gather
