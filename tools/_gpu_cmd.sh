mkdir -p gpurun_out/r6d && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_status_guard.py tests/test_gpu_wino63.py -m gpu -x -q > gpurun_out/r6d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6d/pytest.log
python bench.py --no-cpu-baseline --no-secondary --steps 20 > gpurun_out/r6d/bench.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6d/bench.json')); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), d['roofline']['clock_mhz']['mean'])
PY
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6d/stats -- python3 bench.py --no-cpu-baseline --no-secondary --steps 10 > gpurun_out/r6d/rp.log 2>&1 )
cd $GRAFT_REPO_ROOT && python tools/kernel_stats.py gpurun_out/r6d/stats/*/*_kernel_stats.csv | head -12
