mkdir -p gpurun_out/r6g && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_association.py tests/test_association.py -x -q > gpurun_out/r6g/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6g/pytest.log
python tools/bench_association.py --skip-reference > gpurun_out/r6g/association_bench.json 2>gpurun_out/r6g/ab.err; cat gpurun_out/r6g/association_bench.json
python bench.py --workload cfg5 --no-cpu-baseline --steps 4 > gpurun_out/r6g/bench_cfg5.json 2> gpurun_out/r6g/cfg5.err; echo "cfg5 rc=$?"
python bench.py --workload cfg5 --associate process --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r6g/bench_cfg5_associate.json 2> gpurun_out/r6g/cfg5a.err; echo "cfg5 associate rc=$?"; tail -3 gpurun_out/r6g/cfg5a.err
python bench.py --workload cfg5 --associate thread --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r6g/bench_cfg5_associate_thread.json 2> gpurun_out/r6g/cfg5t.err; echo "cfg5 associate thread rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6g/bench_cfg5.json')); print('cfg5', round(d['ms_per_step'],1), d['config']['stage_ms'])
for f in ('bench_cfg5_associate','bench_cfg5_associate_thread'):
    d=json.load(open('gpurun_out/r6g/%s.json'%f)); print(f, round(d['ms_per_step'],1), d['config']['stage_ms']); a=d['config']['association']; a.pop('what'); print(json.dumps(a))
PY
