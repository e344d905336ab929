mkdir -p gpurun_out/r6j && cd $GRAFT_REPO_ROOT
for fc in 45 63 72 90 100 127 180 45 90; do
python bench.py --workload cfg5 --no-cpu-baseline --steps 3 --frame-chunk $fc > gpurun_out/r6j/bench_cfg5_fc$fc.json 2> gpurun_out/r6j/err.txt
python - <<PY
import json
d=json.load(open('gpurun_out/r6j/bench_cfg5_fc$fc.json')); print($fc, round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['config']['stage_ms'].items()}, round(d['roofline']['frac'],4))
PY
done
