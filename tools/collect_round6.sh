#!/bin/bash
# Turn gpurun_out/prof6 (tools/profile_round6.sh) into the tracked summaries under profiles/r6/ (run in the build container).
# gpurun MERGES a run's files into gpurun_out/ next to those of earlier runs: every lookup below takes the newest file.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof6
mkdir -p profiles/r6/probes
for w in cfg2 cfg3 backbone pb; do
  f=$(ls -t $(find $O/stats_$w -name "*kernel_stats.csv") | head -1)
  n=bench_kernel_stats.csv; [ $w = cfg3 ] && n=bench_cfg3_kernel_stats.csv; [ $w = backbone ] && n=backbone_kernel_stats.csv; [ $w = pb ] && n=pair_builder_kernel_stats.csv
  python tools/kernel_stats.py $f profiles/r6/$n > /dev/null
done
cp $O/bench_latest.json $O/bench_cfg3.json $O/bench_cfg5.json $O/bench_cfg5_chunk36.json $O/bench_cfg5_associate.json \
   $O/bench_cfg5_associate_thread.json $O/association_bench.json $O/pair_builder.txt $O/time_tail_io.txt profiles/r6/
[ -f $O/ab_tail_io.txt ] && grep -v amdgpu.ids $O/ab_tail_io.txt > profiles/r6/probes/tail_io_r5_vs_r6.txt
cp $O/tail_io_sq_counters.txt profiles/r6/probes/tail_io_sq_counters.txt
python tools/pmc_summary.py $O/pmc_cfg2 profiles/r6 --videos 16 --workload cfg2 > /dev/null
( cd tools && python pmc_backbone.py ../$O/bb_pmc ../profiles/r6 --frames 72 --passes 4 --note "one-launch blocks + role-split tails, chunk 18" | tail -3 )
python - <<'PY'
import json
p = 'profiles/pmc_traffic.json'
d = json.load(open(p))
for k, v in d['sets'].items():
    if k.startswith('cfg5'):
        v['source'] = 'profiles/r6/pmc_hbm_traffic_backbone.csv'
    if k.startswith('cfg2'):
        v['source'] = 'profiles/r6/pmc_hbm_traffic.csv'
        # entries of kernels this pass did not launch keep the source they were measured in; drop stale template-less twins
        for name in [n for n in v['kernels'] if n + '<true>' in v['kernels'] or n + '<false>' in v['kernels']]:
            del v['kernels'][name]
json.dump(d, open(p, 'w'), indent=1, sort_keys=True)
PY
# pair builder: per-kernel fabric bytes
python - <<'PY'
import csv, glob, collections, os, re
O = 'gpurun_out/prof6'
def short(n):
    m = re.search(r"(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)(<[^>]*>)?\(", n); return (m.group(1) + (m.group(2) or "")) if m else n[:60]
rows = collections.defaultdict(dict)
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = max(glob.glob(f"{O}/pb_pmc/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)   # gpurun merges runs
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            agg[short(r["Kernel_Name"])].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    for k, v in agg.items():
        rows[k][ctr] = (sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), len(v))
with open('profiles/r6/pair_builder.csv', 'w') as fh:
    fh.write("# tools/bench_pair_builder.py (cfg2 shape: N=32, T=150, D=2048 -> [992, 4096, 150] fp32) under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); KiB per dispatch as reported; reads x2 on gfx950 (tools/pmc_summary.py)\n")
    fh.write("kernel,dispatches,fetch_KiB_reported,write_KiB,avg_duration_us,fabric_bytes(2*fetch+write),fabric_TB_per_s,written_TB_per_s\n")
    for k, d in sorted(rows.items()):
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d and ("gather" in k or "geometry" in k):
            f_, w_ = d["FETCH_SIZE"], d["WRITE_SIZE"]; dur = (f_[1] + w_[1]) / 2
            tot = (2 * f_[0] + w_[0]) * 1024
            fh.write(f"{k},{f_[2]},{f_[0]:.0f},{w_[0]:.0f},{dur:.1f},{tot:.0f},{tot / dur / 1e6:.2f},{w_[0] * 1024 / dur / 1e6:.2f}\n")
print(open('profiles/r6/pair_builder.csv').read())
PY
# SQ counter tables
for d in sq_cfg2_a sq_cfg2_b sq_cfg2_c; do echo "== $d (bench.py, cfg2)"; python tools/pmc_table.py $(ls -t $(find $O/$d -name "*counter_collection.csv") | head -1) heads_pairgrid4 conv3_wino63; done > profiles/r6/sq_counters_cfg2.txt
for d in sq_cfg3_a sq_cfg3_b; do echo "== $d (bench.py --workload cfg3)"; python tools/pmc_table.py $(ls -t $(find $O/$d -name "*counter_collection.csv") | head -1) heads_pairgrid_bf16 conv3_bf16_big; done > profiles/r6/sq_counters_cfg3.txt
cp temporal-span-proposal-network-vidvrd_amd/kernel_resources.json profiles/r6/kernel_resources.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('profiles/r6/bench_*.json')):
    d = json.load(open(f)); r = d.get('roofline') or {}
    print(f"{f[12:]:36s} {d['value']:11.0f} {d['ms_per_step']:8.2f} ms frac={r.get('frac') and round(r['frac'], 4)} kernel_ms={r.get('avg_launch_ms') and round(r['avg_launch_ms'], 2)}", d['config'].get('stage_ms', ''))
PY
