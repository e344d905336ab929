#!/bin/bash
# Turn gpurun_out/prof4 (tools/profile_round4.sh) into the tracked summaries under profiles/r4/ (run in the build container).
# gpurun merges a run's files into gpurun_out/ next to those of earlier runs: delete gpurun_out/prof4 before a new profile run (or
# its files older than the run) -- the PMC summaries read every csv they find.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof4
for w in cfg2 cfg3 cfg5 backbone; do
  f=$(ls -t $(find $O/stats_$w -name "*kernel_stats.csv") | head -1)    # newest: gpurun MERGES into gpurun_out/, older runs' files stay
  n=bench_kernel_stats.csv; [ $w = cfg3 ] && n=bench_cfg3_kernel_stats.csv; [ $w = cfg5 ] && n=bench_cfg5_kernel_stats.csv; [ $w = backbone ] && n=backbone_kernel_stats.csv
  python tools/kernel_stats.py $f profiles/r4/$n > /dev/null
done
cp $O/bench_latest.json $O/bench_ops_level.json $O/bench_cfg4_shard.json $O/bench_force_collective.json $O/bench_cfg3.json $O/bench_cfg5.json $O/association_bench.json profiles/r4/
for m in naive pageable pinned prefetch prefetch-pinned; do cp $O/bench_host_$m.json profiles/r4/; done
cp $O/backbone.txt profiles/r4/backbone_chunk9_vs_chunk8.txt
cp $O/time_bt.txt profiles/r4/time_bottleneck_tails.txt
python tools/pmc_summary.py $O/pmc_cfg2 profiles/r4 --videos 16 --workload cfg2 > /dev/null
python tools/pmc_summary.py $O/pmc_cfg3 profiles/r4 --videos 4 --workload cfg3 > /dev/null
( cd tools && python pmc_backbone.py ../$O/bb_pmc ../profiles/r4 --frames 72 --passes 4 --note "--chunk 9" | tail -3
  python pmc_backbone.py ../$O/bb_pmc_next ../profiles/r4 --frames 72 --passes 4 --note "--chunk 9 --next" --name pmc_hbm_traffic_backbone_next_conv1.csv --no-json | tail -2 )
python - <<'PY'
import json
p = 'profiles/pmc_traffic.json'
d = json.load(open(p))
d['sets']['cfg5:1']['source'] = 'profiles/r4/pmc_hbm_traffic_backbone.csv'
json.dump(d, open(p, 'w'), indent=1, sort_keys=True)
PY
cp temporal-span-proposal-network-vidvrd_amd/kernel_resources.json profiles/r4/kernel_resources.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('profiles/r4/bench_*.json')):
    d = json.load(open(f)); r = d.get('roofline') or {}
    print(f"{f[12:]:34s} {d['value']:11.0f} {d['ms_per_step']:8.2f} ms frac={r.get('frac') and round(r['frac'], 4)} kernel_ms={r.get('avg_launch_ms') and round(r['avg_launch_ms'], 2)}", d['config'].get('stage_ms', ''))
print(open('profiles/r4/association_bench.json').read())
print(open('profiles/r4/backbone_chunk9_vs_chunk8.txt').read())
print(open('profiles/r4/time_bottleneck_tails.txt').read())
PY
