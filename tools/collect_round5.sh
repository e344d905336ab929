#!/bin/bash
# Turn gpurun_out/prof5 (tools/profile_round5.sh) into the tracked summaries under profiles/r5/ (run in the build container).
# gpurun merges a run's files into gpurun_out/ next to those of earlier runs: delete gpurun_out/prof5 before a new profile run.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof5
mkdir -p profiles/r5
for w in cfg2 cfg3 cfg5 backbone; do
  f=$(ls -t $(find $O/stats_$w -name "*kernel_stats.csv") | head -1)
  n=bench_kernel_stats.csv; [ $w = cfg3 ] && n=bench_cfg3_kernel_stats.csv; [ $w = cfg5 ] && n=bench_cfg5_kernel_stats.csv; [ $w = backbone ] && n=backbone_kernel_stats.csv
  python tools/kernel_stats.py $f profiles/r5/$n > /dev/null
done
cp $O/bench_latest.json $O/bench_ops_level.json $O/bench_cfg4_shard.json $O/bench_force_collective.json $O/bench_cfg3.json $O/bench_cfg5.json \
   $O/bench_cfg5_fused_block_off.json $O/bench_cfg5_tail_io_waves_off.json profiles/r5/
cp $O/backbone.txt profiles/r5/backbone_blocks_on_off.txt
cp $O/time_block.txt profiles/r5/time_block.txt
cp $O/time_tail_io.txt profiles/r5/time_tail_io.txt
cp $O/time_conv1x1.txt profiles/r5/time_conv1x1.txt
python tools/pmc_summary.py $O/pmc_cfg2 profiles/r5 --videos 16 --workload cfg2 > /dev/null
( cd tools && python pmc_backbone.py ../$O/bb_pmc ../profiles/r5 --frames 72 --passes 4 --note "one-launch blocks, chunk 18" | tail -3
  python pmc_backbone.py ../$O/bb_pmc_chain ../profiles/r5 --frames 72 --passes 4 --note "--no-block (round-4 chain), chunk 18" --name pmc_hbm_traffic_backbone_chain.csv --no-json | tail -2 )
python - <<'PY'
import json
p = 'profiles/pmc_traffic.json'
d = json.load(open(p))
for k, v in d['sets'].items():
    if k.startswith('cfg5'):
        v['source'] = 'profiles/r5/pmc_hbm_traffic_backbone.csv'
    if k.startswith('cfg2'):
        v['source'] = 'profiles/r5/pmc_hbm_traffic.csv'
json.dump(d, open(p, 'w'), indent=1, sort_keys=True)
PY
cp temporal-span-proposal-network-vidvrd_amd/kernel_resources.json profiles/r5/kernel_resources.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('profiles/r5/bench_*.json')):
    d = json.load(open(f)); r = d.get('roofline') or {}
    print(f"{f[12:]:36s} {d['value']:11.0f} {d['ms_per_step']:8.2f} ms frac={r.get('frac') and round(r['frac'], 4)} kernel_ms={r.get('avg_launch_ms') and round(r['avg_launch_ms'], 2)} clock={r.get('shader_clock_mhz')}", d['config'].get('stage_ms', ''))
print(open('profiles/r5/backbone_blocks_on_off.txt').read())
print(open('profiles/r5/time_block.txt').read())
PY
