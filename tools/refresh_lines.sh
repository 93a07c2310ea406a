#!/bin/bash
# Regenerates the bench lines the docs quote (GPU box, repo root): default cfg-3 line with both baselines, emulated
# shards (every rank timed in turn; contiguous and folded channel assignment), cfg 4 in both sum modes, the cfg-5 rank
# shares.  Since round 5 every line carries the seeded legs (numpy's PCG64 / legacy streams generated in l ranges).
# usage: bash tools/refresh_lines.sh <tag>   -> gpurun_out/lines_<tag>/
TAG=${1:-r05}
OUT=gpurun_out/lines_$TAG
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-host-delivered"
python3 bench.py > $OUT/bench_cfg3_${TAG}_with_baselines.json 2> $OUT/default.err || exit 1
for n in 2 4 8; do $B --emulate-shard $n > $OUT/bench_cfg3_emulated_shard${n}_${TAG}.json 2>> $OUT/shard.err || exit 1; done
$B --emulate-shard 8 --fold > $OUT/bench_cfg3_emulated_shard8_fold_${TAG}.json 2>> $OUT/shard.err || exit 1
$B --workload cfg4 --sum-mode joint --steps 4 --warmup 1 > $OUT/bench_cfg4_${TAG}_joint.json 2>> $OUT/cfg4.err || exit 1
$B --workload cfg4 --sum-mode separate --steps 4 --warmup 1 > $OUT/bench_cfg4_${TAG}_separate.json 2>> $OUT/cfg4.err || exit 1
$B --workload cfg5 --steps 3 --warmup 1 > $OUT/bench_cfg5_emulated_shard_${TAG}.json 2>> $OUT/cfg5.err || exit 1
$B --workload cfg5 --steps 3 --warmup 1 --fold > $OUT/bench_cfg5_emulated_shard_fold_${TAG}.json 2>> $OUT/cfg5.err || exit 1
for f in $OUT/*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); c=d['config']; s=c.get('seeded_numpy_mode') or {}; l=c.get('legacy_rng_mode') or {}; e=c.get('emulated_ranks') or {}
print('$f'.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],2), d.get('stages_ms'), round(d['roofline']['frac'],4), 'seeded', round(s.get('ms_per_step',0),2), 'legacy', round(l.get('ms_per_step',0),2), 'per-rank', e.get('per_rank_ms'))"; done
