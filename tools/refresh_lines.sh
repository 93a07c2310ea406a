#!/bin/bash
# Regenerates the bench lines the docs quote (GPU box, repo root): default cfg-3 line with both baselines, emulated
# shards, cfg 4 in both sum modes, the cfg-5 rank share.  usage: bash tools/refresh_lines.sh <tag>   -> gpurun_out/lines_<tag>/
TAG=${1:-r04}
OUT=gpurun_out/lines_$TAG
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-host-delivered --no-seeded-modes"
python3 bench.py > $OUT/bench_cfg3_${TAG}_with_baselines.json 2> $OUT/default.err || exit 1
for n in 2 4 8; do $B --emulate-shard $n > $OUT/bench_cfg3_emulated_shard${n}_${TAG}.json 2>> $OUT/shard.err || exit 1; done
$B --workload cfg4 --sum-mode joint --steps 4 --warmup 1 > $OUT/bench_cfg4_${TAG}_joint.json 2>> $OUT/cfg4.err || exit 1
$B --workload cfg4 --sum-mode separate --steps 4 --warmup 1 > $OUT/bench_cfg4_${TAG}_separate.json 2>> $OUT/cfg4.err || exit 1
$B --workload cfg5 --steps 3 --warmup 1 > $OUT/bench_cfg5_emulated_shard_${TAG}.json 2>> $OUT/cfg5.err || exit 1
for f in $OUT/*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],2), d.get('stages_ms'), round(d['roofline']['frac'],4))"; done
