#!/bin/bash
# Round profile: rocprofv3 kernel stats of the default bench + PMC passes (separate --pmc runs, kernel-trace only)
# for every kernel of the step: SQ pass, FETCH_SIZE pass, WRITE_SIZE pass.  tools/pmc_summary.py turns the CSVs into
# profiles/<tag>_pmc.json (per-kernel HBM bytes per launch, GB/s, MFMA instructions).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-delivered > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $n -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-delivered > /dev/null 2> $OUT/$n.err
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVES
run tcc2 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc3 WRITE_SIZE GRBM_GUI_ACTIVE
# cfg-5 rank share (one GPU doing the work of the most loaded of 8 ranks): kernel stats + the SQ pass, so that K4's
# fraction of the MFMA peak at that geometry is counter-backed (executed MFMA instructions, busy cycles)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o cfg5stats -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-host-delivered > $OUT/cfg5_bench_under_rocprof.json 2> $OUT/cfg5stats.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVES --output-format csv -d $OUT -o cfg5sq1 -- python3 bench.py --workload cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-host-delivered > /dev/null 2> $OUT/cfg5sq1.err
# the "next" rows: kernel stats of their benchmarks (program directly behind `--`: no launcher hop)
for nb in analysis pol flatsky corrfunc; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o next_$nb -- python3 tools/bench_$nb.py > $OUT/next_$nb.json 2> $OUT/next_$nb.err
done
python3 tools/pmc_summary.py $OUT $TAG > $OUT/${TAG}_pmc.json
cat $OUT/${TAG}_pmc.json | head -80
find $OUT -name "*.csv" | head -20
