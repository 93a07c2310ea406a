#!/bin/bash
# LDS bank-conflict survey of every kernel of the library (one --pmc pass per driver script, kernel-trace only).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_all
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o $n -- "$@" > /dev/null 2> $OUT/$n.err
}
run bench python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
run analysis python3 tools/bench_analysis.py
run corrfunc python3 tools/bench_corrfunc.py
run pol python3 tools/bench_pol.py
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_all"
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if v["SQ_LDS_IDX_ACTIVE"] > 0 and "at::" not in k:
            # SQ_BUSY_CYCLES is summed over SEs/XCDs; LDS active is per CU summed: report ratio to wave cycles too
            print(os.path.basename(f)[:10], k, "conflict/active=%.2f" % (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]),
                  "lds_active=%.3g" % v["SQ_LDS_IDX_ACTIVE"], "wave_cycles=%.3g" % v["SQ_WAVE_CYCLES"], "busy=%.3g" % v["SQ_BUSY_CYCLES"])
PY
