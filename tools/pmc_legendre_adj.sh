#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_adj
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVES --kernel-include-regex "legendre_adj_kernel" --output-format csv -d $OUT -o sq1 -- python3 tools/bench_analysis.py > /dev/null 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --kernel-include-regex "legendre_adj_kernel" --output-format csv -d $OUT -o sq2 -- python3 tools/bench_analysis.py > /dev/null 2> $OUT/sq2.err
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_adj"
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    acc = collections.defaultdict(float); disp=set()
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n=len(disp)
    print(os.path.basename(f), "launches", n, {k: "%.4g" % (v/n) for k, v in acc.items()})
PY
