#!/bin/bash
# Instruction-mix survey of the bench kernels (one --pmc pass, kernel-trace only).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o mix -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/mix.err
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_valu"
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        if "at::" in k or "rocclr" in k: continue
        n = len(nd[k])
        print(k, "n=%d" % n, {a: "%.3g" % (b / n) for a, b in v.items()})
PY
