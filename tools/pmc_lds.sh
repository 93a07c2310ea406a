#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lds
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT -o lds -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-delivered --no-seeded-modes > /dev/null 2> $OUT/lds.err
python3 - <<PY
import csv, collections
f="$OUT/lds_counter_collection.csv"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][-48:]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    if v.get("SQ_LDS_IDX_ACTIVE",0)>0:
        print("%-50s conflict/active %.3f  lds insts %.3g  active_inst_lds/wave_cycles %.3f" % (k, v.get("SQ_LDS_BANK_CONFLICT",0)/v["SQ_LDS_IDX_ACTIVE"], v.get("SQ_INSTS_LDS",0), v.get("SQ_ACTIVE_INST_LDS",0)/max(v.get("SQ_WAVE_CYCLES",1),1)))
PY
