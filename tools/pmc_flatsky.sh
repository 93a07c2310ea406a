#!/bin/bash
# PMC passes for the flat-sky line-FFT kernel, separate --pmc runs with kernel-trace only.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fs
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "linefft_kernel" --output-format csv -d $OUT -o $n -- python3 tools/flatsky_probe.py > /dev/null 2> $OUT/$n.err
}
run lds1 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES
run val1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES
run mem1 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
ls $OUT
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_fs"
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40] + "|" + r.get("Grid_Size", "")][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(os.path.basename(f), k, dict(v))
PY
