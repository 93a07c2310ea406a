#!/bin/bash
# PMC passes for the K4 legendre kernel (separate --pmc runs, no tracing domains besides kernel-trace)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
run() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KREGEX" --output-format csv -d $OUT -o $n -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/$n.err
}
KREGEX=${KREGEX:-legendre}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES
run tcc1 TCC_HIT TCC_MISS GRBM_GUI_ACTIVE
run tcc2 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc3 WRITE_SIZE GRBM_GUI_ACTIVE
ls $OUT
