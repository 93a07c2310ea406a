#!/bin/bash
# PMC pass for the K5 ring-FFT kernels (LDS behaviour), separate --pmc run with kernel-trace only.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_k5
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "ringfft_kernel" --output-format csv -d $OUT -o $n -- python3 tools/k5_probe.py > /dev/null 2> $OUT/$n.err
}
run lds1 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES
run val1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES
ls $OUT
