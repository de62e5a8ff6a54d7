#!/bin/bash
# GPU-box tool: SQ / LDS / TCC counters of the GEMM kernels on one shape (separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG:-pmc_gemm}
rm -rf $OUT; mkdir -p $OUT
export TILES=${TILES:-22,33} SHAPES=${SHAPES:-conv} ROUNDS=1 REPS=3
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python $R/tools/gemm_ab.py > $OUT/$n.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS
run sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_UNALIGNED_STALL
run tcc1 TCC_HIT_sum TCC_MISS_sum
run tcc2 FETCH_SIZE
python $R/tools/pmc_traffic.py $OUT/summary.json $(find $OUT -name '*counter_collection.csv')
