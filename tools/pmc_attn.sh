#!/bin/bash
# GPU-box tool: SQ counters of the d = 40 self-attention forward, round-3 kernel (W4MODE=0) vs attn_w4 (W4MODE=2), separate
# --pmc passes with --kernel-trace only (the harness refuses counter collection combined with other trace domains)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in 0 2; do
  OUT=$R/gpurun_out/pmc_attn_w4mode$mode
  rm -rf $OUT; mkdir -p $OUT
  run() { n=$1; shift; W4MODE=$mode timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python $R/tools/attn_w4_abl.py > $OUT/$n.log 2>&1; }
  run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS
  run sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA
  run grbm GRBM_GUI_ACTIVE
  python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_attn_w4mode$mode.json $(find $OUT -name '*counter_collection.csv')
  rm -rf $OUT
done
