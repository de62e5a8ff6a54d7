#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FIRST=1 ONLY=1
for set in "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  set -- $set; n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmcn_$n -- python $R/tools/attn_quick.py > $OUT/pmcn_$n.log 2>&1
done
python $R/tools/pmc_traffic.py $OUT/attn32_pmc_summary.json $(find $OUT/pmcn_* -name '*counter_collection.csv') > $OUT/pmcn_agg.log 2>&1
