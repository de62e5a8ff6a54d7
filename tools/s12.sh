#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s12
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -rP > $OUT/pytest.log 2>&1
grep "passed\|failed" $OUT/pytest.log | tail -1
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
export FIRST=1 ONLY=0
for set in "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "tcc FETCH_SIZE" "tccw WRITE_SIZE"; do
  set -- $set; n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$n -- python $R/tools/attn_quick.py > $OUT/pmc_$n.log 2>&1
done
python $R/tools/pmc_traffic.py $OUT/attn16_pmc_summary.json $(find $OUT/pmc_* -name '*counter_collection.csv') > $OUT/pmc_agg.log 2>&1
rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_tcc $OUT/pmc_tccw
