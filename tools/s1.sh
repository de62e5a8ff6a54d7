#!/bin/bash
# GPU-box session 1 of round 3: instruction-rate microbenchmarks, the GPU test suite with its printed parity
# measurements, SQ counters of the current self-attention kernel, GEMM numbers of the short-K shapes.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s1
mkdir -p $OUT
cd $R
timeout 120 ./llm-groundeddiffusion_amd/build/ubench > $OUT/ubench.log 2>&1
timeout 600 python -m pytest tests -m gpu -q -rP -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 200 python tools/attn_quick.py > $OUT/attn_quick.log 2>&1
TILES=0 SHAPES=geglu,plain ROUNDS=3 timeout 300 python tools/gemm_ab.py > $OUT/gemm_shortk.log 2>&1
cd /tmp && export TMPDIR=/tmp
export FIRST=1
for set in "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "tcc FETCH_SIZE" "tccw WRITE_SIZE"; do
  set -- $set; n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$n -- python $R/tools/attn_quick.py > $OUT/pmc_$n.log 2>&1
done
python $R/tools/pmc_traffic.py $OUT/attn_pmc_summary.json $(find $OUT -name '*counter_collection.csv') > $OUT/pmc_agg.log 2>&1
