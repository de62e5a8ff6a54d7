#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s6
mkdir -p $OUT
cd $R
for P in 1 0; do
LGD_GEMM_PERSIST=$P TILES=33,34 SHAPES=plain,geglu FIRST=13 ROUNDS=3 timeout 400 python tools/gemm_ab.py > $OUT/ab_p$P.log 2>&1
echo "PERSIST=$P"; grep "^M" $OUT/ab_p$P.log | cut -c1-170
done
