#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s10
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -rP > $OUT/pytest.log 2>&1
grep "passed\|failed" $OUT/pytest.log | tail -2; grep "^E  " $OUT/pytest.log | head -10
for P in 1 0; do
LGD_SPLITK_IN_LAUNCH=$P TILES=0 SHAPES=small ROUNDS=3 timeout 300 python tools/gemm_ab.py > $OUT/small_p$P.log 2>&1
echo "in-launch=$P"; grep "^M" $OUT/small_p$P.log | cut -c1-120
done
timeout 600 python bench.py --steps 2 --warmup 1 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
LGD_SPLITK_IN_LAUNCH=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_2launch.log 2>&1
tail -1 $OUT/bench_2launch.log | cut -c1-200
