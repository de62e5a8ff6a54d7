#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s23
mkdir -p $OUT
cd $R
LGD_TUNE_FRESH=1 timeout 600 python tools/tune_gemm.py sdxl_refiner $OUT/tuning_sdxl.json > $OUT/tune.log 2>&1 || true
tail -4 $OUT/tune.log
python - <<'PY'
import json,os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
p=f"{R}/llm-groundeddiffusion_amd/tuning_gfx950.json"; t=json.load(open(p))
n=json.load(open(f"{R}/gpurun_out/s23/tuning_sdxl.json")) if os.path.exists(f"{R}/gpurun_out/s23/tuning_sdxl.json") else {}
new={k:v for k,v in n.items() if k not in t}
t.update(new); json.dump(t,open(p,"w"),indent=0,sort_keys=True); print("merged",len(new),"new shapes")
PY
timeout 400 python bench.py --workload sdxl_refiner --steps 2 --warmup 1 --lanes 2 --no-cpu-baseline > $OUT/refiner_tuned.log 2>&1; tail -n 1 $OUT/refiner_tuned.log | cut -c1-300
timeout 400 python bench.py --workload sdxl_refiner --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-roofline > $OUT/refiner_tuned_l1.log 2>&1; tail -n 1 $OUT/refiner_tuned_l1.log | cut -c1-200
