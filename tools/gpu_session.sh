#!/bin/bash
# One gpurun session = one stage of this script (kept as ONE file; per-session scratch scripts are not committed).
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <stage> [args]'
R=${GRAFT_REPO_ROOT:-/root/repo}
STAGE=${1:-tests}; shift
OUT=$R/gpurun_out/$STAGE
rm -rf $OUT; mkdir -p $OUT
cd $R
case $STAGE in
  tests)      # full GPU suite + smoke
    timeout 900 python -m pytest tests -q -m gpu -x -rP "$@" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
    grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -n 3
    timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -n 1 $OUT/smoke.log ;;
  pick)       # selected tests: args = pytest selection
    timeout 900 python -m pytest -q -m gpu -x -rP "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
    grep -E "passed|failed|error|^\[" $OUT/pytest.log | tail -n 60 ;;
  bench)      # args = bench.py flags; writes bench.json
    timeout 900 python bench.py "$@" > $OUT/bench.log 2>&1; echo "bench rc=$?"
    grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json ;;
  tune)       # args: TOP [streams]; re-tunes the TOP most expensive GEMM shapes (latency table, or the lanes table with streams > 1)
    TOP=${1:-100}; STREAMS=${2:-1}
    if [ "$STREAMS" -gt 1 ]; then
      python - <<'PY'
import json
a = json.load(open("llm-groundeddiffusion_amd/tuning_gfx950.json")); a.update(json.load(open("llm-groundeddiffusion_amd/tuning_gfx950_lanes.json")))
json.dump(a, open("gpurun_out/tune/lanes_full.json", "w"), indent=0, sort_keys=True)
PY
      LGD_TUNE_STREAMS=$STREAMS LGD_TUNE_TOP=$TOP timeout ${TUNE_TIMEOUT:-480} python tools/tune_gemm.py sd14_gligen $OUT/lanes_full.json > $OUT/tune.log 2>&1
    else
      cp llm-groundeddiffusion_amd/tuning_gfx950.json $OUT/latency.json
      LGD_TUNE_TOP=$TOP timeout ${TUNE_TIMEOUT:-480} python tools/tune_gemm.py sd14_gligen $OUT/latency.json > $OUT/tune.log 2>&1
    fi
    echo "tune rc=$?"; tail -n 4 $OUT/tune.log ;;
  tune_only)  # args: TILES TOP [streams]; tries only the given tile codes (e.g. "46,47") against the entries the table holds,
              # on the TOP most expensive shapes; the candidate table lands in $OUT (latency.json | lanes_full.json + lanes_new.json)
    TL=${1:-46,47}; TOP=${2:-150}; STREAMS=${3:-1}
    if [ "$STREAMS" -gt 1 ]; then
      python - <<PY
import json
a = json.load(open("llm-groundeddiffusion_amd/tuning_gfx950.json")); a.update(json.load(open("llm-groundeddiffusion_amd/tuning_gfx950_lanes.json")))
json.dump(a, open("gpurun_out/tune_only/lanes_full.json", "w"), indent=0, sort_keys=True)
PY
      LGD_TUNE_ONLY_TILES=$TL LGD_TUNE_STREAMS=$STREAMS LGD_TUNE_TOP=$TOP timeout ${TUNE_TIMEOUT:-700} python tools/tune_gemm.py sd14_gligen $OUT/lanes_full.json > $OUT/tune.log 2>&1
      echo "tune rc=$?"; tail -n 3 $OUT/tune.log
      python - <<PY
import json
lat = json.load(open("llm-groundeddiffusion_amd/tuning_gfx950.json"))
full = json.load(open("gpurun_out/tune_only/lanes_full.json"))
new = {k: v for k, v in full.items() if k not in lat or (lat[k]["tile"], lat[k]["splits"]) != (v["tile"], v["splits"])}
json.dump(new, open("gpurun_out/tune_only/lanes_new.json", "w"), indent=0, sort_keys=True)
print("lanes table:", len(new), "entries")
PY
    else
      cp llm-groundeddiffusion_amd/tuning_gfx950.json $OUT/latency.json
      LGD_TUNE_ONLY_TILES=$TL LGD_TUNE_TOP=$TOP timeout ${TUNE_TIMEOUT:-700} python tools/tune_gemm.py sd14_gligen $OUT/latency.json > $OUT/tune.log 2>&1
      echo "tune rc=$?"; tail -n 3 $OUT/tune.log
    fi
    grep -c "tile 4[67]" $OUT/tune.log ;;
  tune_big)   # round 5: the GEMM shapes of the 16- / 32-image main plans and 8- / 16-image guidance plans (bigger UNet
              # calls, bench.py --group / --max-batch) added to the shared-GPU table; args: [streams]
    STREAMS=${1:-2}
    python - <<'PY'
import json
a = json.load(open("llm-groundeddiffusion_amd/tuning_gfx950.json")); a.update(json.load(open("llm-groundeddiffusion_amd/tuning_gfx950_lanes.json")))
json.dump(a, open("gpurun_out/tune_big/lanes_full.json", "w"), indent=0, sort_keys=True)
PY
    LGD_TUNE_BATCHES=${TUNE_MAIN:-16,32} LGD_TUNE_GUIDE_BATCHES=${TUNE_GUIDE:-8,16} LGD_TUNE_STREAMS=$STREAMS \
      timeout ${TUNE_TIMEOUT:-560} python tools/tune_gemm.py sd14_gligen $OUT/lanes_full.json > $OUT/tune.log 2>&1
    echo "tune rc=$?"; grep -c "TF/s" $OUT/tune.log; tail -n 3 $OUT/tune.log ;;
  sweep)      # round 5: (lanes x steps per lane job x images per UNet call) on the default workload; args = bench flags
    for cfg in "4 1 8 4" "2 2 16 8" "4 2 16 8" "1 4 32 16" "2 4 32 16"; do
      set -- $cfg
      timeout 600 python bench.py --steps ${SWEEP_STEPS:-8} --warmup 2 --no-cpu-baseline --no-roofline --lanes $1 --group $2 \
        --max-batch $3 --max-batch-guided $4 > $OUT/l$1_g$2_b$3.log 2>&1
      echo "lanes=$1 group=$2 max_batch=$3/$4: $(grep '^{' $OUT/l$1_g$2_b$3.log | tail -1 | cut -c1-130)"
      grep '^{' $OUT/l$1_g$2_b$3.log | tail -1 > $OUT/l$1_g$2_b$3.json
    done ;;
  retune_ab)  # re-tune the TOP most expensive shapes of the shared-GPU table (4 streams) and A/B it against the committed
              # table with the driver's command, alternating arms on this box; the candidate table lands in $OUT
    TOP=${1:-120}
    cp llm-groundeddiffusion_amd/tuning_gfx950_lanes.json $OUT/lanes_old.json
    python - <<'PY'
import json
a = json.load(open("llm-groundeddiffusion_amd/tuning_gfx950.json")); a.update(json.load(open("llm-groundeddiffusion_amd/tuning_gfx950_lanes.json")))
json.dump(a, open("gpurun_out/retune_ab/lanes_full.json", "w"), indent=0, sort_keys=True)
PY
    arm() { timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/$1.log 2>&1; echo "$1: $(grep '^{' $OUT/$1.log | tail -1 | cut -c1-105)"; }
    arm A1
    LGD_TUNE_STREAMS=4 LGD_TUNE_TOP=$TOP timeout ${TUNE_TIMEOUT:-480} python tools/tune_gemm.py sd14_gligen $OUT/lanes_full.json > $OUT/tune.log 2>&1
    echo "tune rc=$?"; tail -n 2 $OUT/tune.log
    python - <<'PY'
import json
lat = json.load(open("llm-groundeddiffusion_amd/tuning_gfx950.json"))
old = json.load(open("gpurun_out/retune_ab/lanes_old.json"))
full = json.load(open("gpurun_out/retune_ab/lanes_full.json"))
new = {k: v for k, v in full.items() if k not in lat or (lat[k]["tile"], lat[k]["splits"]) != (v["tile"], v["splits"])}
for k, v in old.items():                     # entries the tuner did not visit stay as they were
    if k not in new and (k not in lat or (lat[k]["tile"], lat[k]["splits"]) != (v["tile"], v["splits"])) and full.get(k, v) == v:
        new[k] = v
json.dump(new, open("gpurun_out/retune_ab/lanes_new.json", "w"), indent=0, sort_keys=True)
ch = sum(1 for k in new if k not in old or (old[k]["tile"], old[k]["splits"]) != (new[k]["tile"], new[k]["splits"]))
print("new lanes table:", len(new), "entries,", ch, "differ from the committed one;", sum(1 for k in old if k not in new), "dropped")
PY
    cp $OUT/lanes_new.json llm-groundeddiffusion_amd/tuning_gfx950_lanes.json; arm B1
    cp $OUT/lanes_old.json llm-groundeddiffusion_amd/tuning_gfx950_lanes.json; arm A2
    cp $OUT/lanes_new.json llm-groundeddiffusion_amd/tuning_gfx950_lanes.json; arm B2
    cp $OUT/lanes_old.json llm-groundeddiffusion_amd/tuning_gfx950_lanes.json ;;
  final)      # end-of-round validation: full GPU suite + smoke, then the bench lines DESIGN.md / README.md quote (tag = $1)
    T=${1:-r05}
    timeout 1500 python -m pytest tests -q -m gpu -rP > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
    grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -n 3
    timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -n 1 $OUT/smoke.log
    line() { # name, bench args...
      n=$1; shift
      timeout 900 python bench.py "$@" > $OUT/$n.log 2>&1
      grep '^{' $OUT/$n.log | tail -1 > $OUT/${T}_${n}_bench_line.json; echo "$n: $(cut -c1-150 $OUT/${T}_${n}_bench_line.json)"
    }
    line bench_driver_command --gpus 1 --steps 20 --warmup 5
    line bench_default
    line lmd --workload lmd --steps 8 --warmup 2
    line lmd_v0.1_100prompts --workload lmd_v0.1 --prompts 100 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline
    line lmd_v0.1_400prompts --workload lmd_v0.1 --prompts 400 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline
    line backward_guidance_sd21 --workload backward_guidance --steps 4 --warmup 1 --no-cpu-baseline
    line sdxl_refiner --workload sdxl_refiner --steps 2 --warmup 1 --no-cpu-baseline ;;
  baseline)   # args: TAG; GPU suite + the driver's command + one launch sequence alone with the per-shape profile
    T=${1:-r06}
    timeout 1500 python -m pytest tests -q -m gpu -rP > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
    grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -n 3
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver.log 2>&1
    grep '^{' $OUT/driver.log | tail -1 > $OUT/${T}_bench_driver_command_bench_line.json; echo "driver: $(cut -c1-150 $OUT/${T}_bench_driver_command_bench_line.json)"
    timeout 600 python bench.py --lanes 1 --steps 4 --warmup 1 --no-cpu-baseline --shape-profile $OUT/${T}_shape_profile.json > $OUT/lanes1.log 2>&1
    grep '^{' $OUT/lanes1.log | tail -1 > $OUT/${T}_bench_lanes1_bench_line.json; echo "lanes1: $(cut -c1-150 $OUT/${T}_bench_lanes1_bench_line.json)" ;;
  phase_abl)  # ablation of the phase-split GEMM tiles (46, 47) on big shapes: builds the -DLGD_GEMM_ABLATION library on the box
    python llm-groundeddiffusion_amd/build.py --force > $OUT/build.log 2>&1
    bash tools/build_abl.sh >> $OUT/build.log 2>&1; tail -n 1 $OUT/build.log
    for A in ${ABLS:-0 1 2 3 4 16 32 64}; do
      echo "== ABL $A" >> $OUT/abl.log
      LGD_GEMM_ABL=$A TILES=${TILES:-46} SHAPES=sq,geglu ROUNDS=3 timeout 200 python tools/gemm_ab.py 2>&1 | grep "^M" | cut -c1-90 >> $OUT/abl.log
      LGD_GEMM_ABL=$A TILES=${TILES2:-47:2} SHAPES=conv FIRST=2 ROUNDS=3 timeout 200 python tools/gemm_ab.py 2>&1 | grep "^M" | cut -c1-90 >> $OUT/abl.log
    done
    cat $OUT/abl.log
    python llm-groundeddiffusion_amd/build.py --force > /dev/null 2>&1 ;;
  evidence)   # args: TAG; rocprofv3 kernel-trace stats of the bench lines DESIGN.md quotes, the HBM-traffic PMC passes of the default
              # command, and SQ / TCC / HBM counters of the phase-split GEMM tiles on the benchmark's heaviest shapes (one shape per pass)
    T=${1:-r06}
    cd /tmp && export TMPDIR=/tmp
    prof() { tag=$1; shift
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python $R/bench.py "$@" > $OUT/$tag.log 2>&1
      grep '^{' $OUT/$tag.log | tail -1 > $OUT/${tag}_bench_line.json
      f=$(find $OUT/$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
      rm -rf $OUT/$tag; cut -c1-160 $OUT/${tag}_bench_line.json; }
    prof ${T}_bench_lanes1 --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline
    prof ${T}_bench_driver --steps 4 --warmup 1 --no-cpu-baseline --no-roofline
    TAG=evidence/traffic bash $R/tools/bench_traffic.sh > $OUT/traffic.log 2>&1
    cp $OUT/traffic/bench_traffic_pmc.json $OUT/${T}_bench_traffic_pmc.json 2>/dev/null
    rm -rf $OUT/traffic/FETCH_SIZE $OUT/traffic/WRITE_SIZE
    i=0; dirs=""
    for spec in "47:65536,320,2880,9,320,64,0" "46:65536,2560,320,1,320,0,1" "46:16384,5120,640,1,640,0,1" "46:4096,10240,1280,1,1280,0,1" "47:65536,320,320,1,320,0,0" "46:8192,8192,4096,1,4096,0,0"; do
      tile=${spec%%:*}; shp=${spec#*:}; d=$OUT/shape$i; mkdir -p $d; echo $shp > $d/shape.txt; echo $tile > $d/tile.txt
      run() { n=$1; shift; SHAPE=$shp TILES=$tile ROUNDS=1 REPS=3 timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d/$n -- python $R/tools/gemm_ab.py > $d/$n.log 2>&1; }
      run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS
      run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
      run tcc1 TCC_HIT_sum TCC_MISS_sum
      run tcc2 FETCH_SIZE
      run tcc3 WRITE_SIZE
      dirs="$dirs $d"; i=$((i+1))
    done
    python $R/tools/pmc_shapes.py $OUT/${T}_gemm_pmc_summary.json $dirs
    find $OUT -name '*.csv' -path '*shape*' -delete 2>/dev/null
    ls $OUT ;;
  pmc_shapes) # args: OUT.json SPEC... (SPEC = tile:M,N,K,taps,cin,h,geglu): the per-shape PMC passes of the evidence stage on
              # further shapes, merged into an existing summary (profiles/r06_gemm_pmc_summary.json)
    J=$1; shift
    cd /tmp && export TMPDIR=/tmp
    i=0; dirs=""
    for spec in "$@"; do
      tile=${spec%%:*}; shp=${spec#*:}; d=$OUT/shape$i; mkdir -p $d; echo $shp > $d/shape.txt; echo $tile > $d/tile.txt
      run() { n=$1; shift; SHAPE=$shp TILES=$tile ROUNDS=1 REPS=3 timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d/$n -- python $R/tools/gemm_ab.py > $d/$n.log 2>&1; }
      run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS
      run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
      run tcc1 TCC_HIT_sum TCC_MISS_sum
      run tcc2 FETCH_SIZE
      run tcc3 WRITE_SIZE
      dirs="$dirs $d"; i=$((i+1))
    done
    python $R/tools/pmc_shapes.py $OUT/new.json $dirs
    python - <<PY
import json
old = json.load(open("$R/$J")); new = json.load(open("$OUT/new.json"))
old["shapes"].update(new["shapes"])
json.dump(old, open("$OUT/merged.json", "w"), indent=1)
print("merged:", list(old["shapes"]))
PY
    find $OUT -name '*.csv' -delete 2>/dev/null ;;
  tune_vae)   # the VAE decoder's GEMM shapes (decode batches 1, 2) added to the latency table; the untuned shapes only
    cp llm-groundeddiffusion_amd/tuning_gfx950.json $OUT/latency.json
    LGD_TUNE_VAE=${1:-1,2} timeout ${TUNE_TIMEOUT:-900} python tools/tune_gemm.py sd14_gligen $OUT/latency.json > $OUT/tune.log 2>&1
    echo "tune rc=$?"; grep -c "TF/s" $OUT/tune.log; grep "VAE decode\|sum over\|REJECTED" $OUT/tune.log; tail -n 30 $OUT/tune.log | cut -c1-160
    for f in "" "--no-decode"; do python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline $f 2>&1 | grep "^{" | cut -c1-120; done
    cp $OUT/latency.json llm-groundeddiffusion_amd/tuning_gfx950.json
    python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>&1 | grep "^{" | cut -c1-120 ;;
  *) echo "unknown stage $STAGE"; exit 2 ;;
esac
