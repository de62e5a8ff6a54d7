#!/bin/bash
# GPU-box tool: effective shader clock of a GEMM variant = GRBM_GUI_ACTIVE / kernel duration (one --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG:-pmc_clock}
mkdir -p $OUT
export ROUNDS=1 REPS=3
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/run -- python $R/tools/gemm_ab.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
cc = glob.glob("$OUT/run/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("$OUT/run/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(r["Dispatch_Id"])
    if not d or "gemm" not in d[1]: continue
    agg[(d[1].split("(")[0][-60:], d[2])].append((float(r["Counter_Value"]), d[0]))
for k, v in agg.items():
    c = sum(x[0] for x in v) / len(v); t = sum(x[1] for x in v) / len(v)
    print(f"{k[0]:62s} grid {k[1]:>8s}  {t/1e3:8.1f} us  GUI_ACTIVE {c:12.0f}  -> {c/t:6.3f} GHz (x#SE?)")
PY
