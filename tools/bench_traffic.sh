#!/bin/bash
# GPU-box tool: HBM traffic per kernel launch of the BENCHMARK's own command (short: 2 DDIM steps, same batch
# shapes as the default run), from rocprofv3 PMC passes — FETCH_SIZE and WRITE_SIZE in separate passes with
# --kernel-trace only, as MI355X_MICROARCH.md prescribes — plus the kernel-trace stats of the same command.
# Writes profiles-ready files under gpurun_out/$TAG/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r02_bench_traffic}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --num-inference-steps 2 --layouts 4 --no-cpu-baseline --no-roofline --no-decode"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- $CMD > $OUT/$c.log 2>&1
done
python $R/tools/pmc_traffic.py $OUT/per_kernel.json $(find $OUT -name '*counter_collection.csv') > $OUT/agg.log 2>&1
python - <<PY
import json, re
j = json.load(open("$OUT/per_kernel.json"))
def norm(k):
    k = k.replace(" ", "")
    m = re.match(r"(gemm_pipe_kernel)<(\d+),(\d+),(\d+),(\d+),(\d+),", k)
    if m: return "%s<%s,%s,%s,%s,%s>" % m.groups()
    m = re.match(r"(gemm_phase_kernel)<(\d+),(\d+),", k)
    if m: return "%s<%s,%s>" % m.groups()
    m = re.match(r"(gemm_dma_kernel)<(\d+),(\d+),(\d+)>", k)
    if m: return "%s<%s,%s,%s>" % m.groups()
    return k
out = {}
for k, v in j.items():
    n = norm(k)
    rd, wr, ln = v.get("hbm_read_bytes_per_launch"), v.get("hbm_write_bytes_per_launch"), v["launches"]
    if rd is None or wr is None: continue
    e = out.setdefault(n, dict(read=0.0, write=0.0, launches=0))
    e["read"] += rd * ln; e["write"] += wr * ln; e["launches"] += ln
res = dict(method="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
                  "'bench.py --steps 1 --warmup 0 --num-inference-steps 2 --layouts 4 --no-decode' (the default run's batch "
                  "shapes); FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE as reported; averages over "
                  "all launches of a kernel",
           kernels={n: dict(hbm_bytes_per_launch=round((e["read"] + e["write"]) / e["launches"]),
                            hbm_read_bytes_per_launch=round(e["read"] / e["launches"]),
                            hbm_write_bytes_per_launch=round(e["write"] / e["launches"]), launches=e["launches"])
                    for n, e in sorted(out.items())})
json.dump(res, open("$OUT/bench_traffic_pmc.json", "w"), indent=1)
print("kernels:", len(res["kernels"]))
PY
