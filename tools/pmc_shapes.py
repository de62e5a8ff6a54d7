"""Builds profiles/r06_gemm_pmc_summary.json from per-shape PMC passes (tools/gpu_session.sh evidence): one entry per
benchmark shape with the kernel's duration, MFMA-pipe busy share, wait share, scalar / vector issue counts, L2 hit rate
and HBM bytes against the shape's ALGORITHMIC bytes (operands read once + output written once [+ residual]).

    python tools/pmc_shapes.py out.json dir_of_shape_1 dir_of_shape_2 ...     (each dir: shape.txt + */*counter_collection.csv + *kernel_trace.csv)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for d in sys.argv[2:]:
    M, N, K, taps, cin, h, geglu = (int(x) for x in open(os.path.join(d, "shape.txt")).read().split(","))
    tile = open(os.path.join(d, "tile.txt")).read().strip()
    cnt, num = defaultdict(float), defaultdict(int)
    dur, ndur = 0.0, 0
    kname = None
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "gemm_" not in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]:
                continue
            kname = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
            cnt[r["Counter_Name"]] += float(r["Counter_Value"])
            num[r["Counter_Name"]] += 1
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "gemm_" in r["Kernel_Name"] and "splitk" not in r["Kernel_Name"]:
                dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                ndur += 1
    c = {k: v / num[k] for k, v in cnt.items()}
    n_out = N // 2 if geglu else N
    a_bytes = M * (cin if taps == 9 else K) * 2              # the activation map is read once (nine taps re-read it from cache)
    alg = a_bytes + N * K * 2 + M * n_out * 2 + (0 if geglu else M * n_out * 2)     # + weights + output (+ the fp16 residual gemm_ab adds)
    e = dict(kernel=kname, tile=tile, M=M, N=N, K=K, taps=taps, geglu=bool(geglu),
             us_under_pmc=round(dur / max(ndur, 1) / 1e3, 1), algorithmic_bytes=alg)
    if "SQ_WAVE_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES quad-cycles summed over waves; two waves per SIMD (512-thread
        # workgroups, one per CU): the SIMD's time = 4 x WAVE_CYCLES / 2  (same formula as profiles/r05_gemm_256x128_pmc_summary.json)
        e["mfma_busy_share"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (2.0 * c["SQ_WAVE_CYCLES"]), 3)
    for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA",
              "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F16",
              "GRBM_GUI_ACTIVE"):
        if k in c:
            e[k] = c[k]
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        e["wait_share_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 3)
    if "FETCH_SIZE" in c:
        e["hbm_read_bytes"] = round(c["FETCH_SIZE"] * 1024 * 2)      # gfx950: 128-B requests tallied at 64 B (MI355X_MICROARCH.md)
    if "WRITE_SIZE" in c:
        e["hbm_write_bytes"] = round(c["WRITE_SIZE"] * 1024)
    if "hbm_read_bytes" in e:
        e["traffic"] = e["hbm_read_bytes"] + e.get("hbm_write_bytes", 0)
        e["traffic_ratio"] = round(e["traffic"] / alg, 2)
        e["read_ratio"] = round(e["hbm_read_bytes"] / (alg - M * n_out * 2), 2)
    key = f"M{M}_N{N}_K{K}_t{taps}" + ("_geglu" if geglu else "")
    res[key] = e
json.dump(dict(method="per shape: tools/gemm_ab.py on ONE shape and ONE tile under rocprofv3 --pmc (separate passes per counter set, "
                      "--kernel-trace only); FETCH_SIZE x2 (gfx950), WRITE_SIZE as reported; algorithmic bytes = activation map + "
                      "weights read once, output written once, fp16 residual read once (the tool's plain / conv cases add one)",
               shapes=res), open(out, "w"), indent=1)
print("wrote", out, list(res))
