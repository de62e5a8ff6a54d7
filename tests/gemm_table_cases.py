"""Launch modes of the measured GEMM table (llm-groundeddiffusion_amd/tuning_gfx950.json).

The benchmark's GEMM launches are whatever the table says for their shape, so "the kernels the benchmark
runs" = the distinct (tile, taps, stride, ups, two-source, GEGLU, split-K) combinations in the table.
`cases()` returns one real table shape per combination; tests/test_bench_path_gpu.py launches each of
them against fp32 torch, and tests/test_tuning_table.py asserts (on CPU) that every table entry's
combination is among them — a re-tuned table can never select a code path the GPU suite does not run.
"""
import json
import os
import re
from collections import namedtuple

TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-groundeddiffusion_amd",
                     "tuning_gfx950.json")
KEY_RE = re.compile(r"M(\d+)_N(\d+)_K(\d+)_t(\d)_c(\d+)\+(\d+)_h(\d+)x(\d+)_s(\d)_u(\d)_e(\d)_b(\d+)$")

Shape = namedtuple("Shape", "key M N K taps c0 c1 hin hout stride ups geglu batches tile splits count")


def parse(key, entry):
    m = KEY_RE.match(key)
    if not m:
        raise ValueError(f"malformed tuning key {key!r}")
    M, N, K, taps, c0, c1, hin, hout, stride, ups, geglu, b = (int(g) for g in m.groups())
    return Shape(key, M, N, K, taps, c0, c1, hin, hout, stride, ups, geglu, b, int(entry["tile"]),
                 int(entry["splits"]), int(entry.get("count", 1)))


def mode(s: Shape):
    """What selects a code path inside lgd_gemm_f16: tile template, gather kind, sources, epilogue, split-K."""
    return (s.tile, s.taps, s.stride, s.ups, s.c1 > 0, s.geglu, s.splits > 1)


TABLE_LANES = TABLE.replace("tuning_gfx950.json", "tuning_gfx950_lanes.json")     # ops.TUNING_MODE == "throughput"


def table():
    """Entries of both tables; a shape both hold appears twice (key suffixed) when its launch configuration differs."""
    out = {k: parse(k, e) for k, e in json.load(open(TABLE)).items()}
    if os.path.exists(TABLE_LANES):
        for k, e in json.load(open(TABLE_LANES)).items():
            s = parse(k, e)
            if k not in out or (out[k].tile, out[k].splits) != (s.tile, s.splits):
                out[k + "@lanes"] = s._replace(key=k + "@lanes")
    return out


def cases():
    """One representative (the most frequently launched, then the largest) table shape per mode."""
    best = {}
    for s in table().values():
        m = mode(s)
        cur = best.get(m)
        if cur is None or (s.count, s.M * s.N * s.K) > (cur.count, cur.M * cur.N * cur.K):
            best[m] = s
    return [best[m] for m in sorted(best)]
