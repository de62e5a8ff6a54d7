"""Host logic of the GEMM launch configuration: the measured table only holds tile codes the library
accepts, split-K factors keep whole K tiles, and the fallback heuristic always returns a legal code."""
import json
import os
import re

import lgd_amd  # noqa: F401
from lgd_amd import ops

TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-groundeddiffusion_amd",
                     "tuning_gfx950.json")
PIPE = {33: 160, 34: 128, 35: 64, 37: 160, 38: 128, 39: 64, 40: 160, 41: 128, 42: 64, 44: 256, 45: 128,      # code -> tile width
        46: 256, 47: 320}
TWO_STAGE = {44, 45}                                  # plain single-source contractions only; 44 (256 x 256): one split
PHASE = {46, 47}                                      # round 6: single source; plain or 3x3 stride-1 same-size convolution
LEGAL = (set(range(1, 11)) | {16 + t for t in range(1, 11)} | set(PIPE)) - {8, 24}
TILE_BN = {1: 128, 2: 64, 3: 128, 4: 64, 5: 128, 6: 160, 7: 160, 9: 320, 10: 128}


def test_table_entries_are_legal():
    table = json.load(open(TABLE))
    assert len(table) >= 200
    lanes = TABLE.replace("tuning_gfx950.json", "tuning_gfx950_lanes.json")
    entries = list(table.items()) + (list(json.load(open(lanes)).items()) if os.path.exists(lanes) else [])
    for key, e in entries:
        m = re.match(r"M(\d+)_N(\d+)_K(\d+)_t(\d)_c(\d+)\+(\d+)_h(\d+)x(\d+)_s(\d)_u(\d)_e(\d)_b(\d+)$", key)
        assert m, key
        M, N, K, taps, c0, c1 = (int(m.group(i)) for i in range(1, 7))
        geglu = int(m.group(11))
        assert e["tile"] in LEGAL, (key, e)
        assert K == taps * (c0 + c1)
        assert 1 <= e["splits"] <= 16 and (e["splits"] == 1 or K // 64 >= e["splits"])
        if geglu:                                                          # 160/320-wide tiles cannot pair value|gate rows
            assert PIPE.get(e["tile"], 0) not in (160, 320) and (e["tile"] in PIPE or (e["tile"] & 15) not in (6, 7, 9)), (key, e)
        if e["tile"] in PIPE:
            assert K % 64 == 0 and (c0 + c1) % 64 == 0 and c0 % 64 == 0, (key, e)
        if e["tile"] in PHASE:
            hin, hout, stride, ups = int(m.group(7)), int(m.group(8)), int(m.group(9)), int(m.group(10))
            assert c1 == 0 and int(m.group(12)) == 1 and (taps == 1 or (stride == 1 and ups == 0 and hin == hout)), (key, e)
        if e["tile"] in TWO_STAGE:
            assert taps == 1 and c1 == 0 and (e["tile"] != 44 or (e["splits"] == 1 and N % 8 == 0)), (key, e)
        assert e["us"] > 0 and e["tflops"] > 0


def test_heuristic_returns_legal_codes():
    for M in (1, 30, 64, 256, 1024, 4100, 16384, 65536):
        for N in (64, 320, 640, 960, 1280, 2560, 5120, 10240):
            for geglu in (False, True):
                if geglu and N % 32:
                    continue
                for batches in (1, 8):
                    t = ops.choose_tile(M, N, batches, geglu, 640)
                    assert t in LEGAL and t > 16
                    if geglu:
                        assert (t & 15) not in (6, 7, 9)
                    if (t & 15) in (6, 7):
                        assert N % 160 == 0
    assert ops.choose_splits(64, 1280, 11520) > 1 and ops.choose_splits(65536, 320, 2880) == 1


def test_every_table_mode_has_a_gpu_parity_case():
    """tests/test_bench_path_gpu.py::test_tuned_gemm_mode_on_its_table_shape is parametrised over
    gemm_table_cases.cases(); no table entry may select a (tile, gather, sources, epilogue, split-K)
    combination outside that list, and every case must be a launchable table shape."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gemm_table_cases as gtc
    cases = gtc.cases()
    covered = {gtc.mode(s) for s in cases}
    for s in gtc.table().values():
        assert gtc.mode(s) in covered, s.key
    assert len(cases) == len(covered) >= 20
    for s in cases:
        assert s.K == s.taps * (s.c0 + s.c1) and s.K % 64 == 0 and s.tile in LEGAL
        if s.taps == 9:
            b = s.M // (s.hout * s.hout)
            assert b >= 1 and b * s.hout * s.hout == s.M
            assert s.hout == ((2 * s.hin if s.ups else s.hin) - 1) // s.stride + 1


def test_table_tile_44_is_not_handed_to_an_epilogue_it_does_not_have():
    """The table is keyed by (M, N, K, gather, GEGLU): a caller with the same key but an fp32 output, an fp32 residual, a
    narrow / unaligned output or a second source must get the heuristic tile, not the 256 x 256 ring (which would
    return LGD_ERR_ARG at launch)."""
    import torch
    key44 = None
    for mode in ("latency", "throughput"):
        with ops.tuning(mode):
            for k, e in ops.tuning_table().items():
                if e["tile"] == 44 and k.endswith("_e0_b1"):
                    key44, mode44 = k, mode
                    break
        if key44:
            break
    if key44 is None:
        return                                             # no such entry in this table version: nothing to guard
    m = re.match(r"M(\d+)_N(\d+)_K(\d+)_", key44)
    M, N, K = (int(m.group(i)) for i in (1, 2, 3))
    a = torch.empty(1, dtype=torch.float16)                 # descriptors only hold addresses; nothing is launched here
    with ops.tuning(mode44):
        plain = ops.gemm_desc(a, a, a, M, N, K)
        assert plain.tile == 44 and plain.splits == 1
        assert ops.gemm_desc(a, a, a, M, N, K, epi=ops.EPI_OUT_F32).tile != 44
        assert ops.gemm_desc(a, a, a, M, N, K, res=a, ldr=N, epi=ops.EPI_RES_F32).tile != 44
        assert ops.gemm_desc(a, a, a, M, N, K, ldc=N + 4).tile != 44
        assert ops.gemm_desc(a[0:0].new_empty(9)[1:], a, a, M, N, K).tile == 44      # the OPERAND's alignment is not the rule
        assert ops.gemm_desc(a, a, torch.empty(9, dtype=torch.float16)[1:], M, N, K).tile != 44   # output base not 16-byte aligned
