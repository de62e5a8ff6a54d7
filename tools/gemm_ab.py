"""GPU-box tool: within-process A/B of GEMM tile variants on benchmark shapes — every variant is first
checked against fp32 torch, then timed in interleaved rounds (median / min of per-round averages).

    TILES=22,33,34 [SPLITS=1] [SHAPES=conv|plain|geglu|all] [ROUNDS=5] python tools/gemm_ab.py
"""
import os
import statistics
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd  # noqa: E402,F401
from lgd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ALL = {  # (M, N, K, taps, cin, h, geglu)
    "conv": [(65536, 320, 2880, 9, 320, 64, 0), (32768, 320, 2880, 9, 320, 64, 0), (16384, 640, 5760, 9, 640, 32, 0),
             (4096, 1280, 11520, 9, 1280, 16, 0), (65536, 320, 5760, 9, 640, 64, 0), (8192, 640, 5760, 9, 640, 32, 0),
             (16384, 320, 2880, 9, 320, 64, 0)],
    "plain": [(65536, 320, 1280, 1, 1280, 0, 0), (65536, 320, 320, 1, 320, 0, 0), (65536, 960, 320, 1, 320, 0, 0),
              (16384, 640, 2560, 1, 2560, 0, 0), (16384, 640, 640, 1, 640, 0, 0), (4096, 1280, 5120, 1, 5120, 0, 0),
              (4096, 1280, 1280, 1, 1280, 0, 0), (16384, 1920, 640, 1, 640, 0, 0), (4096, 3840, 1280, 1, 1280, 0, 0),
              (32768, 320, 1280, 1, 1280, 0, 0)],
    "sq": [(4096, 4096, 4096, 1, 4096, 0, 0), (8192, 8192, 4096, 1, 4096, 0, 0)],
    "small": [(256, 1280, 1280, 1, 1280, 0, 0), (1024, 1280, 1280, 1, 1280, 0, 0), (512, 1280, 1280, 1, 1280, 0, 0),
              (2048, 640, 640, 1, 640, 0, 0), (1024, 1280, 5120, 1, 5120, 0, 0), (256, 1280, 11520, 9, 1280, 8, 0),
              (1024, 1280, 11520, 9, 1280, 16, 0), (64, 1280, 11520, 9, 1280, 8, 0), (2048, 1280, 11520, 9, 1280, 16, 0),
              (4096, 640, 5760, 9, 640, 32, 0), (512, 1280, 11520, 9, 1280, 8, 0)],
    "geglu": [(65536, 2560, 320, 1, 320, 0, 1), (16384, 5120, 640, 1, 640, 0, 1), (4096, 10240, 1280, 1, 1280, 0, 1)],
}
which = os.environ.get("SHAPES", "conv")
SHAPES = sum(ALL.values(), []) if which == "all" else sum((ALL[w] for w in which.split(",")), [])
if os.environ.get("SHAPE"):          # one explicit shape: SHAPE=M,N,K,taps,cin,h,geglu
    SHAPES = [tuple(int(x) for x in os.environ["SHAPE"].split(","))]
# TILES: comma list of tile codes, each optionally "tile:splits" (tile 0 = the tuning table's choice)
splits = int(os.environ.get("SPLITS", "1"))
if os.environ.get("FIRST"):
    SHAPES = SHAPES[:int(os.environ["FIRST"])]
tiles = [(int(x.split(":")[0]), int(x.split(":")[1]) if ":" in x else splits)
         for x in os.environ.get("TILES", "22,33").split(",")]
rounds = int(os.environ.get("ROUNDS", "5"))
reps = int(os.environ.get("REPS", "10"))


def pack_conv_w(w):
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


for (M, N, K, taps, cin, h, geglu) in SHAPES:
    g = torch.Generator().manual_seed(1)
    bias = torch.randn(N, generator=g).to(dev)
    n_out = N // 2 if geglu else N
    if taps == 9:
        B = M // (h * h)
        x = torch.randn(B, cin, h, h, generator=g).to(dev).half()
        w4 = (torch.randn(N, cin, 3, 3, generator=g) * K ** -0.5).to(dev).half()
        w = pack_conv_w(w4)
        a0 = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
        ref = F.conv2d(x.float(), w4.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(M, N)
        kw = dict(taps=9, hin=h, win=h, hout=h, wout=h)
    else:
        a0 = torch.randn(M, K, generator=g).to(dev).half()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).half()
        ref = a0.float() @ w.float().t() + bias
        kw = dict(taps=1)
    res = None
    if geglu:
        idx = []
        n = N // 2
        for j in range(n // 16):
            idx += list(range(16 * j, 16 * j + 16)) + list(range(n + 16 * j, n + 16 * j + 16))
        idx = torch.tensor(idx, device=dev)
        v, gg = ref.chunk(2, dim=-1)
        ref = v * F.gelu(gg)
        w, bias = w[idx].contiguous(), bias[idx].contiguous()
    else:
        res = torch.randn(M, N, generator=g).to(dev).half()
        ref = ref + res.float()
    c = torch.empty(M, n_out, device=dev, dtype=torch.float16)
    descs = {}
    for tile, sp in tiles:
        if geglu and (((tile & 15) in (6, 7, 9) and tile < 32) or tile in (33, 37, 40, 47)):
            continue
        if sp > 1 and K // 64 < 2 * sp:
            continue
        try:
            d = ops.gemm_desc(a0, w, c, M, N, K, c0=cin, lda0=cin, bias=bias, res=res, ldr=n_out, epi=geglu | (int(os.environ.get("EPI_ABL", "0")) << 16), ldc=n_out,
                              tile=tile, splits=None if tile == 0 else sp, **kw)
        except RuntimeError as e:
            print(f"  tile {tile}/s{sp}: {e}")
            continue
        c.zero_()
        try:
            ops.gemm_launch(d)
            torch.cuda.synchronize()
        except RuntimeError as e:
            print(f"  tile {tile}: launch failed ({e})")
            continue
        err = float((c.float() - ref).abs().max() / ref.abs().max())
        descs[(tile, sp)] = (d, err)
    times = {t: [] for t in descs}
    for _ in range(rounds):
        for t, (d, _) in descs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.gemm_launch(d)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) * 1e3 / reps)
    fl = 2.0 * M * N * K
    line = f"M{M:6d} N{N:5d} K{K:6d} t{taps} g{geglu}:"
    for t, (d, err) in descs.items():
        med, mn = statistics.median(times[t]), min(times[t])
        line += f"  [{d.tile}/s{d.splits}] {med:7.1f}us {fl / med / 1e6:6.0f}TF (min {mn:6.1f}) err {err:.1e}{'' if err < 3e-3 else ' WRONG'}"
    print(line, flush=True)
