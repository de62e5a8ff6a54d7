"""GPU-box tool: TF/s of the GEMM kernel on the benchmark's heaviest shapes (tuned tile/split)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
SHAPES = [  # (M, N, K, taps, c0, c1, h, geglu)
    (65536, 320, 2880, 9, 320, 0, 64, 0), (32768, 320, 2880, 9, 320, 0, 64, 0), (16384, 640, 5760, 9, 640, 0, 32, 0),
    (4096, 1280, 11520, 9, 1280, 0, 16, 0), (1024, 1280, 11520, 9, 1280, 0, 8, 0), (65536, 2560, 320, 1, 320, 0, 0, 1),
    (16384, 5120, 640, 1, 640, 0, 0, 1), (65536, 320, 1280, 1, 1280, 0, 0, 0), (65536, 320, 320, 1, 320, 0, 0, 0),
    (65536, 960, 320, 1, 320, 0, 0, 0), (4096, 1280, 1280, 1, 1280, 0, 0, 0), (32768, 320, 5760, 9, 640, 0, 64, 0),
    (1024, 1280, 1280, 1, 1280, 0, 0, 0), (256, 1280, 1280, 1, 1280, 0, 0, 0), (256, 1280, 11520, 9, 1280, 0, 8, 0),
    (64, 1280, 11520, 9, 1280, 0, 8, 0), (16384, 640, 640, 1, 640, 0, 0, 0),
]
if os.environ.get("CONVBIG"): SHAPES = [s for s in SHAPES if s[3] == 9 and s[0] >= 16384]
if os.environ.get("SMALL"): SHAPES = [s for s in SHAPES if s[0] <= 4096 or s[2] <= 640]
tiles = [int(x) for x in os.environ.get("TILES", "0").split(",")]
tot = 0.0
for (M, N, K, taps, c0, c1, h, geglu) in SHAPES:
    rows_in = M
    a0 = torch.randn(rows_in, c0, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    n_out = N // 2 if geglu else N
    c = torch.empty(M, n_out, device=dev, dtype=torch.float16)
    bias = torch.zeros(N, device=dev)
    res = torch.zeros(M, n_out, device=dev, dtype=torch.float16) if (os.environ.get("RES") and not geglu) else None
    best = None
    for tile in tiles:
        d = ops.gemm_desc(a0, w, c, M, N, K, c0=c0, c1=c1, lda0=c0, taps=taps, hin=h, win=h, hout=h, wout=h,
                          bias=bias, res=res, ldr=n_out, epi=geglu, ldc=n_out, tile=tile,
                          splits=None if tile == 0 else int(os.environ.get("SPLITS", "1")))
        try:
            ops.gemm_launch(d); torch.cuda.synchronize()
        except RuntimeError:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_launch(d)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        if best is None or us < best[0]: best = (us, d.tile, d.splits)
    if best is None:
        continue
    tot += best[0]
    print(f"M{M:6d} N{N:5d} K{K:6d} t{taps} g{geglu}: tile {best[1]} split {best[2]} {best[0]:8.1f} us {2.0*M*N*K/best[0]/1e6:7.1f} TF/s")
print(f"total {tot:.1f} us")
