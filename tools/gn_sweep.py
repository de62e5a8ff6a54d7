"""GPU-box tool: sweep of the two-launch GroupNorm's launch geometry (statistics chunks per image, apply workgroups per
launch) on the benchmark's large maps; graph-captured timing (tools/gn_bench.py timeit)."""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd  # noqa: E402,F401
from lgd_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, n=40):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


SHAPES = [(16, 4096, 320, 0), (8, 4096, 320, 0), (4, 4096, 320, 0), (16, 1024, 640, 0), (8, 1024, 640, 0), (16, 4096, 320, 320),
          (16, 1024, 1280, 640), (16, 1024, 1280, 0), (16, 256, 2560, 0)]
for B, HW, C0, C1 in SHAPES:
    C, G = C0 + C1, 32
    x0 = torch.randn(B * HW, C0, device=dev).half()
    x1 = torch.randn(B * HW, C1, device=dev).half() if C1 else None
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    out = torch.empty(B * HW, C, device=dev, dtype=torch.float16)
    line = f"B{B:3d} HW{HW:5d} C{C0}+{C1}:"
    for swgs in (256, 512, 1024, 2048):
        ops.GN_STATS_WGS = swgs
        nch = ops.gn_chunks(B, HW)
        part = torch.empty(B, nch, G, 2, device=dev)
        for awgs in (256, 512, 1024, 2048):
            ops.set_option("gn_apply_wgs", awgs)
            t = timeit(lambda: ops.groupnorm(x0, B, HW, G, 1e-5, gamma, beta, True, x1=x1, out=out, part=part))
            line += f"  s{swgs}/a{awgs} {t:5.1f}"
    print(line, flush=True)
ops.set_option("gn_apply_wgs", 1024)
