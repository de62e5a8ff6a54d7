"""A/B timing of GroupNorm forward / backward (one-launch slab kernels, option gn_slab = 1, against the two-launch kernels)
and of the statistics-only LayerNorm (streaming kernel, option ln_stream = 1, against the row kernels) on the benchmark's shapes.  python tools/gn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgd_amd  # noqa: E402,F401
from lgd_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
FWD = [(16, 4096, 320, 0), (8, 4096, 320, 0), (4, 4096, 320, 0), (16, 1024, 640, 0), (8, 1024, 640, 0), (16, 4096, 320, 320), (16, 4096, 640, 320),
       (16, 1024, 1280, 640), (16, 1024, 640, 640), (16, 256, 1280, 0), (4, 1024, 640, 0)]
BWD = [(4, 64, 1280, 0), (4, 256, 1280, 0), (4, 4096, 320, 0), (4, 1024, 640, 0), (4, 64, 1280, 1280), (4, 256, 1280, 1280), (4, 256, 1280, 640),
       (4, 1024, 320, 0), (4, 256, 640, 0)]


def timeit(fn, n=40):
    """Average device time of fn: n calls captured into ONE hipGraph and replayed (eager launches through ctypes are
    host-bound below ~11 us per call and cannot resolve the small kernels)."""
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


def run(kind, shapes):
    for B, HW, C0, C1 in shapes:
        C, G = C0 + C1, 32
        x0 = torch.randn(B * HW, C0, device=dev).half()
        x1 = torch.randn(B * HW, C1, device=dev).half() if C1 else None
        gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
        gy = torch.randn(B * HW, C, device=dev).half()
        st = torch.zeros(B, G, 2, device=dev)
        out = torch.empty(B * HW, C, device=dev, dtype=torch.float16)
        gx0 = torch.empty_like(x0)
        gx1 = torch.empty_like(x1) if C1 else None
        part = torch.empty(B, ops.gn_chunks(B, HW), G, 2, device=dev)
        ops.groupnorm(x0, B, HW, G, 1e-5, gamma, beta, True, x1=x1, stats=st, out=out, part=part)
        res = {}
        for slab in (0, 1):
            ops.set_option("gn_slab", slab)
            if kind == "fwd":
                res[slab] = timeit(lambda: ops.groupnorm(x0, B, HW, G, 1e-5, gamma, beta, True, x1=x1, stats=st, out=out, part=part))
            else:
                res[slab] = timeit(lambda: ops.groupnorm_bwd(gy, x0, B, HW, G, gamma, beta, True, st, x1=x1, gx0=gx0, gx1=gx1, part=part))
        by = (4.0 if kind == "fwd" else 6.0) * B * HW * C           # bytes moved once each way (+ gy for the backward)
        print(f"{kind} B{B:3d} HW{HW:5d} C{C0}+{C1}: two-launch {res[0]:7.1f} us  slab {res[1]:7.1f} us  x{res[0] / res[1]:.2f}  "
              f"slab {by / res[1] * 1e-6:.2f} TB/s (single-pass bytes)")
    ops.set_option("gn_slab", 1)


run("fwd", FWD)
run("bwd", BWD)
for rows, C in [(65536, 320), (32768, 320), (16384, 640), (8192, 640), (4096, 1280), (2048, 1280), (1024, 1280), (16384, 320)]:
    x = torch.randn(rows, C, device=dev).half()
    st = torch.empty(rows, 2, device=dev)
    res = {}
    for stream in (0, 1):
        ops.set_option("ln_stream", stream)
        res[stream] = timeit(lambda: ops.layernorm_stats(x, C, stats=st))
    ops.set_option("ln_stream", 1)
    print(f"ln stats R{rows:6d} C{C:5d}: rows kernel {res[0]:6.1f} us  stream {res[1]:6.1f} us  x{res[0] / res[1]:.2f}  {2.0 * rows * C / res[1] * 1e-6:.2f} TB/s")
