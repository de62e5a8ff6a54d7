"""GPU-box tool: effective bandwidth of GroupNorm(+SiLU) / LayerNorm on the benchmark's shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, HW, C0, C1) in [(8, 4096, 320, 0), (8, 4096, 320, 320), (8, 4096, 640, 320), (8, 1024, 640, 0), (8, 1024, 1280, 640),
                        (8, 256, 1280, 0), (8, 256, 1280, 1280), (8, 64, 1280, 0), (4, 4096, 320, 0), (4, 1024, 640, 0)]:
    C = C0 + C1
    x = torch.randn(B * HW, C0, device=dev).half()
    x1 = torch.randn(B * HW, C1, device=dev).half() if C1 else None
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty(B * HW, C, device=dev, dtype=torch.float16)
    stats = torch.empty(B, 32, 2, device=dev)
    us = timeit(lambda: ops.groupnorm(x, B, HW, 32, 1e-5, g, b, True, x1=x1, out=out, stats=stats))
    nbytes = B * HW * C * 2 * 3
    print(f"GN  B{B} HW{HW} C{C0}+{C1}: {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s (2R+1W)")
for (rows, C) in [(32768, 320), (8192, 640), (2048, 1280), (512, 1280), (16384, 320)]:
    x = torch.randn(rows, C, device=dev).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    us = timeit(lambda: ops.layernorm(x, g, b, out=out))
    print(f"LN  rows{rows} C{C}: {us:7.1f} us  {rows * C * 4 / us / 1e6:6.2f} TB/s (1R+1W)")
