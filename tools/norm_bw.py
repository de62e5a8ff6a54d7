"""GPU-box tool: GroupNorm(+SiLU) / LayerNorm forward bandwidth on the benchmark's map sizes (both GN kernels together;
bytes = read x twice + write y once for GN, read + write for LN)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[2]
for (B, HW, C) in [(8, 4096, 320), (8, 4096, 640), (8, 1024, 640), (8, 1024, 1280), (8, 256, 1280), (8, 256, 2560), (8, 64, 1280), (4, 4096, 320), (1, 4096, 320)]:
    x = torch.randn(B * HW, C, device=dev).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    part = torch.empty((B, ops.gn_chunks(B, HW), 32, 2), device=dev)
    us = timeit(lambda: ops.groupnorm(x, B, HW, 32, 1e-5, g, b, True, out=out, part=part))
    mb = x.numel() * 2 / 1e6
    print(f"groupnorm B{B} HW{HW} C{C}: {mb:6.1f} MB map, stats+apply {us:6.1f} us -> {3 * mb / us:5.2f} TB/s (2 reads + 1 write)", flush=True)
for (rows, C) in [(32768, 320), (8192, 640), (2048, 1280), (512, 1280)]:
    x = torch.randn(rows, C, device=dev).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    us = timeit(lambda: ops.layernorm(x, g, b, 1e-5, out=out))
    mb = x.numel() * 2 / 1e6
    print(f"layernorm rows{rows} C{C}: {mb:6.1f} MB, {us:6.1f} us -> {2 * mb / us:5.2f} TB/s", flush=True)
