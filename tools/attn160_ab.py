"""GPU-box tool: the d = 160 self-attention forward (16x16 level) — checked against fp32 torch and timed from a captured
graph (40 launches).  LGD_ATTN160=0 selects the 64-query workgroups it replaced (one process per arm)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd  # noqa: E402,F401
from lgd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


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


for (B, H, S, Sk, d) in [(16, 8, 256, 256, 160), (8, 8, 256, 256, 160), (4, 8, 256, 256, 160), (16, 8, 256, 286, 160), (8, 8, 256, 286, 160),
                         (4, 8, 256, 286, 160), (3, 8, 300, 300, 160), (16, 8, 64, 64, 160)]:
    C = H * d
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, S, C, generator=g).to(dev).half()
    k = torch.randn(B, Sk, C, generator=g).to(dev).half()
    v = torch.randn(B, Sk, C, generator=g).to(dev).half()
    o = torch.zeros(B, S, C, device=dev, dtype=torch.float16)
    f = lambda: ops.attn_fwd(q, k, v, o, B, H, S, Sk, d, d ** -0.5)
    f()
    torch.cuda.synchronize()
    err = 0.0
    for b, h in ((0, 0), (B - 1, H - 1), (B // 2, 3)):
        sl = slice(h * d, (h + 1) * d)
        p = (q[b, :, sl].float() @ k[b, :, sl].float().t() * d ** -0.5).softmax(-1)
        ref = p @ v[b, :, sl].float()
        err = max(err, float((o[b, :, sl].float() - ref).abs().max() / ref.abs().max()))
    us = timeit(f)
    print(f"LGD_ATTN160={os.environ.get('LGD_ATTN160', '1')} B{B} H{H} S{S}x{Sk} d{d}: {us:6.1f} us {4.0 * B * H * S * Sk * d / us / 1e6:6.1f} TF/s err {err:.1e}{'' if err < 4e-3 else ' WRONG'}",
          flush=True)
