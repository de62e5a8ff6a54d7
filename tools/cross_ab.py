"""GPU-box tool: cross-attention forward WITHOUT map capture (77 text keys) — the flash kernels (option cross_resident = 0)
against the resident-keys kernel of round 6 on the same operands; graph-captured timing (the option is read at launch, i.e. at capture)."""
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


for (B, H, S, d) in [(16, 8, 4096, 40), (8, 8, 4096, 40), (4, 8, 4096, 40), (16, 8, 1024, 80), (8, 8, 1024, 80), (16, 8, 256, 160), (4, 8, 256, 160),
                     (16, 8, 64, 160)]:
    C, T = H * d, 77
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, S, C, generator=g).to(dev).half()
    kv = torch.randn(B, T, 2 * C, generator=g).to(dev).half()
    k, v = kv, kv[:, :, C:]
    view = (2 * C, T * 2 * C)
    o1 = torch.zeros(B, S, C, device=dev, dtype=torch.float16)
    o2 = torch.zeros(B, S, C, device=dev, dtype=torch.float16)
    f1 = lambda: ops.cross_attn_fwd(q, k, v, o1, B, H, S, T, d, d ** -0.5, k_view=view, v_view=view)
    f2 = lambda: ops.cross_attn_fwd(q, k, v, o2, B, H, S, T, d, d ** -0.5, k_view=view, v_view=view)
    ops.set_option("cross_resident", 0)
    f1()
    ops.set_option("cross_resident", 1)
    f2(); torch.cuda.synchronize()
    sl = slice(0, d)
    p = (q[0, :, sl].float() @ kv[0, :, sl].float().t() * d ** -0.5).softmax(-1)
    ref = p @ kv[0, :, C:C + d].float()
    e1 = float((o1[0, :, sl].float() - ref).abs().max() / ref.abs().max())
    e2 = float((o2[0, :, sl].float() - ref).abs().max() / ref.abs().max())
    ops.set_option("cross_resident", 0)
    t1 = timeit(f1)
    ops.set_option("cross_resident", 1)
    t2 = timeit(f2)
    print(f"B{B} S{S} d{d}: flash kernels {t1:6.1f} us (err {e1:.1e})   resident keys {t2:6.1f} us (err {e2:.1e})  x{t1 / t2:.2f}  {4.0 * B * S * C / t2 * 1e-6:.2f} TB/s of Q + O", flush=True)
