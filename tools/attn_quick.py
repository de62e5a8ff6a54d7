"""GPU-box tool: times the self-attention forward (and backward) kernels on the benchmark's shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
bwd = "--bwd" in sys.argv
for (B, H, S, d) in [(8, 8, 4096, 40), (4, 8, 4096, 40), (8, 8, 1024, 80), (8, 8, 256, 160), (8, 8, 64, 160)]:
    C = H * d
    qkv = torch.randn(B, S, 3 * C, device=dev).half()
    o = torch.empty(B, S, C, device=dev, dtype=torch.float16)
    lse = torch.empty(B, H, S, device=dev)
    view = (3 * C, S * 3 * C)
    f = lambda: ops.attn_fwd(qkv, qkv[:, :, C:], qkv[:, :, 2 * C:], o, B, H, S, S, d, d ** -0.5, lse=lse,
                             q_view=view, k_view=view, v_view=view)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"fwd B{B} H{H} S{S} d{d}: {us:8.1f} us  {4.0 * B * H * S * S * d / us / 1e6:7.1f} TF/s (algorithmic)")
    if bwd:
        q, k, v = (torch.randn(B, S, C, device=dev).half() for _ in range(3))
        go = torch.randn(B, S, C, device=dev).half()
        ops.attn_fwd(q, k, v, o, B, H, S, S, d, d ** -0.5, lse=lse)
        gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty(B, H, S, device=dev)
        fb = lambda: ops.attn_bwd(q, k, v, o, go, lse, delta, gq, gk, gv, B, H, S, S, d, d ** -0.5)
        fb(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5): fb()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 200
        print(f"bwd B{B} H{H} S{S} d{d}: {us:8.1f} us  {10.0 * B * H * S * S * d / us / 1e6:7.1f} TF/s (algorithmic)")
