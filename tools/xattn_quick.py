"""GPU-box tool: times the cross-attention dQ kernel on the guidance pass's shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
for (B, H, Sq, d, with_gp) in [(4, 8, 4096, 40, False), (4, 8, 1024, 80, False), (4, 8, 256, 160, True), (4, 8, 64, 160, True)]:
    Sk, C = 77, H * d
    q, go = (torch.randn(B, Sq, C, device=dev).half() for _ in range(2))
    k, v = (torch.randn(B, Sk, C, device=dev).half() for _ in range(2))
    gp = torch.randn(B, H, Sq, Sk, device=dev) if with_gp else None
    gq = torch.empty_like(q)
    f = lambda: ops.cross_attn_bwd(q, k, v, go, gp, gq, B, H, Sq, Sk, d, d ** -0.5)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"xattn bwd B{B} H{H} Sq{Sq} d{d} gp={with_gp}: {e0.elapsed_time(e1) * 100:8.1f} us")
