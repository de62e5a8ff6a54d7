import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
for (B, H, S, d) in [(2, 8, 256, 40), (16, 8, 1024, 40), (2, 8, 256, 80), (2, 8, 300, 24)]:
    C = H * d
    g = torch.Generator(device="cpu").manual_seed(0)
    q, k, v = (torch.randn(B, S, C, generator=g).to(dev).half() for _ in range(3))
    o = torch.empty(B, S, C, device=dev, dtype=torch.float16)
    lse = torch.empty(B, H, S, device=dev)
    ops.attn_fwd(q, k, v, o, B, H, S, S, d, d ** -0.5, lse=lse)
    sp = lambda t: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    sc = (sp(q) @ sp(k).transpose(-1, -2)) * d ** -0.5
    ref = (sc.softmax(-1) @ sp(v)).permute(0, 2, 1, 3).reshape(B, S, C)
    lse_ref = torch.logsumexp(sc, -1) * 1.4426950408889634
    e = float((o.float() - ref).abs().max() / ref.abs().max())
    el = float((lse - lse_ref).abs().max())
    print(f"B{B} S{S} d{d}: out relerr {e:.3e}  lse abserr {el:.3e}  nan={bool(torch.isnan(o).any())}")
