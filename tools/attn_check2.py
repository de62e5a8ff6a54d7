import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
B, H, S, d = 2, 8, 256, 40
C = H * d
g = torch.Generator(device="cpu").manual_seed(0)
q0, k0, v0 = (torch.randn(B, S, H, d, generator=g) for _ in range(3))
def run(name, q, k, v):
    q, k, v = (t.reshape(B, S, C).to(dev).half() for t in (q, k, v))
    o = torch.empty(B, S, C, device=dev, dtype=torch.float16)
    lse = torch.empty(B, H, S, device=dev)
    ops.attn_fwd(q, k, v, o, B, H, S, S, d, d ** -0.5, lse=lse)
    sp = lambda t: t.float().reshape(B, S, H, d).permute(0, 2, 1, 3)
    sc = (sp(q) @ sp(k).transpose(-1, -2)) * d ** -0.5
    ref = (sc.softmax(-1) @ sp(v)).permute(0, 2, 1, 3).reshape(B, S, C)
    lse_ref = torch.logsumexp(sc, -1) * 1.4426950408889634
    print(f"{name:12s}: out relerr {float((o.float() - ref).abs().max() / ref.abs().max()):.3e}  lse abserr {float((lse - lse_ref).abs().max()):.3e}  lse[0,0,:3]={lse[0,0,:3].tolist()} ref={lse_ref[0,0,:3].tolist()}")
z = torch.zeros_like(q0)
run("q=0", z, k0, v0)
m = torch.zeros(d); m[:32] = 1
run("dims<32", q0 * m, k0, v0)
run("dims>=32", q0 * (1 - m), k0, v0)
run("full", q0, k0, v0)
