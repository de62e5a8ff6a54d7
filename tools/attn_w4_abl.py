"""GPU-box tool: time the d = 40 kernel of attn_w4.hip with one ingredient removed (library from tools/build_abl.sh;
LGD_W4_ABL is read once per process, so every variant runs in its own process)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0"); d = 40
B, H, S, Sk = 16, 8, 4096, 4096
g = torch.Generator().manual_seed(0)
q = torch.randn(B, S, H * d, generator=g).to(dev).half(); k = torch.randn(B, Sk, H * d, generator=g).to(dev).half()
v = torch.randn(B, Sk, H * d, generator=g).to(dev).half(); o = torch.empty_like(q)
ops.set_option("attn_w4", int(os.environ.get("W4MODE", "2")))
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attn_fwd(q, k, v, o, B, H, S, Sk, d, d ** -0.5)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
print(f"ABL={os.environ.get('LGD_W4_ABL', '0'):>3}: {sorted(ts)[2]:7.1f} us per launch (B16 H8 S4096)")
