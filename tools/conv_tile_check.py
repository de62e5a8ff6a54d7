"""GPU-box tool: one 3x3 convolution shape under several tile codes against fp32 torch — where do the errors sit?"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgd_amd  # noqa
from lgd_amd import ops
dev = torch.device("cuda:0")
B, H, C, Cout = int(os.environ.get("B", 1)), int(os.environ.get("H", 64)), int(os.environ.get("C", 320)), int(os.environ.get("COUT", 320))
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(B, C, H, H, generator=g).to(dev).half()
w = (torch.randn(Cout, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(dev).half()
b = torch.randn(Cout, generator=g).to(dev)
r = torch.randn(B * H * H, Cout, generator=g).to(dev).half()
xl = x.permute(0, 2, 3, 1).reshape(B * H * H, C).contiguous()
wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(B * H * H, Cout) + r.float()
for tile in [int(t) for t in os.environ.get("TILES", "20,41,38,34,33").split(",")]:
    for rep in range(3):
        y = ops.conv3x3(xl, wp, B, H, H, bias=b, res=r, tile=tile, splits=1).float()
        d = (y - ref).abs()
        bad = (d > 0.02).nonzero()
        print(f"tile {tile} rep {rep}: max err {d.max().item():.3e} rel-L2 {((y - ref).norm() / ref.norm()).item():.3e}  elements > 0.02: {bad.shape[0]}"
              + (f"  rows {bad[:, 0].min().item()}..{bad[:, 0].max().item()} cols {bad[:, 1].min().item()}..{bad[:, 1].max().item()}" if bad.shape[0] else ""), flush=True)
