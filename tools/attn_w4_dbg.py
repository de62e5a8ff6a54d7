import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0"); d = 40
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
for (B, H, S, Sk) in [(1, 1, 256, 192), (1, 1, 256, 256)]:
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, S, H * d, generator=g).to(dev).half()
    k = torch.randn(B, Sk, H * d, generator=g).to(dev).half()
    v = torch.randn(B, Sk, H * d, generator=g).to(dev).half()
    o = torch.full((B, S, H * d), float("nan"), device=dev, dtype=torch.float16)
    L = torch.full((B, H, S), float("nan"), device=dev)
    ops.set_option("attn_w4", 2)
    ops.attn_fwd(q, k, v, o, B, H, S, Sk, d, d ** -0.5, lse=L)
    torch.cuda.synchronize()
    logits = q[0].float() @ k[0].float().t() * d ** -0.5
    ref = logits.softmax(-1) @ v[0].float()
    lref = torch.logsumexp(logits, -1) * 1.4426950408889634
    e = (o[0].float() - ref).abs()
    bad = ~(e < 1e-2)
    print(f"S{S}x{Sk}: nan {int(torch.isnan(o).sum())} bad {int(bad.sum())} of {bad.numel()}; lse bad {int((~((L[0,0]-lref).abs()<2e-2)).sum())}")
    print(" bad per dv column:", bad.sum(0).tolist())
    print(" bad per query (first 70):", bad.sum(1)[:70].tolist())
    if bad.any():
        qi = int(bad.sum(1).argmax())
        print(" query", qi, "got", o[0, qi].float().cpu(), "\n ref", ref[qi].cpu())
