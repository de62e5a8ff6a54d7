import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
VARIANTS = [("default", dict(attn32=1, attn32_nw=8)), ("32x32 nw8 forced", dict(attn32=2, attn32_nw=8)), ("32x32 nw4 forced", dict(attn32=2, attn32_nw=4))]
for (B, H, S, Sk, d) in [(16, 8, 1024, 1024, 80), (8, 8, 1024, 1024, 80), (4, 8, 1024, 1024, 80), (8, 8, 1024, 1054, 80), (4, 8, 1024, 1054, 80)]:
    C = H * d
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, S, C, generator=g).to(dev).half(); k = torch.randn(B, Sk, C, generator=g).to(dev).half(); v = torch.randn(B, Sk, C, generator=g).to(dev).half()
    o = torch.empty(B, S, C, device=dev, dtype=torch.float16)
    f = lambda: ops.attn_fwd(q, k, v, o, B, H, S, Sk, d, d ** -0.5)
    line = f"B{B} S{S}x{Sk} d{d}:"
    times = {n: [] for n, _ in VARIANTS}
    for _ in range(6):
        for name, opts in VARIANTS:
            for kk, vv in opts.items(): ops.set_option(kk, vv)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 100)
    for name, _ in VARIANTS:
        us = sorted(times[name][1:])[2]
        line += f"  [{name}] {us:6.1f} us {4.0 * B * H * S * Sk * d / us / 1e6:5.0f} TF"
    print(line, flush=True)
