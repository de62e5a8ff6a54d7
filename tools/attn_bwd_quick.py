"""GPU-box tool: checks (vs torch autograd on one head) and times the self-attention backward kernels on the guidance
pass's shapes.  LGD_ATTN_BWD=0 / 1 / 2 selects single-buffered, double-buffered, double-buffered 8-wave (A/B)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
print("LGD_ATTN_BWD =", os.environ.get("LGD_ATTN_BWD", "default"))
for (B, H, S, Sk, d) in [(4, 8, 4096, 4096, 40), (4, 8, 4096, 4126, 40), (2, 8, 4096, 4096, 40), (4, 8, 1024, 1024, 80),
                         (4, 8, 1024, 1054, 80), (4, 8, 256, 256, 160), (2, 5, 9216, 9216, 64)]:
    C = H * d
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, S, C, generator=g).to(dev).half()
    k = torch.randn(B, Sk, C, generator=g).to(dev).half()
    v = torch.randn(B, Sk, C, generator=g).to(dev).half()
    go = torch.randn(B, S, C, generator=g).to(dev).half()
    o = torch.empty(B, S, C, device=dev, dtype=torch.float16)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    delta = torch.empty_like(lse)
    gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    scale = d ** -0.5
    ops.attn_fwd(q, k, v, o, B, H, S, Sk, d, scale, lse=lse)
    f = lambda: ops.attn_bwd(q, k, v, o, go, lse, delta, gq, gk, gv, B, H, S, Sk, d, scale)
    f(); torch.cuda.synchronize()
    err = 0.0
    for b, h in ((0, 0), (B - 1, H - 1)):
        sl = slice(h * d, (h + 1) * d)
        qq, kk, vv = (t[b, :, sl].float().clone().requires_grad_(True) for t in (q, k, v))
        out = (qq @ kk.t() * scale).softmax(-1) @ vv
        out.backward(go[b, :, sl].float())
        for mine, ref in ((gq, qq.grad), (gk, kk.grad), (gv, vv.grad)):
            err = max(err, float((mine[b, :, sl].float() - ref).abs().max() / ref.abs().max()))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    us = sorted(ts)[2]
    print(f"bwd B{B} H{H} S{S}x{Sk} d{d}: {us:8.1f} us  {10.0 * B * H * S * Sk * d / us / 1e6:7.1f} TF/s (algorithmic)  err {err:.1e}"
          f"{'' if err < 6e-3 else ' WRONG'}", flush=True)
