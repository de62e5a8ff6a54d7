"""GPU-box tool: checks (vs fp32 torch, one head slice at a time) and times the self-attention forward kernel on the
benchmark's shapes.  LGD_ATTN_NW=8 selects the 8-wave workgroup variant (A/B)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import ops
dev = torch.device("cuda:0")
VARIANTS = [("16x16x32", dict(attn32=0)), ("32x32x16 forced", dict(attn32=2, attn32_nw=8, attn32_var=0)), ("default", dict(attn32=1))]
if os.environ.get("ONLY"): VARIANTS = [VARIANTS[int(os.environ["ONLY"])]]
SHAPES = [(16, 8, 4096, 4096, 40), (8, 8, 4096, 4096, 40), (8, 8, 4096, 4126, 40), (16, 8, 1024, 1024, 80),
          (16, 8, 1024, 1054, 80), (8, 5, 9216, 9216, 64), (16, 8, 256, 256, 160)]
if os.environ.get("FIRST"):
    SHAPES = SHAPES[:int(os.environ["FIRST"])]
for (B, H, S, Sk, d) in SHAPES:
    C = H * d
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, S, C, generator=g).to(dev).half()
    k = torch.randn(B, Sk, C, generator=g).to(dev).half()
    v = torch.randn(B, Sk, C, generator=g).to(dev).half()
    o = torch.empty(B, S, C, device=dev, dtype=torch.float16)
    f = lambda: ops.attn_fwd(q, k, v, o, B, H, S, Sk, d, d ** -0.5)
    line = f"fwd B{B} H{H} S{S}x{Sk} d{d}:"
    errs, times = {}, {n: [] for n, _ in VARIANTS}
    def select(opts):
        for kk, vv in opts.items():
            ops.set_option(kk, vv)
    for name, opts in VARIANTS:
        select(opts)
        o.zero_()
        f(); torch.cuda.synchronize()
        err = 0.0
        for b, h in ((0, 0), (B - 1, H - 1)):
            sl = slice(h * d, (h + 1) * d)
            p = (q[b, :, sl].float() @ k[b, :, sl].float().t() * d ** -0.5).softmax(-1)
            ref = p @ v[b, :, sl].float()
            err = max(err, float((o[b, :, sl].float() - ref).abs().max() / ref.abs().max()))
        errs[name] = err
    for _ in range(6):                      # interleaved rounds: the first variant of a process is not penalised
        for name, opts in VARIANTS:
            select(opts)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 100)
    for name, _ in VARIANTS:
        us = sorted(times[name][1:])[len(times[name][1:]) // 2]
        line += f"  [{name}] {us:7.1f} us {4.0 * B * H * S * Sk * d / us / 1e6:6.1f} TF/s err {errs[name]:.1e}{'' if errs[name] < 4e-3 else ' WRONG'}"
    print(line, flush=True)
ops.set_option("attn32", 1); ops.set_option("attn32_nw", 8); ops.set_option("attn32_var", 0)
