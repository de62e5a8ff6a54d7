"""GPU-box tool: one-launch GroupNorm (option gn_fused) against the two-launch kernels on the benchmark's small-map
shapes, and a whole no-grad forward of the full-width UNet either way."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgd_amd  # noqa
from lgd_amd import ops, weights
from lgd_amd.unet import UNetEngine
dev = torch.device("cuda:0")


def timeit(f, n=50):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, HW, C0, C1) in [(8, 64, 1280, 0), (8, 64, 1280, 1280), (8, 256, 640, 0), (8, 256, 1280, 0), (8, 256, 1280, 1280), (8, 256, 1280, 640),
                        (16, 256, 1280, 1280), (16, 64, 1280, 1280), (8, 1024, 640, 0), (8, 1024, 640, 320), (8, 1024, 1280, 640), (16, 1024, 1280, 640)]:
    C = C0 + C1
    x = torch.randn(B * HW, C0, device=dev).half()
    x1 = torch.randn(B * HW, C1, device=dev).half() if C1 else None
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty(B * HW, C, device=dev, dtype=torch.float16)
    part = torch.empty(B, ops.gn_chunks(B, HW), 32, 2, device=dev)
    f = lambda: ops.groupnorm(x, B, HW, 32, 1e-5, g, b, True, x1=x1, out=out, part=part)
    ops.set_option("gn_fused", 0); t2 = timeit(f)
    ops.set_option("gn_fused", 1024); t1 = timeit(f)
    print(f"GN B{B} HW{HW} C{C0}+{C1}: two launches {t2:6.1f} us, one launch {t1:6.1f} us", flush=True)

cfg = weights.CONFIGS["sd14_gligen"]
eng = UNetEngine(cfg, dev, state_dict=weights.synth_state_dict(cfg, 0), max_text_batch=32)
un, co = weights.synth_embeddings(cfg, 1)
eng.prepare_timesteps([501]); eng.set_step(0)
for B in (8, 16):
    eng.prepare_text(torch.cat([un] * (B // 2) + [co] * (B // 2)).to(dev))
    x = torch.randn(B, 4, 64, 64, device=dev)
    plan = eng.plan(B, 64, fuser=False)
    res = {}
    for rep in range(3):
        for hw in (0, 256, 1024):
            ops.set_option("gn_fused", hw)
            plan.forward(x); torch.cuda.synchronize()
            res.setdefault(hw, []).append(timeit(lambda: plan.forward(), 10) / 1e3)
    print(f"forward B={B}: " + ", ".join(f"gn_fused={hw}: {sorted(v)[1]:.3f} ms" for hw, v in res.items()), flush=True)
ops.set_option("gn_fused", 256)
