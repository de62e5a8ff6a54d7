"""A/B of the LayerNorm fold (UNetEngine.fold_ln) on one box, one process: B = 8 and B = 16 no-grad plans of the
full-width sd14_gligen UNet, graph-free launch sequences timed with HIP events, alternating arms."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lgd_amd  # noqa
from lgd_amd import weights
from lgd_amd.unet import UNetEngine

dev = torch.device("cuda:0")
cfg = weights.CONFIGS["sd14_gligen"]
eng = UNetEngine(cfg, dev, state_dict=weights.synth_state_dict(cfg, 0), max_text_batch=32)
un, co = weights.synth_embeddings(cfg, 1)
eng.prepare_timesteps([501]); eng.set_step(0)
for B in (8, 16):
    eng.prepare_text(torch.cat([un] * (B // 2) + [co] * (B // 2)).to(dev))
    x = torch.randn(B, 4, 64, 64, device=dev)
    plans = {}
    for fold in (False, True):
        eng.fold_ln = fold
        plans[fold] = eng.plan(B, 64, fuser=False)
        plans[fold].forward(x)
    torch.cuda.synchronize()
    ts = {False: [], True: []}
    for it in range(12):
        for fold in (False, True):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                plans[fold].forward()
            e1.record(); torch.cuda.synchronize()
            ts[fold].append(e0.elapsed_time(e1) / 5)
    a, b = sorted(ts[False])[len(ts[False]) // 2], sorted(ts[True])[len(ts[True]) // 2]
    d = (plans[False].forward().float() - plans[True].forward().float()).abs().max().item()
    print(f"B={B}: two ops {a:.3f} ms, folded {b:.3f} ms ({(a / b - 1) * 100:+.2f}%), max |eps diff| {d:.2e}", flush=True)
