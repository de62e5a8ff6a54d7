import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lgd_amd
from lgd_amd import weights, ops
from lgd_amd.unet import UNetEngine
from lgd_amd.scheduler import DDIMScheduler
import restate as R
dev = torch.device("cuda:0")
GOLD = os.path.join(ROOT, "tests", "golden")
L = 32
def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
def cfgd(cfg):
    return dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
                attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups,
                norm_eps=cfg.norm_eps, gligen_positive_len=cfg.gligen_positive_len)
cfg = weights.CONFIGS["tiny"]; sd = weights.synth_state_dict(cfg, 0); cd = cfgd(cfg)
eng = UNetEngine(cfg, dev, sd)
g = np.load(os.path.join(GOLD, "loops_tiny.npz"))
ehs = torch.from_numpy(g["ehs"])
lat = torch.from_numpy(g["lat_all_in"])[0]
sch = DDIMScheduler(); sch.set_timesteps(4)
eng.prepare_timesteps([int(t) for t in sch.timesteps]); eng.prepare_text(ehs)
for keys in ([], [("mid",0,0,0)]):
    plan = eng.plan(2, L, fuser=False, save_keys=keys)
    for idx in range(2):
        eng.set_step(idx)
        t = int(sch.timesteps[idx])
        eps = plan.forward(lat.expand(2, 4, L, L).to(dev))
        torch.cuda.synchronize()
        with torch.no_grad():
            taps = {}
            ref = R.unet_forward(sd, cd, torch.cat([lat]*2), t, ehs, taps=taps)
        print("keys", keys, "step", idx, "t", t, "eps rel", rel(eps, ref), "finite", bool(torch.isfinite(eps).all()))
        if rel(eps, ref) > 0.05 or not torch.isfinite(eps).all():
            for k, v in plan.dbg.items():
                r = taps[k]
                r = r.permute(0, 2, 3, 1).reshape(-1, r.shape[1]) if r.dim() == 4 else r.reshape(-1, r.shape[-1])
                e = rel(v.t, r)
                print(f"    {k:55s} {e:.3e}")
                if e > 0.05: break
# step kernel
ctab = sch.coef_table(7.5, dev)
x = lat.to(dev).clone(); xo = torch.empty_like(x)
hist = torch.zeros(5, 1, 4, L, L, device=dev)
eps = torch.randn(2, 4, L, L, device=dev)
for idx in range(4):
    eng.set_step(idx)
    ops.cfg_ddim_step(eps, x, xo, ctab, eng.step_idx, hist=hist)
    torch.cuda.synchronize()
    print("step", idx, "xo norm", float(xo.norm()), "hist norms", [round(float(hist[i].norm()), 2) for i in range(5)])
