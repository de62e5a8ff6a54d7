import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lgd_amd
from lgd_amd import weights, ops
from lgd_amd.unet import UNetEngine
from lgd_amd.sampler import LMDSampler
from lgd_amd.scheduler import DDIMScheduler
dev = torch.device("cuda:0")
GOLD = os.path.join(ROOT, "tests", "golden")
L = 32
cfg = weights.CONFIGS["tiny"]; sd = weights.synth_state_dict(cfg, 0)
eng = UNetEngine(cfg, dev, sd)
g = np.load(os.path.join(GOLD, "loops_tiny.npz"))
ehs = torch.from_numpy(g["ehs"]); lall = torch.from_numpy(g["lat_all_in"])
sm = LMDSampler(eng, DDIMScheduler())
orig = ops.cfg_ddim_step
def wrapped(eps, x, x_out, *a, **k):
    r = orig(eps, x, x_out, *a, **k)
    if os.environ.get("LGD_SYNC"): torch.cuda.synchronize()
    print("   step: eps", float(eps.norm()), "x", float(x.norm()), "xo", float(x_out.norm()), "idx", int(eng.step_idx.item()))
    return r
ops.cfg_ddim_step = wrapped
import lgd_amd.sampler as S
out = sm.denoise(lall[0], ehs, 4)
print("hist norms", [float(out["latents_all"][i].norm()) for i in range(5)])
out = sm.denoise(lall[0], ehs, 4, saved_cross_attn_keys=[("mid",0,0,0)])
print("hist norms (1 key)", [float(out["latents_all"][i].norm()) for i in range(5)])
