"""GPU-box debugging aid (not part of the product): localises engine-vs-oracle differences."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lgd_amd
from lgd_amd import weights, ops
from lgd_amd.unet import UNetEngine
from lgd_amd.sampler import LMDSampler, prepare_gligen_condition
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

def part_a():
    print("== A: tiny denoise variants vs restate")
    cfg = weights.CONFIGS["tiny"]; sd = weights.synth_state_dict(cfg, 0); cd = cfgd(cfg)
    eng = UNetEngine(cfg, dev, sd)
    g = np.load(os.path.join(GOLD, "loops_tiny.npz"))
    ehs = torch.from_numpy(g["ehs"]); inp = (ehs, ehs[:1], ehs[1:])
    lall = torch.from_numpy(g["lat_all_in"]); fm = torch.from_numpy(g["frozen_mask"])
    sm = LMDSampler(eng, DDIMScheduler())
    # 1: no guidance, no frozen
    ps = []
    ref = R.generate_partial_frozen(sd, cd, R.DDIM(), lall, fm, inp, 4, 0, per_step=ps)
    out = sm.denoise(lall[0], ehs, 4)
    for i in range(4):
        print("  nofrozen step", i, rel(out["latents_all"][i + 1], ps[i]))
    ps = []
    ref = R.generate_partial_frozen(sd, cd, R.DDIM(), lall, fm, inp, 4, 2, per_step=ps)
    out = sm.denoise(lall, ehs, 4, frozen_steps=2, frozen_mask=fm)
    for i in range(4):
        print("  frozen step", i, rel(out["latents_all"][i + 1], ps[i]))

def part_b():
    print("== B: tiny_gligen forward, per-stage taps")
    name = "tiny_gligen"
    cfg = weights.CONFIGS[name]; sd = weights.synth_state_dict(cfg, 0); cd = cfgd(cfg)
    eng = UNetEngine(cfg, dev, sd)
    g = np.load(os.path.join(GOLD, f"unet_fwd_{name}.npz"))
    plan = eng.plan(2, L, fuser=True, save_keys=[])
    eng.prepare_timesteps([int(g["t"])]); eng.set_step(0)
    eng.prepare_text(torch.from_numpy(g["ehs"]))
    gl = dict(boxes=torch.from_numpy(g["gl_boxes"]), masks=torch.from_numpy(g["gl_masks"]),
              positive_embeddings=torch.from_numpy(g["gl_emb"]))
    eng.prepare_gligen(**gl)
    eps = plan.forward(torch.from_numpy(g["x"]).to(dev))
    torch.cuda.synchronize()
    print("  eps", rel(eps, g["eps"]), " per-batch:", rel(eps[0], g["eps"][0]), rel(eps[1], g["eps"][1]))
    objs_ref = R.position_net(sd, gl["boxes"], gl["masks"], gl["positive_embeddings"])
    print("  objs", rel(eng._objs.view(2, 30, -1), objs_ref))
    taps = {}
    with torch.no_grad():
        R.unet_forward(sd, cd, torch.from_numpy(g["x"]), int(g["t"]), torch.from_numpy(g["ehs"]), gligen=gl, taps=taps)
    for k, v in plan.dbg.items():
        if k in taps:
            r = taps[k]
            if r.dim() == 4:
                r = r.permute(0, 2, 3, 1).reshape(-1, r.shape[1])
            else:
                r = r.reshape(-1, r.shape[-1])
            print(f"  {k:60s} {rel(v.t, r):.3e}")

if __name__ == "__main__":
    which = sys.argv[1:] or ["a", "b"]
    if "a" in which: part_a()
    if "b" in which: part_b()
