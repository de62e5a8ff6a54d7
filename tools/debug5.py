import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgd_amd
from lgd_amd import weights
from lgd_amd.unet import UNetEngine
from lgd_amd.sampler import LMDSampler, prepare_gligen_condition
from lgd_amd.scheduler import DDIMScheduler
dev = torch.device("cuda:0")
GOLD = os.path.join(ROOT, "tests", "golden")
KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]
OBJ_KEY = ("down", 2, 1, 0)
BBOXES = [[74 / 512, 177 / 512, (74 + 183) / 512, (177 + 235) / 512], [314 / 512, 193 / 512, (314 + 189) / 512, (193 + 216) / 512]]
OBJ_POS = [[1, 2, 3], [5, 6, 7]]
def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
cfg = weights.CONFIGS["tiny_gligen"]; sd = weights.synth_state_dict(cfg, 0)
g = np.load(os.path.join(GOLD, "loops_tiny_gligen.npz"))
ehs = torch.from_numpy(g["ehs"])
for use_graphs in (False, True):
    eng = UNetEngine(cfg, dev, sd)
    sm = LMDSampler(eng, DDIMScheduler(), use_graphs=use_graphs)
    guid = dict(bboxes=BBOXES, object_positions=OBJ_POS, loss_scale=5, loss_threshold=0.0, max_iter=[2, 1],
                max_index_step=3, guidance_attn_keys=KEYS, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0)
    gl = prepare_gligen_condition(BBOXES, torch.from_numpy(g["phrase_emb"]), dev)
    for rep in range(2):
        tr = []
        out = sm.denoise(torch.from_numpy(g["lat_all_in"]), ehs, 4, gligen=gl, gligen_scheduled_sampling_beta=0.5,
                         guidance=guid, frozen_steps=2, frozen_mask=torch.from_numpy(g["frozen_mask"]),
                         saved_cross_attn_keys=[OBJ_KEY, *KEYS], return_cond_ca_only=True, return_token_ca_only=7, trace=tr)
        torch.cuda.synchronize()
        print("graphs", use_graphs, "rep", rep, "lat_all", [round(rel(out["latents_all"][i], g["gligen_latents_all"][i]), 4) for i in range(5)],
              "map", rel(out["saved"][("up", 1, 1, 0)][1], g["gligen_saved_up11_step1"]), "losses", [round(t["loss"], 3) for t in tr])
