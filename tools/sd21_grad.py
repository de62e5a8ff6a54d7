"""GPU-box tool: the SD2.1-768 guidance gradient (one latent_backward_guidance iteration, loss_scale 30 as
generation/backward_guidance.py) vs the oracle, for several internal gradient scales of the fp16 backward plan."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lgd_amd
from lgd_amd import weights
from lgd_amd.unet import UNetEngine
from lgd_amd.sampler import LMDSampler
from lgd_amd.scheduler import DDIMScheduler
import restate as R
dev = torch.device("cuda:0")
KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]
BOXES = [[0.15, 0.35, 0.5, 0.8], [0.6, 0.38, 0.98, 0.8]]
OBJ_POS = [[1, 2, 3], [5, 6, 7]]
cfg = weights.CONFIGS["sd21"]
sd = weights.synth_state_dict(cfg, 0)
eng = UNetEngine(cfg, dev, sd)
cd = dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
          attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps,
          gligen_positive_len=cfg.gligen_positive_len)
torch.set_num_threads(min(os.cpu_count() or 1, 32))
L = cfg.sample_size
x = torch.randn((1, 4, L, L), generator=torch.Generator().manual_seed(0))
_, cond = weights.synth_embeddings(cfg, 1, seed=1)
rs = R.DDIM(prediction_type=cfg.prediction_type); rs.set_timesteps(50)
tr_ref = []
R.latent_backward_guidance(sd, cd, rs, cond, 1, BOXES, OBJ_POS, rs.timesteps[1], x.clone(), torch.tensor(1e4),
                           loss_scale=30, loss_threshold=0.0, max_iter=1, max_index_step=25, guidance_attn_keys=KEYS,
                           use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0, trace=tr_ref)
b = tr_ref[0]["grad"].double().reshape(-1)
print("oracle loss", tr_ref[0]["loss"], "grad absmax", float(b.abs().max()), flush=True)
for ls in (30, 5):
    for gs in (1024.0, 128.0, 16.0, 2.0):
        sm = LMDSampler(eng, DDIMScheduler(prediction_type=cfg.prediction_type), grad_scale=gs, use_graphs=False)
        guid = dict(bboxes=BOXES, object_positions=OBJ_POS, loss_scale=ls, loss_threshold=0.0, max_iter=1, max_index_step=25,
                    guidance_attn_keys=KEYS, use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0)
        tr = []
        sm.guidance_only(x, cond, 50, 1, guid, trace=tr)
        a = tr[0]["grad"].cpu().double().reshape(-1) * (30.0 / ls)
        cos = float(a @ b / (a.norm() * b.norm()))
        print(f"loss_scale {ls:3d} grad_scale {gs:7.1f}: loss {tr[0]['loss']:.4f} cosine {cos:.6f} rel-L2 {float((a - b).norm() / b.norm()):.3e} "
              f"finite {bool(torch.isfinite(a).all())}", flush=True)
