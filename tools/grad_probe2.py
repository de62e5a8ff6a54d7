"""GPU-box tool: splits the guidance-gradient error of a config into (a) the energy's selection (map gradient) and
(b) the network backward: the ORACLE's map gradients are fed into the HIP backward plan."""
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
torch.set_num_threads(min(os.cpu_count() or 1, 32))
name, L = os.environ.get("CFG", "sd21:96").split(":"); L = int(L)
keys = [KEYS[int(c)] for c in os.environ.get("KEYSET", "0")]
cfg = weights.CONFIGS[name]
sd = weights.synth_state_dict(cfg, 0)
eng = UNetEngine(cfg, dev, sd)
cd = dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
          attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps,
          gligen_positive_len=cfg.gligen_positive_len)
x = torch.randn((1, 4, L, L), generator=torch.Generator().manual_seed(0))
_, cond = weights.synth_embeddings(cfg, 1, seed=1)
rs = R.DDIM(prediction_type=cfg.prediction_type); rs.set_timesteps(50)
t = rs.timesteps[1]
kw = dict(use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0)
lat = x.clone().requires_grad_(True)
saved = {}
R.unet_forward(sd, cd, lat, t, cond, saved=saved, save_keys=keys, stop_after=keys[-1])
loss = R.compute_ca_lossv3(saved, BOXES, OBJ_POS, keys, index=1, **kw) * 30
grads = torch.autograd.grad(loss, [saved[k] for k in keys] + [lat])
g_maps_ref, g_lat_ref = grads[:-1], grads[-1]
print("oracle loss", float(loss), flush=True)
sm = LMDSampler(eng, DDIMScheduler(prediction_type=cfg.prediction_type), use_graphs=False)
tr = []
guid = dict(bboxes=BOXES, object_positions=OBJ_POS, loss_scale=30, loss_threshold=0.0, max_iter=1, max_index_step=25,
            guidance_attn_keys=keys, **kw)
sm.guidance_only(x, cond, 50, 1, guid, trace=tr)
pg = eng.plan(1, L, grad=True, fuser=False, stop_key=eng.last_key(keys), save_keys=keys, text_batch_offset=1)
def cos(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    return float(a @ b / (a.norm() * b.norm())), float((a - b).norm() / b.norm())
print("latent gradient, HIP end to end:", cos(tr[0]["grad"], g_lat_ref), flush=True)
for k, gm in zip(keys, g_maps_ref):
    m_h, m_r = pg.maps[k].float().cpu(), saved[k].detach()
    print(k, "map relerr", float((m_h - m_r).abs().max() / m_r.abs().max()), "rel-L2", float((m_h - m_r).norm() / m_r.norm()))
    gh = pg.gmaps[k].float().cpu() / sm.grad_scale
    print(k, "map-gradient cosine / rel-L2:", cos(gh, gm), "nonzeros hip/ref", int((gh != 0).sum()), int((gm != 0).sum()),
          "same support", float(((gh != 0) == (gm != 0)).float().mean()), flush=True)
# the oracle's map gradients through the HIP backward plan
for k, gm in zip(keys, g_maps_ref):
    pg.gmaps[k].copy_((gm * sm.grad_scale).to(dev))
g2 = pg.backward(sm.grad_scale).clone()
print("latent gradient, oracle map-gradients through the HIP backward:", cos(g2, g_lat_ref), flush=True)
# forward activations in front of the key
