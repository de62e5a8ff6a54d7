"""GPU-box tool: where does the guidance gradient of a config deviate from the oracle?  One latent_backward_guidance
iteration with the loss restricted to ONE guidance key at a time (the gradient then flows only through the part of the
network in front of that key), HIP plan vs oracle autograd, for several configs / latent sizes."""
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
for name, L in [(c, int(l)) for c, l in (x.split(":") for x in os.environ.get("CFGS", "tiny:32,tiny_sd21:32,tiny_sd21:48,tiny:48").split(","))]:
    cfg = weights.CONFIGS[name]
    sd = weights.synth_state_dict(cfg, 0)
    eng = UNetEngine(cfg, dev, sd)
    cd = dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
              attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps,
              gligen_positive_len=cfg.gligen_positive_len)
    x = torch.randn((1, 4, L, L), generator=torch.Generator().manual_seed(0))
    _, cond = weights.synth_embeddings(cfg, 1, seed=1)
    sel = os.environ.get("KEYSETS", "0;1;2;3;0123").split(";")
    for keys in [[KEYS[int(c)] for c in ks_] for ks_ in sel]:
        rs = R.DDIM(prediction_type=cfg.prediction_type); rs.set_timesteps(50)
        tr_ref, tr = [], []
        kw = dict(loss_scale=30, loss_threshold=0.0, max_iter=1, max_index_step=25, guidance_attn_keys=keys,
                  use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0)
        R.latent_backward_guidance(sd, cd, rs, cond, 1, BOXES, OBJ_POS, rs.timesteps[1], x.clone(), torch.tensor(1e4),
                                   trace=tr_ref, **kw)
        sm = LMDSampler(eng, DDIMScheduler(prediction_type=cfg.prediction_type), use_graphs=False)
        sm.guidance_only(x, cond, 50, 1, dict(bboxes=BOXES, object_positions=OBJ_POS, **kw), trace=tr)
        a, b = tr[0]["grad"].cpu().double().reshape(-1), tr_ref[0]["grad"].double().reshape(-1)
        cos = float(a @ b / (a.norm() * b.norm()))
        print(f"{name} L={L} keys={[k[0] + str(k[2]) for k in keys]}: loss {tr[0]['loss']:.4f}/{tr_ref[0]['loss']:.4f} "
              f"cosine {cos:.6f} rel-L2 {float((a - b).norm() / b.norm()):.3e}", flush=True)
    del eng
    torch.cuda.empty_cache()
