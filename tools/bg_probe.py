"""GPU-box tool: the ratio-guided steps of tests/golden/run_backward_guidance_tiny.npz, iteration by iteration —
HIP gradient vs the oracle's gradient AT THE SAME LATENTS (single-iteration error) and the drift of the loop."""
import json, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lgd_amd
from lgd_amd import weights
from lgd_amd.unet import UNetEngine
from lgd_amd.sampler import LMDSampler
from lgd_amd.scheduler import DDIMScheduler
import restate as R
dev = torch.device("cuda:0")
KEYS = R.DEFAULT_GUIDANCE_ATTN_KEYS
cfg = weights.CONFIGS["tiny"]
sd = weights.synth_state_dict(cfg, 0)
eng = UNetEngine(cfg, dev, sd)
cd = dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
          attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps,
          gligen_positive_len=cfg.gligen_positive_len)
g = np.load(os.path.join(ROOT, "tests/golden/run_backward_guidance_tiny.npz"))
gs_list = [float(x) for x in os.environ.get("GRAD_SCALES", "1024").split(",")]
for gscale in gs_list:
  for tag in "ab":
    kw = json.loads(str(g[f"{tag}_kwargs"]))
    ehs = torch.from_numpy(g[f"{tag}_text_embeddings"])
    bb, op = json.loads(str(g[f"{tag}_bboxes"])), json.loads(str(g[f"{tag}_object_positions"]))
    rs = R.DDIM(); rs.set_timesteps(8)
    for step in range(kw["overall_max_index_step"]):
        x0 = torch.from_numpy(g[f"{tag}_starts"][step])
        gk = dict(loss_scale=kw["overall_loss_scale"], loss_threshold=kw["overall_loss_threshold"],
                  max_iter=kw["overall_max_iter"], max_index_step=kw["overall_max_index_step"], guidance_attn_keys=KEYS)
        sm = LMDSampler(eng, DDIMScheduler(), use_graphs=False, grad_scale=gscale)
        tr = []
        lat, loss, _ = sm.guidance_only(x0, ehs[1:], 8, step, dict(bboxes=bb, object_positions=op, **gk), trace=tr)
        want = torch.from_numpy(g[f"{tag}_guided"][step])
        print(f"[gs {gscale}] {tag} step {step}: guided latents relerr {float((lat.cpu() - want).abs().max() / want.abs().max()):.3e} "
              f"(update size {float((want - x0).abs().max() / want.abs().max()):.3e})")
        # single-iteration error: oracle gradient at the latents the HIP loop had before each iteration
        x = x0.clone()
        a_t = rs.alphas_cumprod[int(rs.timesteps[step])]
        for it, e in enumerate(tr):
            trr = []
            R.latent_backward_guidance(sd, cd, rs, ehs[1:], step, bb, op, rs.timesteps[step], x.clone(), torch.tensor(1e4),
                                       trace=trr, **{**gk, "max_iter": 1, "loss_threshold": 0.0})
            a, b = e["grad"].cpu().double().reshape(-1), trr[0]["grad"].double().reshape(-1)
            print(f"    it {it}: loss hip {e['loss']:.5f} oracle {trr[0]['loss']:.5f}  grad cosine {float(a @ b / (a.norm() * b.norm())):.6f} "
                  f"rel-L2 {float((a - b).norm() / b.norm()):.3e} |g| {float(b.abs().max()):.3e}")
            x = x - (1 - a_t) ** 0.5 * e["grad"].cpu().float()
