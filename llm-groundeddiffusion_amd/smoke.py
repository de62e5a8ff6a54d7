"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the
CPU oracle (oracle/restate.py — imported here only as the checker, as the tier rules allow)."""
import os
import sys
import time

import torch


def run():
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import restate as R
    from . import weights
    from .pipeline import CachedLayout, lmd_plus_generate
    from .sampler import LMDSampler, prepare_gligen_condition
    from .scheduler import DDIMScheduler
    from .unet import UNetEngine
    from .vae import make_hip_vae

    dev = torch.device("cuda:0")
    cfg = weights.CONFIGS["tiny_gligen"]
    sd = weights.synth_state_dict(cfg, 0)
    eng = UNetEngine(cfg, dev, sd)
    L = 32
    keys = R.DEFAULT_GUIDANCE_ATTN_KEYS
    boxes = [[0.15, 0.35, 0.5, 0.8], [0.6, 0.38, 0.98, 0.8]]
    pos = [[1, 2, 3], [5, 6, 7]]
    # ---- 1. one CFG UNet forward (B=2, GLIGEN fuser on, map capture) vs the oracle
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 4, L, L), generator=g)
    unc, cond = weights.synth_embeddings(cfg, 1, seed=1)
    ehs = torch.cat([unc, cond])
    pe = torch.randn((2, 768), generator=g)
    gl = prepare_gligen_condition(boxes, pe, dev)
    plan = eng.plan(2, L, fuser=True, save_keys=keys)
    eng.prepare_timesteps([501])
    eng.set_step(0)
    eng.prepare_text(ehs)
    eng.prepare_gligen(boxes=gl[0], positive_embeddings=gl[1], masks=gl[2])
    eps = plan.forward(x.to(dev)).cpu()
    cd = dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
              attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups,
              norm_eps=cfg.norm_eps, gligen_positive_len=768)
    saved = {}
    with torch.no_grad():
        ref = R.unet_forward(sd, cd, x, 501, ehs, saved=saved, save_keys=keys,
                             gligen=dict(boxes=gl[0].cpu(), positive_embeddings=gl[1].cpu(), masks=gl[2].cpu()))
    e = float((eps - ref).abs().max() / ref.abs().max())
    em = max(float((plan.maps[k].cpu() - saved[k]).abs().max() / saved[k].abs().max()) for k in keys)
    print(f"[smoke] UNet fwd (tiny_gligen, B=2, fuser on): eps relerr {e:.2e}, worst map relerr {em:.2e}")
    assert e < 2e-2 and em < 3e-2, "HIP UNet forward deviates from the oracle"
    # ---- 2. one backward-guidance iteration (energy + latent gradient) vs the oracle
    sm = LMDSampler(eng, DDIMScheduler())
    guid = dict(bboxes=boxes, object_positions=pos, loss_scale=5, loss_threshold=0.0, max_iter=1, max_index_step=10,
                guidance_attn_keys=keys, use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0)
    tr = []
    sm.guidance_only(x[:1], cond, 10, 1, guid, gligen=gl, fuser=True, trace=tr)
    rs = R.DDIM()
    rs.set_timesteps(10)
    tr_ref = []
    R.latent_backward_guidance(sd, cd, rs, cond, 1, boxes, pos, rs.timesteps[1], x[:1].clone(), torch.tensor(1e4),
                               loss_scale=5, loss_threshold=0.0, max_iter=1, max_index_step=10,
                               guidance_attn_keys=keys, use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0,
                               gligen=dict(boxes=gl[0][:1].cpu(), positive_embeddings=gl[1][:1].cpu(),
                                           masks=gl[2][:1].cpu()), trace=tr_ref)
    a, b = tr[0]["grad"].cpu().double().reshape(-1), tr_ref[0]["grad"].double().reshape(-1)
    cos = float(a @ b / (a.norm() * b.norm()))
    print(f"[smoke] guidance: loss hip {tr[0]['loss']:.4f} oracle {tr_ref[0]['loss']:.4f}, latent-grad cosine {cos:.5f}")
    assert abs(tr[0]["loss"] - tr_ref[0]["loss"]) / tr_ref[0]["loss"] < 2e-2 and cos > 0.98
    # ---- 3. a tiny end-to-end LMD+ run (2 boxes, 6 steps) just has to execute and stay finite
    sm = LMDSampler(eng, DDIMScheduler(), vae=make_hip_vae(dev))
    lay = CachedLayout.synthetic(cfg, [("a white deer", [74, 177, 183, 235]), ("a gray bear", [314, 193, 189, 216])], 3)
    t0 = time.time()
    out = lmd_plus_generate(sm, lay, num_inference_steps=6, height=8 * L, width=8 * L, overall_loss_threshold=0.0,
                            overall_max_index_step=4, overall_max_iter=[1])
    torch.cuda.synchronize()
    assert torch.isfinite(out["latents"]).all() and out["image"].shape == (8 * L, 8 * L, 3)
    print(f"[smoke] LMD+ end-to-end (tiny, 6 steps, 2 boxes): ok in {time.time() - t0:.2f}s, "
          f"{out['guidance_iters']} guidance iterations")
