"""ORACLE TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the REAL reference code
(/root/reference, unmodified, through oracle/ref_harness.py) on CPU with the seeded synthetic
weights, and checks oracle/restate.py against it on the spot.

    python oracle/make_golden.py            # build container only (needs /root/reference)

The fixtures are what pins the restatement (the reference ships no tests or golden vectors:
SURVEY.md §4, §8c); tests/test_oracle.py re-checks restate.py against them anywhere.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import lgd_amd  # noqa: E402,F401
from lgd_amd import weights  # noqa: E402
import ref_harness as rh  # noqa: E402
import restate as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
KEYS = R.DEFAULT_GUIDANCE_ATTN_KEYS
OBJ_KEY = ("down", 2, 1, 0)
L = 32
# canonical 2-box layout = demo cache entry 3, boxes xyxy in [0,1]
BOXES_XYWH = [("a white deer", [74, 177, 183, 235]), ("a gray bear", [314, 193, 189, 216])]
BBOXES = [[x / 512, y / 512, (x + w) / 512, (y + h) / 512] for _, (x, y, w, h) in BOXES_XYWH]
OBJ_POS = [[1, 2, 3], [5, 6, 7]]
WORD_TOK = [3, 7]


def key_str(k):
    return "_".join(str(x) for x in k)


def maxrel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def cfg_dict(cfg):
    return dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
                attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups,
                norm_eps=cfg.norm_eps, gligen_positive_len=cfg.gligen_positive_len)


class FakeTok(dict):
    def to(self, *_a, **_k):
        return self


class FakeTokenizer:
    def __call__(self, phrases, **kw):
        return FakeTok(n=len(phrases))


class FakeTextEncoder:
    def __init__(self, emb):
        self.emb = emb

    def __call__(self, n=None, **kw):
        class O:
            pass
        o = O()
        o.pooler_output = self.emb[:n]
        return o


def seeded(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    mods = rh.ref_modules()
    pipelines, guidance = mods["pipelines"], mods["guidance"]
    report = {}

    # ------------------------------------------------------------------ G1: UNet forward + maps
    for name in ("tiny", "tiny_gligen"):
        cfg = weights.CONFIGS[name]
        cd = cfg_dict(cfg)
        sd = weights.synth_state_dict(cfg, 0)
        md = rh.build_model_dict(cfg, 0)
        unet = md.unet
        x = seeded((2, 4, L, L), 11)
        unc, cond = weights.synth_embeddings(cfg, 1, seed=1)
        ehs = torch.cat([unc, cond])
        t = torch.tensor(501)
        saved = {}
        kw = {"save_attn_to_dict": saved, "save_keys": [OBJ_KEY, *KEYS]}
        gl = None
        if cfg.use_gated_attention:
            pe = seeded((2, 768), 3)
            tok, te = FakeTokenizer(), FakeTextEncoder(pe)
            boxes, emb, masks, _ = pipelines.prepare_gligen_condition([BBOXES], [["a", "b"]], torch.float32, tok, te, 1)
            kw["gligen"] = {"boxes": boxes, "positive_embeddings": emb, "masks": masks}
            gl = dict(boxes=boxes, positive_embeddings=emb, masks=masks)
            b2, e2, m2, _ = R.prepare_gligen_condition([BBOXES], [pe])
            assert torch.equal(b2, boxes) and torch.equal(e2, emb) and torch.equal(m2, masks)
            pipelines.gligen_enable_fuser(unet, True)
        with torch.no_grad():
            eps = unet(x, t, encoder_hidden_states=ehs, cross_attention_kwargs=kw).sample
            saved2 = {}
            eps2 = R.unet_forward(sd, cd, x, t, ehs, saved=saved2, save_keys=[OBJ_KEY, *KEYS], gligen=gl)
        report[f"{name}/eps"] = maxrel(eps2, eps)
        arrs = dict(x=x.numpy(), t=np.int64(501), ehs=ehs.numpy(), eps=eps.numpy())
        for k, v in saved.items():
            arrs["map_" + key_str(k)] = v.numpy()
            report[f"{name}/map_{key_str(k)}"] = maxrel(saved2[k], v)
        if gl is not None:
            arrs.update(gl_boxes=gl["boxes"].numpy(), gl_emb=gl["positive_embeddings"].numpy(),
                        gl_masks=gl["masks"].numpy(), phrase_emb=pe.numpy())
        np.savez_compressed(os.path.join(OUT, f"unet_fwd_{name}.npz"), **arrs)

        # -------------------------------------------------------------- G4: backward guidance call
        lat = seeded((1, 4, L, L), 21)
        sched = md.scheduler
        sched.set_timesteps(10)
        tt = sched.timesteps[1]
        gkw = dict(loss_scale=5, loss_threshold=0.0, max_iter=3, max_index_step=10, use_ratio_based_loss=False,
                   guidance_attn_keys=KEYS, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0)
        ca_kw = {"offload_cross_attn_to_cpu": False, "enable_flash_attn": False}
        if gl is not None:
            ca_kw["gligen"] = {"boxes": gl["boxes"][:1], "positive_embeddings": gl["positive_embeddings"][:1],
                               "masks": gl["masks"][:1], "fuser_attn_kwargs": {"enable_flash_attn": False}}
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with torch.enable_grad():
                lat_out, loss_out = pipelines.latent_backward_guidance(
                    sched, unet, cond, 1, BBOXES, OBJ_POS, tt, lat.clone(), torch.tensor(10000.),
                    cross_attention_kwargs=ca_kw, **gkw)
        rs = R.DDIM(cfg.prediction_type)
        rs.set_timesteps(10)
        rgl = None if gl is None else dict(boxes=gl["boxes"][:1], positive_embeddings=gl["positive_embeddings"][:1],
                                           masks=gl["masks"][:1])
        tr = []
        rkw = dict(gkw)
        lat2, loss2 = R.latent_backward_guidance(sd, cd, rs, cond, 1, BBOXES, OBJ_POS, rs.timesteps[1], lat.clone(),
                                                 torch.tensor(10000.), gligen=rgl, trace=tr, **rkw)
        report[f"{name}/guidance_latents"] = maxrel(lat2, lat_out)
        report[f"{name}/guidance_loss"] = maxrel(loss2, loss_out)
        # full-forward variant must give the same result as the early-exit one
        lat3, _ = R.latent_backward_guidance(sd, cd, rs, cond, 1, BBOXES, OBJ_POS, rs.timesteps[1], lat.clone(),
                                             torch.tensor(10000.), gligen=rgl, early_exit=False, **rkw)
        report[f"{name}/guidance_early_exit_equiv"] = maxrel(lat3, lat2)
        np.savez_compressed(os.path.join(OUT, f"guidance_{name}.npz"), latents_in=lat.numpy(), cond=cond.numpy(),
                            t=np.int64(int(tt)), latents_out=lat_out.detach().numpy(), loss_out=loss_out.detach().numpy(),
                            grad0=tr[0]["grad"].numpy(), losses=np.array([x["loss"] for x in tr], dtype=np.float32))

        # -------------------------------------------------------------- G5: sampler loops
        steps = 4
        if not cfg.use_gated_attention:
            inp = (ehs, unc, cond)
            lat_all_in = seeded((steps + 1, 1, 4, L, L), 31)
            fm = torch.zeros(L, L, dtype=torch.bool)
            fm[8:20, 4:14] = True
            sg = dict(loss_scale=5, loss_threshold=0.0, max_iter=[2, 1], max_index_step=2, use_ratio_based_loss=False,
                      guidance_attn_keys=KEYS, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0, verbose=False)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                lat_pf, _ = pipelines.generate_partial_frozen(md, lat_all_in, fm, inp, steps, 2, bboxes=BBOXES,
                                                              object_positions=OBJ_POS, semantic_guidance_kwargs=sg)
            rsg = {k: v for k, v in sg.items() if k != "verbose"}
            lat_pf2 = R.generate_partial_frozen(sd, cd, R.DDIM(), lat_all_in, fm, inp, steps, 2, bboxes=BBOXES,
                                                object_positions=OBJ_POS, semantic_guidance_kwargs=rsg)
            report[f"{name}/partial_frozen"] = maxrel(lat_pf2, lat_pf)
            lat0 = seeded((1, 4, L, L), 41)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ret = pipelines.generate_semantic_guidance(
                    md, lat0, inp, steps, BBOXES, ["a", "b"], OBJ_POS, semantic_guidance_kwargs=sg,
                    return_saved_cross_attn=True, saved_cross_attn_keys=[OBJ_KEY, *KEYS], return_cond_ca_only=True,
                    return_token_ca_only=3, save_all_latents=True, show_progress=False)
            lat_sg, _, saved_sg, lat_all_sg = ret
            lat_sg2, saved_sg2, lat_all_sg2 = R.generate_semantic_guidance(
                sd, cd, R.DDIM(), lat0, inp, steps, BBOXES, OBJ_POS, semantic_guidance_kwargs=rsg,
                saved_cross_attn_keys=[OBJ_KEY, *KEYS], return_cond_ca_only=True, return_token_ca_only=3)
            report[f"{name}/semantic_guidance_latents_all"] = maxrel(lat_all_sg2, lat_all_sg)
            report[f"{name}/semantic_guidance_saved"] = max(
                maxrel(saved_sg2[s][k], saved_sg[s][k]) for s in range(steps) for k in saved_sg[s])
            np.savez_compressed(os.path.join(OUT, f"loops_{name}.npz"), ehs=ehs.numpy(), lat_all_in=lat_all_in.numpy(),
                                frozen_mask=fm.numpy(), partial_frozen_out=lat_pf.detach().numpy(), lat0=lat0.numpy(),
                                sg_latents_all=lat_all_sg.numpy(),
                                sg_saved_last=np.stack([saved_sg[-1][k].numpy()[0, :, :, 0].mean(0).reshape(-1)[:16]
                                                        for k in KEYS]),
                                sg_saved_up11_step0=saved_sg[0][("up", 1, 1, 0)].numpy())
        else:
            pe = seeded((2, 768), 3)
            md.tokenizer, md.text_encoder = FakeTokenizer(), FakeTextEncoder(pe)
            inp = (ehs, unc, cond)
            lat_all_in = seeded((steps + 1, 1, 4, L, L), 31)
            fm = torch.zeros(L, L, dtype=torch.bool)
            fm[8:20, 4:14] = True
            sg = dict(loss_scale=5, loss_threshold=0.0, max_iter=[2, 1], max_index_step=3, use_ratio_based_loss=False,
                      guidance_attn_keys=KEYS, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0, verbose=False)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ret = pipelines.generate_gligen(
                    md, lat_all_in, inp, steps, BBOXES, ["a", "b"], gligen_scheduled_sampling_beta=0.5,
                    frozen_steps=2, frozen_mask=fm, return_saved_cross_attn=True, saved_cross_attn_keys=[OBJ_KEY, *KEYS],
                    return_cond_ca_only=True, return_token_ca_only=7, semantic_guidance=True,
                    semantic_guidance_bboxes=BBOXES, semantic_guidance_object_positions=OBJ_POS,
                    semantic_guidance_kwargs=sg, save_all_latents=True, show_progress=False)
            lat_g, _, saved_g, lat_all_g = ret
            rsg = {k: v for k, v in sg.items() if k != "verbose"}
            lat_g2, saved_g2, lat_all_g2 = R.generate_gligen(
                sd, cd, R.DDIM(), lat_all_in, inp, steps, BBOXES, pe, gligen_scheduled_sampling_beta=0.5,
                frozen_steps=2, frozen_mask=fm, return_saved_cross_attn=True, saved_cross_attn_keys=[OBJ_KEY, *KEYS],
                return_cond_ca_only=True, return_token_ca_only=7, semantic_guidance=True,
                semantic_guidance_bboxes=BBOXES, semantic_guidance_object_positions=OBJ_POS,
                semantic_guidance_kwargs=rsg)
            report[f"{name}/gligen_latents_all"] = maxrel(lat_all_g2, lat_all_g)
            report[f"{name}/gligen_saved"] = max(maxrel(saved_g2[s][k], saved_g[s][k]) for s in range(steps) for k in saved_g[s])
            np.savez_compressed(os.path.join(OUT, f"loops_{name}.npz"), ehs=ehs.numpy(), lat_all_in=lat_all_in.numpy(),
                                frozen_mask=fm.numpy(), phrase_emb=pe.numpy(), gligen_latents_all=lat_all_g.numpy(),
                                gligen_saved_up11_step1=saved_g[1][("up", 1, 1, 0)].numpy())
        del md, unet

    # ------------------------------------------------------------------ G3: energy + map gradients
    maps = {}
    for i, k in enumerate(KEYS):
        hw = 64 if k[0] == "mid" else 256
        maps[k] = (seeded((1, 8, hw, 77), 50 + i, 2.0)).softmax(-1).requires_grad_(True)
    refs = []  # [obj][step][key] -> (1, heads, hw, 1)
    for o in range(2):
        refs.append([{k: seeded((1, 8, maps[k].shape[2], 1), 70 + o * 4 + i).abs() * 0.05
                      for i, k in enumerate(KEYS)} for _ in range(2)])
    ekw = dict(fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0, use_ratio_based_loss=False)
    arrs = {f"map_{key_str(k)}": v.detach().numpy() for k, v in maps.items()}
    for tag, rkw in (("noref", {}), ("ref", dict(ref_ca_saved_attns=refs, ref_ca_word_token_only=True,
                                                 ref_ca_last_token_only=True, word_token_indices=WORD_TOK,
                                                 index=1, ref_ca_loss_weight=2.0))):
        loss = guidance.compute_ca_lossv3(saved_attn=maps, bboxes=BBOXES, object_positions=OBJ_POS,
                                          guidance_attn_keys=KEYS, **rkw, **ekw)
        grads = torch.autograd.grad(loss, [maps[k] for k in KEYS])
        rk = {**rkw, **ekw}
        loss2 = R.compute_ca_lossv3(maps, BBOXES, OBJ_POS, KEYS, **rk)
        grads2 = torch.autograd.grad(loss2, [maps[k] for k in KEYS])
        report[f"energy/{tag}/loss"] = maxrel(loss2, loss)
        report[f"energy/{tag}/grad"] = max(maxrel(a, b) for a, b in zip(grads2, grads))
        arrs[f"loss_{tag}"] = loss.detach().numpy()
        for k, g in zip(KEYS, grads):
            arrs[f"grad_{tag}_{key_str(k)}"] = g.numpy()
    for o in range(2):
        for i, k in enumerate(KEYS):
            arrs[f"ref_{o}_{key_str(k)}"] = refs[o][1][k].numpy()
    np.savez_compressed(os.path.join(OUT, "energy.npz"), **arrs)

    # ------------------------------------------------------------------ G6: host-side latent prep
    latents_mod = mods["latents"]
    from easydict import EasyDict

    class U:
        class config:
            in_channels = 4
    from diffusers import DDIMScheduler
    md = EasyDict(unet=U(), scheduler=DDIMScheduler(), dtype=torch.float32)
    lst, bg = latents_mod.get_input_latents_list(md, bg_seed=3, fg_seed_start=3 + 123456789, so_boxes=BBOXES,
                                                 fg_blending_ratio=0.1, height=512, width=512)
    lst2, bg2 = R.get_input_latents_list(3, 3 + 123456789, BBOXES, 0.1)
    report["latents/input_list"] = max(maxrel(a, b) for a, b in zip(lst2 + [bg2], lst + [bg]))
    steps = 3
    lall = [seeded((steps + 1, 1, 4, 64, 64), 90 + i) for i in range(2)]
    masks = [R.proportion_to_mask(b, 64, 64).bool() for b in BBOXES]
    comp, fg, _off = latents_mod.compose_latents_with_alignment(md, lall, masks, steps, 1, 512, 512, latents_bg=bg,
                                                          align_with_overall_bboxes=False, overall_bboxes=None)
    comp2, fg2 = R.compose_latents(lall, masks, steps, bg2)
    report["latents/compose"] = maxrel(comp2, comp)
    report["latents/fg_indices_equal"] = float(not torch.equal(fg, fg2))
    np.savez_compressed(os.path.join(OUT, "latents_host.npz"), bg=bg.numpy(), in0=lst[0].numpy(), in1=lst[1].numpy(),
                        composed=comp.numpy(), fg_idx=fg.numpy(), lall0=lall[0].numpy(), lall1=lall[1].numpy())

    print("restate.py vs REAL reference (max relative error):")
    worst = 0.0
    for k, v in report.items():
        print(f"  {k:45s} {v:.3e}")
        worst = max(worst, v)
    with open(os.path.join(OUT, "PINNING.txt"), "w") as fh:
        fh.write("oracle/restate.py vs the reference's own code (oracle/make_golden.py), max relative error\n")
        for k, v in report.items():
            fh.write(f"{k} {v:.3e}\n")
    assert worst < 2e-4, "restatement deviates from the reference"


if __name__ == "__main__":
    main()
