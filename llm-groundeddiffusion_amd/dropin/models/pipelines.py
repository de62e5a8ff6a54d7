"""models/pipelines.py of the reference: the sampler entry points with their original signatures and
return tuples, executed by lgd_amd.sampler.LMDSampler (captured hipGraphs of the HIP engine).

Implemented: latent_backward_guidance :16-82, decode :117-127, generate_semantic_guidance :129-247,
gligen_enable_fuser :280-283, prepare_gligen_condition :285-321, generate_gligen :323-473,
generate_partial_frozen :541-599, and their `use_boxdiff=True` branch (:187-188, :564-565: one gradient step on the
BoxDiff energy of utils/boxdiff.py per denoising step, csrc/boxdiff.hip).  Not on the hot path and not provided:
encode / invert (DDIM inversion, unused by LMD / LMD+), generate (plain SD)."""
import numpy as np
import torch

import utils
from lgd_amd.sampler import DEFAULT_GUIDANCE_ATTN_KEYS, prepare_gligen_condition as _gligen_condition  # noqa: F401
from .models import process_input_embeddings, torch_device  # noqa: F401
from .unet_2d_condition import GatedSelfAttentionDense


def _sampler(model_dict):
    sm = model_dict.get("sampler") if isinstance(model_dict, dict) else getattr(model_dict, "sampler", None)
    if sm is None:
        raise RuntimeError("model_dict has no HIP sampler; build it with models.build_model_dict/load_synthetic/load_sd")
    return sm


def _boxdiff_dict(bboxes, object_positions, kwargs):
    """semantic_guidance_kwargs as latent_backward_guidance_boxdiff receives them (utils/boxdiff.py:199): the
    reference-attention arguments are accepted and must be inert, as generation/boxdiff.py:100-110 passes them."""
    g = dict(kwargs or {})
    if g.pop("ref_ca_saved_attns", None) is not None:
        raise NotImplementedError("BoxDiff with a reference-attention term (utils/boxdiff.py:155-167 warns that the "
                                  "original method has none) is outside the HIP path")
    for drop in ("verbose", "clear_cache", "ref_ca_word_token_only", "ref_ca_last_token_only", "word_token_indices",
                 "ref_ca_loss_weight", "cross_attention_kwargs"):
        g.pop(drop, None)
    g.update(bboxes=bboxes, object_positions=object_positions, use_boxdiff=True)
    return g


def _guidance_dict(bboxes, object_positions, kwargs):
    # `use_ratio_based_loss` travels on to the energy tables; when the caller leaves it out the reference's default
    # (True, utils/guidance.py:91) applies there, exactly as it does behind pipelines.py:48.
    g = dict(kwargs or {})
    for drop in ("verbose", "clear_cache"):
        g.pop(drop, None)
    ref = g.pop("ref_ca_saved_attns", None)
    g.update(bboxes=bboxes, object_positions=object_positions)
    return g, ref


def _ref_maps_from_saved(sm, ref, bboxes, keys, L, T):
    """ref_ca_saved_attns[obj][box?][step][key] (1,H,HW,1) -> fp32 [T][n_boxes_flat][n_keys][H][max_hw]."""
    if ref is None:
        return None
    from collections.abc import Iterable
    flat = []
    for o, per in enumerate(ref):
        flat += list(per) if isinstance(bboxes[o][0], Iterable) else [per]
    hw = sm.map_hw(L)
    heads = sm.heads_of(keys[0])
    mx = max(hw[k] for k in keys)
    out = torch.zeros((T, len(flat), len(keys), heads, mx), device=sm.dev)
    for b, steps in enumerate(flat):
        for s in range(min(T, len(steps))):
            for ki, k in enumerate(keys):
                out[s, b, ki, :, :hw[k]] = steps[s][k][0, :, :, 0].to(sm.dev).float()
    return out


def _fast_args(dynamic_num_inference_steps, fast_after_steps, fast_rate):
    """pipelines.py:151-152,217-218,358-359,439-440: the fast tail needs the step count to follow the schedule."""
    if fast_after_steps is None:
        return {}
    if not dynamic_num_inference_steps:
        raise RuntimeError("fast_after_steps without dynamic_num_inference_steps takes DDIM steps of the wrong size "
                           "(the reference only ever passes both)")
    return dict(fast_after_steps=int(fast_after_steps), fast_rate=int(fast_rate))


def _keys_to_save(sm, return_saved, keys):
    """return_saved_cross_attn with saved_cross_attn_keys=None keeps every layer (attention_processor.py:479:
    `save_keys is None or ...`)."""
    if not return_saved:
        return []
    return [tuple(k) for k in keys] if keys is not None else sm.eng.attn_key_order()


def _saved_list(saved, T):
    """{key: [T,Bp,H,HW,Tp]} -> list over steps of {key: (Bp,H,HW,Tp)} (the reference's saved_attns)."""
    return [{k: v[s] for k, v in saved.items()} for s in range(T)]


def latent_backward_guidance(scheduler, unet, cond_embeddings, index, bboxes, object_positions, t, latents, loss,
                             loss_scale=30, loss_threshold=0.2, max_iter=5, max_index_step=10,
                             cross_attention_kwargs=None, ref_ca_saved_attns=None, guidance_attn_keys=None,
                             verbose=False, clear_cache=False, model_dict=None, **kwargs):
    """pipelines.py:16-82.  `loss` carries over from the previous step (initial 1e4); returns (latents, loss)."""
    import models
    sm = _sampler(model_dict or models.model_dict)
    keys = [tuple(k) for k in (guidance_attn_keys or DEFAULT_GUIDANCE_ATTN_KEYS)]
    T = scheduler.num_inference_steps
    L = latents.shape[-1]
    g = dict(kwargs, bboxes=bboxes, object_positions=object_positions, loss_scale=loss_scale,
             loss_threshold=loss_threshold, max_iter=max_iter, max_index_step=max_index_step,
             guidance_attn_keys=keys, ref_maps=_ref_maps_from_saved(sm, ref_ca_saved_attns, bboxes, keys, L, T))
    gl = (cross_attention_kwargs or {}).get("gligen")
    gs = sm.make_guidance(L, g.pop("bboxes"), g.pop("object_positions"), **g)
    if gs is not None:
        gs.loss = float(loss)
    fuser = gl is not None and all(m.enabled for m in unet.modules())
    glt = None
    if gl is not None:
        z = lambda x: torch.cat([x, x])          # the sampler expects the [uncond; cond] pair; guidance uses item 0
        glt = (z(gl["boxes"]), z(gl["positive_embeddings"]), z(gl["masks"]))
    lat, new_loss, _ = sm.guidance_only(latents, cond_embeddings, T, index, dict(state=gs) if gs else
                                        dict(bboxes=[], object_positions=[]), gligen=glt, fuser=fuser)
    return lat.to(latents.dtype), torch.tensor(new_loss if new_loss is not None else float(loss))


@torch.no_grad()
def decode(vae, latents):
    """pipelines.py:117-127."""
    image = vae.decode(1 / 0.18215 * latents)
    image = getattr(image, "sample", image)
    image = (image / 2 + 0.5).clamp(0, 1)
    image = image.detach().float().cpu().permute(0, 2, 3, 1).numpy()
    return (image * 255).round().astype("uint8")


def gligen_enable_fuser(unet, enabled=True):
    """pipelines.py:280-283."""
    for module in unet.modules():
        if isinstance(module, GatedSelfAttentionDense):
            module.enabled = enabled


def prepare_gligen_condition(bboxes, phrases, dtype, tokenizer, text_encoder, num_images_per_prompt):
    """pipelines.py:285-321 (batch of images; CLIP pooler_output of the phrases)."""
    batch_size, max_objs = len(bboxes), 30
    assert len(phrases) == len(bboxes)
    n_objs = min(max(len(b) for b in bboxes), max_objs)
    boxes = torch.zeros((batch_size, max_objs, 4), device=torch_device, dtype=dtype)
    emb = torch.zeros((batch_size, max_objs, 768), device=torch_device, dtype=dtype)
    masks = torch.zeros((batch_size, max_objs), device=torch_device, dtype=dtype)
    if n_objs > 0:
        for idx, (bb, ph) in enumerate(zip(bboxes, phrases)):
            bb = torch.tensor(bb[:n_objs])
            boxes[idx, :bb.shape[0]] = bb
            tok = tokenizer(ph[:n_objs], padding=True, return_tensors="pt").to(torch_device)
            pe = text_encoder(**tok).pooler_output
            emb[idx, :pe.shape[0]] = pe
            assert bb.shape[0] == pe.shape[0], f"{bb.shape[0]} != {pe.shape[0]}"
            masks[idx, :bb.shape[0]] = 1
    rep = num_images_per_prompt * 2
    cond_len = batch_size * rep
    boxes, emb, masks = boxes.repeat(rep, 1, 1), emb.repeat(rep, 1, 1), masks.repeat(rep, 1)
    masks[:cond_len // 2] = 0
    return boxes, emb, masks, cond_len


def _finish(sm, model_dict, r, T, *, ret_saved, return_box_vis, bboxes, phrases, save_all_latents, offload=True):
    images = decode(model_dict.vae, r["latents"]) if model_dict.vae is not None else None
    ret = [r["latents"], images]
    if ret_saved:
        ret.append(_saved_list(r["saved"], T))
    if return_box_vis:
        ret.append([None for _ in range(1)])            # PIL box visualisation is debug-only (vis is out of scope)
    if save_all_latents:
        ret.append(r["latents_all"].cpu() if offload else r["latents_all"])
    return tuple(ret)


def generate_semantic_guidance(model_dict, latents, input_embeddings, num_inference_steps, bboxes, phrases,
                               object_positions, guidance_scale=7.5, semantic_guidance_kwargs=None,
                               return_cross_attn=False, return_saved_cross_attn=False, saved_cross_attn_keys=None,
                               return_cond_ca_only=False, return_token_ca_only=None,
                               offload_guidance_cross_attn_to_cpu=False, offload_cross_attn_to_cpu=False,
                               offload_latents_to_cpu=True, return_box_vis=False, show_progress=True,
                               save_all_latents=False, dynamic_num_inference_steps=False, fast_after_steps=None,
                               fast_rate=2, use_boxdiff=False):
    """pipelines.py:129-247 -> (latents, images[, saved_attns][, pil][, latents_all])."""
    if return_cross_attn:
        raise NotImplementedError("return_cross_attn is outside the HIP path")
    sm = _sampler(model_dict)
    text_embeddings, _, _ = input_embeddings
    T, L = num_inference_steps, latents.shape[-1]
    guid = None
    if bboxes and use_boxdiff:                               # pipelines.py:187-188
        guid = _boxdiff_dict(bboxes, object_positions, semantic_guidance_kwargs)
    elif bboxes:
        g, ref = _guidance_dict(bboxes, object_positions, semantic_guidance_kwargs)
        keys = [tuple(k) for k in (g.get("guidance_attn_keys") or DEFAULT_GUIDANCE_ATTN_KEYS)]
        g["ref_maps"] = _ref_maps_from_saved(sm, ref, bboxes, keys, L, T)
        guid = g
    r = sm.denoise(latents, text_embeddings, T, guidance_scale=guidance_scale, guidance=guid,
                   saved_cross_attn_keys=_keys_to_save(sm, return_saved_cross_attn, saved_cross_attn_keys),
                   return_cond_ca_only=return_cond_ca_only, return_token_ca_only=return_token_ca_only,
                   **_fast_args(dynamic_num_inference_steps, fast_after_steps, fast_rate))
    T = r["latents_all"].shape[0] - 1 if r["latents_all"] is not None else T       # steps actually run
    return _finish(sm, model_dict, r, T, ret_saved=return_saved_cross_attn, return_box_vis=return_box_vis, bboxes=bboxes,
                   phrases=phrases, save_all_latents=save_all_latents, offload=offload_latents_to_cpu)


@torch.no_grad()
def generate_gligen(model_dict, latents, input_embeddings, num_inference_steps, bboxes, phrases,
                    num_images_per_prompt=1, gligen_scheduled_sampling_beta: float = 0.3, guidance_scale=7.5,
                    frozen_steps=20, frozen_mask=None, return_saved_cross_attn=False, saved_cross_attn_keys=None,
                    return_cond_ca_only=False, return_token_ca_only=None, offload_cross_attn_to_cpu=False,
                    offload_latents_to_cpu=True, semantic_guidance=False, semantic_guidance_bboxes=None,
                    semantic_guidance_object_positions=None, semantic_guidance_kwargs=None, return_box_vis=False,
                    show_progress=True, save_all_latents=False, batched_condition=False,
                    dynamic_num_inference_steps=False, fast_after_steps=None, fast_rate=2):
    """pipelines.py:323-473 -> (latents, images[, saved_attns][, pil][, latents_all])."""
    if batched_condition or num_images_per_prompt != 1:
        raise NotImplementedError("batched_condition / num_images_per_prompt>1 are outside the HIP path")
    sm = _sampler(model_dict)
    text_embeddings, _, _ = process_input_embeddings(input_embeddings)
    T, L = num_inference_steps, latents.shape[-1]
    boxes, emb, masks, _ = prepare_gligen_condition([bboxes], [phrases], torch.float32, model_dict.tokenizer,
                                                    model_dict.text_encoder, 1)
    guid = None
    if semantic_guidance_bboxes and semantic_guidance:
        g, ref = _guidance_dict(semantic_guidance_bboxes, semantic_guidance_object_positions, semantic_guidance_kwargs)
        keys = [tuple(k) for k in (g.get("guidance_attn_keys") or DEFAULT_GUIDANCE_ATTN_KEYS)]
        g["ref_maps"] = _ref_maps_from_saved(sm, ref, semantic_guidance_bboxes, keys, L, T)
        guid = g
    r = sm.denoise(latents, text_embeddings, T, guidance_scale=guidance_scale, gligen=(boxes, emb, masks),
                   gligen_scheduled_sampling_beta=gligen_scheduled_sampling_beta, guidance=guid,
                   frozen_steps=frozen_steps if frozen_mask is not None else 0, frozen_mask=frozen_mask,
                   saved_cross_attn_keys=_keys_to_save(sm, return_saved_cross_attn, saved_cross_attn_keys),
                   return_cond_ca_only=return_cond_ca_only, return_token_ca_only=return_token_ca_only,
                   **_fast_args(dynamic_num_inference_steps, fast_after_steps, fast_rate))
    T = r["latents_all"].shape[0] - 1 if r["latents_all"] is not None else T       # steps actually run
    gligen_enable_fuser(model_dict.unet, False)           # pipelines.py:459-460
    return _finish(sm, model_dict, r, T, ret_saved=return_saved_cross_attn, return_box_vis=return_box_vis, bboxes=bboxes,
                   phrases=phrases, save_all_latents=save_all_latents, offload=offload_latents_to_cpu)


def generate_partial_frozen(model_dict, latents_all, frozen_mask, input_embeddings, num_inference_steps, frozen_steps,
                            guidance_scale=7.5, bboxes=None, phrases=None, object_positions=None,
                            semantic_guidance_kwargs=None, offload_guidance_cross_attn_to_cpu=False, use_boxdiff=False):
    """pipelines.py:541-599 -> (latents, images)."""
    sm = _sampler(model_dict)
    text_embeddings, _, _ = input_embeddings
    T, L = num_inference_steps, latents_all.shape[-1]
    guid = None
    if bboxes and use_boxdiff:                               # pipelines.py:564-565
        guid = _boxdiff_dict(bboxes, object_positions, semantic_guidance_kwargs)
    elif bboxes:
        g, ref = _guidance_dict(bboxes, object_positions, semantic_guidance_kwargs)
        keys = [tuple(k) for k in (g.get("guidance_attn_keys") or DEFAULT_GUIDANCE_ATTN_KEYS)]
        g["ref_maps"] = _ref_maps_from_saved(sm, ref, bboxes, keys, L, T)
        guid = g
    r = sm.denoise(latents_all, text_embeddings, T, guidance_scale=guidance_scale, guidance=guid,
                   frozen_steps=frozen_steps, frozen_mask=frozen_mask, save_all_latents=False)
    images = decode(model_dict.vae, r["latents"]) if model_dict.vae is not None else None
    return r["latents"], images
