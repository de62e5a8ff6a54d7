"""Stage-2 generation of one cached layout on the HIP engine: LMD+ (generation/lmd_plus.py:193-520),
training-free LMD (generation/lmd.py:215-551) and the backward-guidance baseline
(generation/backward_guidance.py).

"Cached layout" = everything stage 1 and the text encoder produce for a prompt, computed ahead of
the hot path: boxes, CLIP hidden states of the per-box / overall / negative
prompts, pooled phrase embeddings for GLIGEN, token positions of the phrases.  SAM mask refinement
(models/sam.py, §8f rank 2) runs when a `mask_refiner` (lgd_amd.sam_refine.SamRefiner over the HIP SAM model) is passed;
without one — benchmarks, no SAM weights — the box mask `proportion_to_mask` stands in.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .sampler import DEFAULT_GUIDANCE_ATTN_KEYS, Job, LMDSampler, prepare_gligen_condition
from .weights import UNetConfig

F32 = torch.float32
OBJ_ATTN_KEY = ("down", 2, 1, 0)                     # generation/lmd_plus.py:380
DEFAULT_MAX_ITER = [4] * 5 + [3] * 5 + [2] * 5 + [2] * 5 + [1] * 10   # lmd_plus.py:202


def convert_box(box, height=512, width=512):
    """utils/parse.py:302-309: (x, y, w, h) in pixels -> (x_min, y_min, x_max, y_max) in [0,1]."""
    x_min, y_min = box[0] / width, box[1] / height
    return x_min, y_min, x_min + box[2] / width, y_min + box[3] / height


@dataclass
class CachedLayout:
    """One prompt's stage-1 + text-encoder products (order follows parse.convert_spec: boxes sorted
    by object name; `overall_groups[g]` lists the box indices that share phrase g)."""
    boxes: List[Tuple[float, float, float, float]]          # xyxy in [0,1], per box
    so_uncond: torch.Tensor                                  # (1,77,Cx) negative prompt (per-box stage)
    so_cond: torch.Tensor                                    # (N,77,Cx) per-box prompts
    so_object_positions: List[List[int]]                     # token positions of the phrase in so prompt i
    so_word_token_index: List[int]
    overall_uncond: torch.Tensor                             # (1,77,Cx)
    overall_cond: torch.Tensor                               # (1,77,Cx)
    overall_groups: List[List[int]]
    overall_object_positions: List[List[int]]                # per group
    overall_word_token_indices: List[int]                    # per group
    phrase_embeddings: torch.Tensor                          # (N,768) CLIP pooler_output per box phrase
    bg_seed: int = 0
    fg_seed_start: int = 20

    def __post_init__(self):
        # The reference pairs the f-th per-box generation with the f-th box of the flattened overall list
        # (lmd_plus.py:441-456, latents.py:85-118): that only works because parse.convert_spec sorts boxes by
        # name and groups them with np.unique.  Everything downstream relies on the same invariant.
        flat = [i for grp in self.overall_groups for i in grp]
        if flat != list(range(len(self.boxes))):
            raise ValueError(f"overall_groups must list the boxes in order (got {self.overall_groups})")

    @property
    def n_boxes(self):
        return len(self.boxes)

    @staticmethod
    def synthetic(cfg: UNetConfig, gen_boxes, index: int = 0, height=512, width=512) -> "CachedLayout":
        """Synthetic text side for a real cached layout: seeded random CLIP states,
        object o occupies tokens [1+4o, 2+4o, 3+4o] (word = last), seeds as generate.py:226-229."""
        gen_boxes = sorted(gen_boxes, key=lambda gb: gb[0])                       # parse.py:315
        names = [n for n, _ in gen_boxes]
        boxes = [convert_box(b, height, width) for _, b in gen_boxes]
        uniq = sorted(set(names))                                                 # np.unique order
        groups = [[i for i, n in enumerate(names) if n == u] for u in uniq]
        cx = cfg.cross_attention_dim
        g = lambda seed, shape: torch.randn(shape, generator=torch.Generator().manual_seed(seed))
        n = len(boxes)
        return CachedLayout(
            boxes=boxes,
            so_uncond=g(7000 + index, (1, 77, cx)), so_cond=g(7100 + index, (max(n, 1), 77, cx))[:n],
            so_object_positions=[[1, 2, 3] for _ in range(n)], so_word_token_index=[3] * n,
            overall_uncond=g(7200 + index, (1, 77, cx)), overall_cond=g(7300 + index, (1, 77, cx)),
            overall_groups=groups,
            overall_object_positions=[[1 + 4 * o, 2 + 4 * o, 3 + 4 * o] for o in range(len(groups))],
            overall_word_token_indices=[3 + 4 * o for o in range(len(groups))],
            phrase_embeddings=g(7400 + index, (max(n, 1), cfg.gligen_positive_len))[:n],
            bg_seed=index, fg_seed_start=index + 123456789)                      # generate.py:226-229,317-344


from .hostprep import (align_with_bboxes, compose as compose_latents, get_centered_box,
                       input_latents_list as _input_latents_list, proportion_to_mask, seeded_noise, shift_tensor)


def get_input_latents_list(bg_seed, fg_seed_start, so_boxes, fg_blending_ratio, in_channels=4, H=64, W=64):
    return _input_latents_list(bg_seed, fg_seed_start, so_boxes, fg_blending_ratio, in_channels, H, W)


# -------------------------------------------------------------------------------------------------
def _ref_maps(sampler: LMDSampler, saved_list, keys, L, T):
    """Stage-A maps R_b (guidance.py:201) -> fp32 [T][n_boxes][n_keys][heads][max_hw]."""
    hw = sampler.map_hw(L)
    heads = sampler.heads_of(keys[0])
    max_hw = max(hw[k] for k in keys)
    out = torch.zeros((T, len(saved_list), len(keys), heads, max_hw), device=sampler.dev, dtype=F32)
    for b, saved in enumerate(saved_list):
        for ki, k in enumerate(keys):
            n = min(T, saved[k].shape[0])                            # a fast-schedule stage A ran fewer steps
            out[:n, b, ki, :, :hw[k]] = saved[k][:n, 0, :, :, 0]     # [T,1,H,HW,1] (cond only, word token)
    return out


def _centered_so_boxes(lay, so_center_box, **centered_box_kwargs):
    """lmd.py:314-324 / lmd_plus.py:286-297: the per-box generations may run on a centred copy of each box
    (the overall generation keeps the original boxes)."""
    if not so_center_box:
        return [list(b) for b in lay.boxes]
    return [get_centered_box(list(b), **centered_box_kwargs) for b in lay.boxes]


def _align_stage_a(d, lay, keys, align_with_overall_bboxes, horizontal_shift_only):
    """latents.py:85-118 + attn.py:40-70 (lmd.py:489-497): shift every per-box history, its mask and its
    saved reference maps so that the mask's centre of mass lands on the centre of the box it will occupy
    in the overall generation (offsets quantised on the 8x8 grid, zero fill)."""
    if not (align_with_overall_bboxes and d["latents_all"]):
        return
    flat = [i for grp in lay.overall_groups for i in grp]                    # utils.expand_overall_bboxes order
    targets = [list(lay.boxes[i]) for i in flat]
    d["latents_all"], d["masks"], offsets = align_with_bboxes(d["latents_all"], d["masks"], targets,
                                                              horizontal_shift_only=horizontal_shift_only)
    for saved, (x_off, y_off) in zip(d["saved"], offsets):
        for k in keys:
            if k not in saved:
                continue
            m = saved[k]                                                      # [T, Bp, heads, HW, Tp]
            side = int(round(m.shape[3] ** 0.5))
            m = shift_tensor(m.unflatten(3, (side, side)), x_off, y_off, offset_normalized=True, ignore_last_dim=True)
            saved[k] = m.flatten(3, 4)


def _token_attn(saved, start):
    """utils/attn.py:9-38 (get_token_attnv2 with input_ca_has_condition_only, one saved token): mean over heads and over
    the steps from `start` on of the object token's map at OBJ_ATTN_KEY -> numpy [side, side]."""
    m = saved[OBJ_ATTN_KEY][start:, 0, :, :, 0]
    if m.shape[0] == 0:
        raise RuntimeError(f"no saved attention at or after step {start}")
    mean = m.float().mean(dim=(0, 1))
    side = int(round(mean.numel() ** 0.5))
    return mean.reshape(side, side).cpu().numpy()


def _so_mask(refiner, kind, image, box, L, token_attn=None):
    """Foreground mask of one single-object generation on the latent grid.  With a `mask_refiner`
    (lgd_amd.sam_refine.SamRefiner) it is SAM's refinement of the decoded image, prompted by the layout box (LMD+,
    generation/lmd_plus.py:122-130) or by the object token's attention map (LMD, generation/lmd.py:124-149); without one
    it is the box itself — the stand-in SURVEY.md §8d prescribes for benchmarks without SAM weights."""
    if refiner is None:
        return proportion_to_mask(box, L, L).bool()
    if image is None:
        raise RuntimeError("mask refinement needs the decoded single-object images (decode=True)")
    mask, _conf = refiner.box(image, box) if kind == "box" else refiner.attn(image, token_attn)
    return torch.as_tensor(mask).bool()


def lmd_plus_generate(sampler: LMDSampler, lay: CachedLayout, **kw):
    """LMD+ for one layout (generation/lmd_plus.py:193-520 with its default arguments)."""
    return lmd_plus_generate_batch(sampler, [lay], **kw)[0]


def lmd_plus_generate_batch(sampler: LMDSampler, lays: List[CachedLayout], *, num_inference_steps=50,
                            frozen_step_ratio=0.5, guidance_scale=7.5, loss_scale=5, loss_threshold=5.0,
                            max_iter=None, max_index_step=0, so_gligen_scheduled_sampling_beta=0.4,
                            overall_gligen_scheduled_sampling_beta=0.4, overall_loss_scale=5,
                            overall_loss_threshold=5.0, overall_max_iter=None, overall_max_index_step=30,
                            overall_fg_top_p=0.2, overall_bg_top_p=0.2, overall_fg_weight=1.0,
                            overall_bg_weight=4.0, ref_ca_loss_weight=2.0, fg_blending_ratio=0.1,
                            use_ref_ca=True, height=512, width=512, decode=True, guidance_attn_keys=None,
                            use_fast_schedule=False, so_center_box=False, so_horizontal_center_only=True,
                            align_with_overall_bboxes=False, horizontal_shift_only=True, mask_refiner=None,
                            overall_first_step=0, overall_n_steps=None, overall_start=None):
    """LMD+ (generation/lmd_plus.py:193-520; per-box attention guidance is off by default there,
    max_index_step=0, :203 — with max_index_step > 0 every per-box GLIGEN generation is guided on its own box with
    the energy's default weights, lmd_plus.py:320-328) for a batch of independent layouts: the per-box generations of ALL layouts run
    as one batched denoising call (B = 2 x total boxes), then the overall generations of all layouts as
    another (B = 2 x layouts; guidance pass B = layouts with a per-image loop exit).  Results per layout
    are independent of how layouts are batched (images only share kernel launches).
    overall_first_step / overall_n_steps / overall_start (one latent tensor per layout): run only those steps of the
    overall generation, from the given state — the teacher-forcing hook of the parity tests."""
    L = height // 8
    T = num_inference_steps
    frozen_steps = int(T * min(max(frozen_step_ratio, 0.0), 1.0))
    keys = [tuple(k) for k in (guidance_attn_keys or DEFAULT_GUIDANCE_ATTN_KEYS)]
    dev = sampler.dev
    C = sampler.eng.cfg.in_channels
    so_boxes = [_centered_so_boxes(lay, so_center_box, horizontal_center_only=so_horizontal_center_only)
                for lay in lays]
    prep = [get_input_latents_list(lay.bg_seed, lay.fg_seed_start, so_boxes[li], fg_blending_ratio, C, L, L)
            for li, lay in enumerate(lays)]
    # use_fast_schedule (lmd_plus.py:360-367): the per-box generations only feed SAM after the steps needed
    # for latent / attention transfer, so the rest runs on every second timestep.
    fast_after = (max(frozen_steps, overall_max_index_step) if use_ref_ca else frozen_steps) if use_fast_schedule else None
    comp_steps = fast_after if use_fast_schedule else T                       # latents.py:46-48,77-78
    # ---- stage A: one GLIGEN generation per box (lmd_plus.py:44-145,162-188), all boxes batched
    jobs, owner = [], []
    if use_ref_ca or frozen_steps > 0:
        for li, lay in enumerate(lays):
            for i, box in enumerate(so_boxes[li]):
                guid = None
                if max_index_step > 0:                       # lmd_plus.py:320-328 (semantic_guidance_kwargs)
                    guid = dict(bboxes=[list(box)], object_positions=[lay.so_object_positions[i]],
                                loss_scale=loss_scale, loss_threshold=loss_threshold,
                                max_iter=max_iter or DEFAULT_MAX_ITER, max_index_step=max_index_step,
                                use_ratio_based_loss=False, guidance_attn_keys=keys)
                jobs.append(Job(prep[li][0][i], torch.cat([lay.so_uncond, lay.so_cond[i:i + 1]]),
                                gligen=prepare_gligen_condition([list(box)], lay.phrase_embeddings[i:i + 1], dev),
                                guidance=guid, token=lay.so_word_token_index[i]))
                owner.append((li, i))
    res_a = sampler.denoise_batch(jobs, T, guidance_scale=guidance_scale, use_gligen=True,
                                  gligen_scheduled_sampling_beta=so_gligen_scheduled_sampling_beta,
                                  saved_cross_attn_keys=[OBJ_ATTN_KEY, *keys] if use_ref_ca else [OBJ_ATTN_KEY],
                                  return_cond_ca_only=True, fast_after_steps=fast_after) if jobs else []
    per_lay = [dict(latents_all=[], masks=[], saved=[], so_images=[]) for _ in lays]
    if decode and res_a:
        imgs = sampler.decode(torch.cat([r["latents"] for r in res_a]))      # feeds SAM in the reference
    for n, ((li, i), r) in enumerate(zip(owner, res_a)):
        d = per_lay[li]
        d["latents_all"].append(r["latents_all"])
        d["saved"].append(r["saved"])
        d["masks"].append(_so_mask(mask_refiner, "box", imgs[n] if decode else None, so_boxes[li][i], L))
        if decode:
            d["so_images"].append(imgs[n:n + 1])
    # ---- composition (lmd_plus.py:398-416) and stage B: overall generation with attention guidance
    jobs_b, comps = [], []
    for li, lay in enumerate(lays):
        d = per_lay[li]
        _align_stage_a(d, lay, keys, align_with_overall_bboxes, horizontal_shift_only)
        composed, fg_idx = compose_latents(d["latents_all"], d["masks"], comp_steps, prep[li][1].to(dev))
        comps.append((composed, fg_idx))
        overall_bboxes = [[list(lay.boxes[i]) for i in grp] for grp in lay.overall_groups]
        flat = [i for grp in lay.overall_groups for i in grp]
        guid = None
        if overall_bboxes:
            guid = dict(bboxes=overall_bboxes, object_positions=lay.overall_object_positions,
                        loss_scale=overall_loss_scale, loss_threshold=overall_loss_threshold,
                        max_iter=overall_max_iter or DEFAULT_MAX_ITER, max_index_step=overall_max_index_step,
                        fg_top_p=overall_fg_top_p, bg_top_p=overall_bg_top_p, fg_weight=overall_fg_weight,
                        bg_weight=overall_bg_weight, ref_ca_word_token_only=True, ref_ca_last_token_only=True,
                        word_token_indices=lay.overall_word_token_indices, ref_ca_loss_weight=ref_ca_loss_weight,
                        use_ratio_based_loss=False, guidance_attn_keys=keys,   # lmd_plus.py:487 / lmd.py:521
                        ref_maps=_ref_maps(sampler, d["saved"], keys, L, T) if use_ref_ca else None)
        gl = prepare_gligen_condition([list(lay.boxes[i]) for i in flat], lay.phrase_embeddings[flat], dev)
        start = composed
        if overall_start is not None:                     # rows 1.. still feed the frozen-mask blend
            start = composed.clone()
            start[0] = torch.as_tensor(overall_start[li]).to(start.device, start.dtype)
        jobs_b.append(Job(start, torch.cat([lay.overall_uncond, lay.overall_cond]), gligen=gl, guidance=guid,
                          frozen_mask=(fg_idx != 0)))
    res_b = sampler.denoise_batch(jobs_b, T, guidance_scale=guidance_scale, use_gligen=True,
                                  gligen_scheduled_sampling_beta=overall_gligen_scheduled_sampling_beta,
                                  frozen_steps=frozen_steps, save_all_latents=False, first_step=overall_first_step,
                                  n_steps=overall_n_steps)
    images = sampler.decode(torch.cat([r["latents"] for r in res_b])) if decode else [None] * len(lays)
    return [dict(image=images[li], latents=res_b[li]["latents"], so_images=per_lay[li]["so_images"],
                 guidance_iters=res_b[li]["guidance_iters"],
                 guidance_iters_fuser_on=res_b[li]["guidance_iters_fuser_on"], composed=comps[li][0],
                 fg_idx=comps[li][1],
                 so_latents_all=per_lay[li]["latents_all"],
                 so_guidance_iters=[r["guidance_iters"] for (lj, _), r in zip(owner, res_a) if lj == li])
            for li in range(len(lays))]


def lmd_generate(sampler: LMDSampler, lay: CachedLayout, **kw):
    """Training-free LMD for one layout (generation/lmd.py:215-551, defaults :215-256)."""
    return lmd_generate_batch(sampler, [lay], **kw)[0]


def lmd_generate_batch(sampler: LMDSampler, lays: List[CachedLayout], *, num_inference_steps=50,
                       frozen_step_ratio=0.4, guidance_scale=7.5, loss_scale=5, loss_threshold=5.0, max_iter=None,
                       max_index_step=30, overall_loss_scale=5, overall_loss_threshold=5.0, overall_max_iter=None,
                       overall_max_index_step=30, fg_top_p=0.2, bg_top_p=0.2, overall_fg_top_p=0.2,
                       overall_bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0, overall_fg_weight=1.0,
                       overall_bg_weight=4.0, ref_ca_loss_weight=2.0, fg_blending_ratio=0.01, use_ref_ca=True,
                       height=512, width=512, decode=True, guidance_attn_keys=None, use_fast_schedule=False,
                       so_center_box=False, so_horizontal_center_only=False, so_vertical_placement="floor_padding",
                       so_floor_padding=0.2, align_with_overall_bboxes=False, horizontal_shift_only=False,
                       mask_refiner=None, attn_aggregation_step_start=10):
    """Training-free LMD (generation/lmd.py:215-551) for a batch of independent layouts: per-box stage =
    generate_semantic_guidance WITH guidance (lmd.py:340-352), all boxes of all layouts in one batched
    denoising call (each image keeps its own guidance loop exit); overall stage = generate_partial_frozen
    with the reference-attention term (lmd.py:530-542), all layouts in another."""
    L = height // 8
    T = num_inference_steps
    frozen_steps = int(T * min(max(frozen_step_ratio, 0.0), 1.0))
    keys = [tuple(k) for k in (guidance_attn_keys or DEFAULT_GUIDANCE_ATTN_KEYS)]
    dev = sampler.dev
    C = sampler.eng.cfg.in_channels
    so_boxes = [_centered_so_boxes(lay, so_center_box, horizontal_center_only=so_horizontal_center_only,
                                   vertical_placement=so_vertical_placement, floor_padding=so_floor_padding)
                for lay in lays]                                              # lmd.py:314-324
    prep = [get_input_latents_list(lay.bg_seed, lay.fg_seed_start, so_boxes[li], fg_blending_ratio, C, L, L)
            for li, lay in enumerate(lays)]
    fast_after = (max(frozen_steps, overall_max_index_step) if use_ref_ca else frozen_steps) \
        if use_fast_schedule else None                                        # lmd.py:399-406
    comp_steps = fast_after if use_fast_schedule else T
    # ---- stage A
    jobs, owner = [], []
    for li, lay in enumerate(lays):
        for i, box in enumerate(so_boxes[li]):
            guid = dict(bboxes=[list(box)], object_positions=[lay.so_object_positions[i]], loss_scale=loss_scale,
                        loss_threshold=loss_threshold, max_iter=max_iter or DEFAULT_MAX_ITER,
                        max_index_step=max_index_step, fg_top_p=fg_top_p, bg_top_p=bg_top_p, fg_weight=fg_weight,
                        bg_weight=bg_weight, use_ratio_based_loss=False, guidance_attn_keys=keys)   # lmd.py:349
            jobs.append(Job(prep[li][0][i], torch.cat([lay.so_uncond, lay.so_cond[i:i + 1]]), guidance=guid,
                            token=lay.so_word_token_index[i]))
            owner.append((li, i))
    res_a = sampler.denoise_batch(jobs, T, guidance_scale=guidance_scale,
                                  saved_cross_attn_keys=[OBJ_ATTN_KEY, *keys], return_cond_ca_only=True,
                                  fast_after_steps=fast_after) if jobs else []
    per_lay = [dict(latents_all=[], masks=[], saved=[], so_images=[]) for _ in lays]
    if decode and res_a:
        imgs = sampler.decode(torch.cat([r["latents"] for r in res_a]))
    for n, ((li, i), r) in enumerate(zip(owner, res_a)):
        d = per_lay[li]
        d["latents_all"].append(r["latents_all"])
        d["saved"].append(r["saved"])
        d["masks"].append(_so_mask(mask_refiner, "attn", imgs[n] if decode else None, so_boxes[li][i], L,
                                  token_attn=_token_attn(r["saved"], attn_aggregation_step_start) if mask_refiner else None))
        if decode:
            d["so_images"].append(imgs[n:n + 1])
    # ---- alignment (latents.py:107-118), composition, stage B
    jobs_b, comps = [], []
    for li, lay in enumerate(lays):
        d = per_lay[li]
        _align_stage_a(d, lay, keys, align_with_overall_bboxes, horizontal_shift_only)
        composed, fg_idx = compose_latents(d["latents_all"], d["masks"], comp_steps, prep[li][1].to(dev))
        comps.append((composed, fg_idx))
        overall_bboxes = [[list(lay.boxes[i]) for i in grp] for grp in lay.overall_groups]
        flat = [i for grp in lay.overall_groups for i in grp]
        guid = None
        if overall_bboxes:
            guid = dict(bboxes=overall_bboxes, object_positions=lay.overall_object_positions,
                        loss_scale=overall_loss_scale, loss_threshold=overall_loss_threshold,
                        max_iter=overall_max_iter or DEFAULT_MAX_ITER, max_index_step=overall_max_index_step,
                        fg_top_p=overall_fg_top_p, bg_top_p=overall_bg_top_p, fg_weight=overall_fg_weight,
                        bg_weight=overall_bg_weight, ref_ca_word_token_only=True, ref_ca_last_token_only=True,
                        word_token_indices=lay.overall_word_token_indices, ref_ca_loss_weight=ref_ca_loss_weight,
                        use_ratio_based_loss=False, guidance_attn_keys=keys,   # lmd_plus.py:487 / lmd.py:521
                        ref_maps=_ref_maps(sampler, d["saved"], keys, L, T) if use_ref_ca else None)
        jobs_b.append(Job(composed, torch.cat([lay.overall_uncond, lay.overall_cond]), guidance=guid,
                          frozen_mask=(fg_idx != 0)))
    res_b = sampler.denoise_batch(jobs_b, T, guidance_scale=guidance_scale, frozen_steps=frozen_steps,
                                  save_all_latents=False)
    images = sampler.decode(torch.cat([r["latents"] for r in res_b])) if decode else [None] * len(lays)
    return [dict(image=images[li], latents=res_b[li]["latents"], so_images=per_lay[li]["so_images"],
                 guidance_iters=res_b[li]["guidance_iters"], composed=comps[li][0], fg_idx=comps[li][1],
                 so_guidance_iters=[r["guidance_iters"] for (lj, _), r in zip(owner, res_a) if lj == li])
            for li in range(len(lays))]


def backward_guidance_generate(sampler: LMDSampler, lay: CachedLayout, **kw):
    """Layout-guidance baseline for one layout (generation/backward_guidance.py:46-49,99-120)."""
    return backward_guidance_generate_batch(sampler, [lay], **kw)[0]


def backward_guidance_generate_batch(sampler: LMDSampler, lays: List[CachedLayout], *, num_inference_steps=50,
                                     guidance_scale=7.5, loss_scale=30, loss_threshold=0.2, max_iter=5,
                                     max_index_step=10, height=512, width=512, decode=True,
                                     guidance_attn_keys=None, first_step=0, n_steps=None, start=None, trace=None,
                                     **energy_kw):
    """Layout-guidance baseline (generation/backward_guidance.py:46-49,99-120; BASELINE config 3 runs it on
    SD2.1-768): one generate_semantic_guidance call per layout on seeded noise, no per-box stage.  The layouts of
    a batch share UNet calls; each keeps its own guidance loop exit.

    The energy is the RATIO-based branch of add_ca_loss_per_attn_map_to_loss (utils/guidance.py:118-130): the plugin's
    guidance kwargs (backward_guidance.py:99-112) do not carry `use_ratio_based_loss`, so the signature default (True,
    guidance.py:91) applies; `ref_ca_saved_attns=None` means no reference-attention term.  `energy_kw` may override
    (`use_ratio_based_loss=False, fg_top_p=...`) for experiments; nothing in the reference's plugin does.
    `first_step` / `n_steps` / `start` (one latent tensor per layout, the state BEFORE step first_step) / `trace` run a
    slice of the schedule from given latents: the teacher-forced parity tests."""
    L = height // 8
    C = sampler.eng.cfg.in_channels
    keys = [tuple(k) for k in (guidance_attn_keys or DEFAULT_GUIDANCE_ATTN_KEYS)]
    jobs = []
    for lay in lays:
        lat = seeded_noise(lay.bg_seed, C, L, L) if start is None else torch.as_tensor(start[len(jobs)]).float()
        overall_bboxes = [[list(lay.boxes[i]) for i in grp] for grp in lay.overall_groups]
        guid = None
        if overall_bboxes:
            guid = dict(bboxes=overall_bboxes, object_positions=lay.overall_object_positions, loss_scale=loss_scale,
                        loss_threshold=loss_threshold, max_iter=max_iter, max_index_step=max_index_step,
                        guidance_attn_keys=keys, **energy_kw)
        jobs.append(Job(lat, torch.cat([lay.overall_uncond, lay.overall_cond]), guidance=guid))
    res = sampler.denoise_batch(jobs, num_inference_steps, guidance_scale=guidance_scale, save_all_latents=False,
                                first_step=first_step, n_steps=n_steps, trace=trace)
    images = sampler.decode(torch.cat([r["latents"] for r in res])) if decode else [None] * len(lays)
    return [dict(image=images[i], latents=r["latents"], guidance_iters=r["guidance_iters"],
                 guidance_iters_fuser_on=0) for i, r in enumerate(res)]


def boxdiff_generate(sampler: LMDSampler, lay: CachedLayout, **kw):
    """BoxDiff baseline for one layout (generation/boxdiff.py:46-131)."""
    return boxdiff_generate_batch(sampler, [lay], **kw)[0]


def boxdiff_generate_batch(sampler: LMDSampler, lays: List[CachedLayout], *, num_inference_steps=50, guidance_scale=7.5,
                           max_index_step=25, height=512, width=512, decode=True, guidance_attn_keys=None, first_step=0,
                           n_steps=None, start=None, trace=None, **boxdiff_kw):
    """The `boxdiff` stage-2 baseline (generation/boxdiff.py:46-131): ONE generate_semantic_guidance call
    per layout on seeded noise with `use_boxdiff=True` (models/pipelines.py:187-188) — before each of the first
    `max_index_step` denoising steps one gradient step on the BoxDiff energy (utils/boxdiff.py:199-259: the five 16x16
    cross-attention maps averaged over layers and heads, x100 token soft-max, smoothed, inner- / outer-box top-k and
    corner terms; csrc/boxdiff.hip), no per-box stage, no reference-attention term (generation/boxdiff.py:104,108 pass
    None / weight 0).  `boxdiff_kw` may carry latent_backward_guidance_boxdiff's own arguments (amp_loss_scale,
    latent_scale, scale_range, P, L, smooth_attentions, sigma); the plugin passes none.
    `first_step` / `n_steps` / `start` / `trace`: a slice of the schedule from given latents (teacher-forced tests)."""
    from .sampler import BOXDIFF_GUIDANCE_ATTN_KEYS
    L = height // 8
    C = sampler.eng.cfg.in_channels
    keys = [tuple(k) for k in (guidance_attn_keys or BOXDIFF_GUIDANCE_ATTN_KEYS)]
    jobs = []
    for lay in lays:
        lat = seeded_noise(lay.bg_seed, C, L, L) if start is None else torch.as_tensor(start[len(jobs)]).float()
        overall_bboxes = [[list(lay.boxes[i]) for i in grp] for grp in lay.overall_groups]
        guid = None
        if overall_bboxes:
            guid = dict(bboxes=overall_bboxes, object_positions=lay.overall_object_positions, use_boxdiff=True,
                        max_index_step=max_index_step, guidance_attn_keys=keys, **boxdiff_kw)
        jobs.append(Job(lat, torch.cat([lay.overall_uncond, lay.overall_cond]), guidance=guid))
    res = sampler.denoise_batch(jobs, num_inference_steps, guidance_scale=guidance_scale, save_all_latents=False,
                                first_step=first_step, n_steps=n_steps, trace=trace)
    images = sampler.decode(torch.cat([r["latents"] for r in res])) if decode else [None] * len(lays)
    return [dict(image=images[i], latents=r["latents"], guidance_iters=r["guidance_iters"], guidance_iters_fuser_on=0)
            for i, r in enumerate(res)]
