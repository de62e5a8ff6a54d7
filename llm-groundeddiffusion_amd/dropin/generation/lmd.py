"""generation/lmd.py of the reference: plugin `lmd` (training-free LMD), on the HIP engine."""
import models
from lgd_amd.pipeline import DEFAULT_MAX_ITER, lmd_generate

from ._common import (DEFAULT_OVERALL_NEGATIVE_PROMPT, DEFAULT_SO_NEGATIVE_PROMPT, EasyDict, build_layout, note_precision,
                      sam_refiner)

version = "lmd"
height = width = 512
guidance_scale = 7.5


def run(spec, bg_seed=1, overall_prompt_override="", fg_seed_start=20, frozen_step_ratio=0.5, num_inference_steps=50,
        loss_scale=5, loss_threshold=5.0, max_iter=DEFAULT_MAX_ITER, max_index_step=30, overall_loss_scale=5,
        overall_loss_threshold=5.0, overall_max_iter=DEFAULT_MAX_ITER, overall_max_index_step=30, fg_top_p=0.2,
        bg_top_p=0.2, overall_fg_top_p=0.2, overall_bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0, overall_fg_weight=1.0,
        overall_bg_weight=4.0, ref_ca_loss_weight=2.0, so_center_box=True, fg_blending_ratio=0.01,
        so_negative_prompt=DEFAULT_SO_NEGATIVE_PROMPT, overall_negative_prompt=DEFAULT_OVERALL_NEGATIVE_PROMPT,
        mask_th_for_point=0.25, so_horizontal_center_only=False, align_with_overall_bboxes=True,
        horizontal_shift_only=False, use_fast_schedule=False, so_vertical_placement="floor_padding",
        so_floor_padding=0.2, use_box_input=False, use_ref_ca=True, use_autocast=False, verbose=False):
    """Argument names and defaults of generation/lmd.py:215-256 (incl. so_center_box=True /
    align_with_overall_bboxes=True: per-box generations run on a centred box, histories, masks and
    reference maps are shifted back onto the overall boxes before composition).  When `models.model_dict` carries
    a SAM model (generate.py:126-127 `model_dict.update(sam.load_sam())`) every per-box mask is SAM's refinement of the
    object token's attention map (generation/lmd.py:124-149) with `mask_th_for_point` / `use_box_input`; otherwise the
    masks are the box masks.  `use_autocast=False` (the reference's default here: fp32) is answered with
    fp16 compute / fp32 accumulation and ONE RuntimeWarning (`_common.note_precision`)."""
    if not use_autocast:                       # the reference's default for this method: fp32 (generation/lmd.py:254,375)
        note_precision("generation.lmd.run", "use_autocast=False")
    if num_inference_steps <= 10:
        # the reference crashes here too (attn_aggregation_step_start=10, generation/lmd.py:36,124-131)
        print("note: the reference's SAM point prompt aggregates maps from step 10 on; with <=10 steps it would fail")
    sm = models.model_dict.sampler
    refiner = sam_refiner(models.model_dict, height, width, discourage_mask_below_coarse_iou=0.25,
                          use_box_input=use_box_input, mask_th_for_point=mask_th_for_point, verbose=verbose)
    lay = build_layout(spec, bg_seed, fg_seed_start, so_negative_prompt, overall_negative_prompt, height, width,
                       overall_prompt_override, verbose)
    out = lmd_generate(sm, lay, num_inference_steps=num_inference_steps, frozen_step_ratio=frozen_step_ratio,
                       guidance_scale=guidance_scale, loss_scale=loss_scale, loss_threshold=loss_threshold,
                       max_iter=max_iter, max_index_step=max_index_step, overall_loss_scale=overall_loss_scale,
                       overall_loss_threshold=overall_loss_threshold, overall_max_iter=overall_max_iter,
                       overall_max_index_step=overall_max_index_step, fg_top_p=fg_top_p, bg_top_p=bg_top_p,
                       overall_fg_top_p=overall_fg_top_p, overall_bg_top_p=overall_bg_top_p, fg_weight=fg_weight,
                       bg_weight=bg_weight, overall_fg_weight=overall_fg_weight, overall_bg_weight=overall_bg_weight,
                       ref_ca_loss_weight=ref_ca_loss_weight, fg_blending_ratio=fg_blending_ratio, use_ref_ca=use_ref_ca,
                       height=height, width=width, use_fast_schedule=use_fast_schedule,
                       so_center_box=so_center_box, so_horizontal_center_only=so_horizontal_center_only,
                       so_vertical_placement=so_vertical_placement, so_floor_padding=so_floor_padding,
                       align_with_overall_bboxes=align_with_overall_bboxes, horizontal_shift_only=horizontal_shift_only,
                       mask_refiner=refiner)
    return EasyDict(image=out["image"], so_img_list=out["so_images"])
