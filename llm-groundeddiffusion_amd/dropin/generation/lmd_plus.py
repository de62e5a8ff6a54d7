"""generation/lmd_plus.py of the reference: plugin `lmd_plus` (version + run), on the HIP engine.
generate.py imports this module after `models.model_dict` is set (generate.py:118-153) and calls
`run(spec, bg_seed=..., fg_seed_start=..., **run_kwargs)`; only `.image` of the result is read (:381)."""
import models
from lgd_amd.pipeline import DEFAULT_MAX_ITER, lmd_plus_generate

from ._common import DEFAULT_OVERALL_NEGATIVE_PROMPT, DEFAULT_SO_NEGATIVE_PROMPT, EasyDict, build_layout, note_precision, sam_refiner

version = "lmd_plus"
height = width = 512
guidance_scale = 7.5


def run(spec, bg_seed=1, overall_prompt_override="", fg_seed_start=20, frozen_step_ratio=0.5, num_inference_steps=50,
        loss_scale=5, loss_threshold=5.0, max_iter=DEFAULT_MAX_ITER, max_index_step=0, overall_loss_scale=5,
        overall_loss_threshold=5.0, overall_max_iter=DEFAULT_MAX_ITER, overall_max_index_step=30,
        so_gligen_scheduled_sampling_beta=0.4, overall_gligen_scheduled_sampling_beta=0.4, overall_fg_top_p=0.2,
        overall_bg_top_p=0.2, overall_fg_weight=1.0, overall_bg_weight=4.0, ref_ca_loss_weight=2.0, so_center_box=False,
        fg_blending_ratio=0.1, so_negative_prompt=DEFAULT_SO_NEGATIVE_PROMPT,
        overall_negative_prompt=DEFAULT_OVERALL_NEGATIVE_PROMPT, so_horizontal_center_only=True,
        align_with_overall_bboxes=False, horizontal_shift_only=True, use_fast_schedule=False, use_ref_ca=True,
        use_autocast=True, verbose=False):
    """Argument names and defaults of generation/lmd_plus.py:193-228.  `use_autocast=True` (the default) is the
    arithmetic the HIP path always runs (fp16 compute, fp32 accumulation); `use_autocast=False` is answered with the same
    arithmetic and ONE RuntimeWarning (`_common.note_precision`)."""
    if not use_autocast:
        note_precision("generation.lmd_plus.run", "use_autocast=False")
    sm = models.model_dict.sampler
    refiner = sam_refiner(models.model_dict, height, width, discourage_mask_below_coarse_iou=0.25, verbose=verbose)
    lay = build_layout(spec, bg_seed, fg_seed_start, so_negative_prompt, overall_negative_prompt, height, width,
                       overall_prompt_override, verbose)
    print("Key generation settings:", spec, bg_seed, fg_seed_start, frozen_step_ratio,
          so_gligen_scheduled_sampling_beta, overall_gligen_scheduled_sampling_beta, overall_max_index_step)
    out = lmd_plus_generate(sm, lay, num_inference_steps=num_inference_steps, frozen_step_ratio=frozen_step_ratio,
                            guidance_scale=guidance_scale, loss_scale=loss_scale, loss_threshold=loss_threshold,
                            max_iter=max_iter, max_index_step=max_index_step,
                            so_gligen_scheduled_sampling_beta=so_gligen_scheduled_sampling_beta,
                            overall_gligen_scheduled_sampling_beta=overall_gligen_scheduled_sampling_beta,
                            overall_loss_scale=overall_loss_scale, overall_loss_threshold=overall_loss_threshold,
                            overall_max_iter=overall_max_iter, overall_max_index_step=overall_max_index_step,
                            overall_fg_top_p=overall_fg_top_p, overall_bg_top_p=overall_bg_top_p,
                            overall_fg_weight=overall_fg_weight, overall_bg_weight=overall_bg_weight,
                            ref_ca_loss_weight=ref_ca_loss_weight, fg_blending_ratio=fg_blending_ratio,
                            use_ref_ca=use_ref_ca, height=height, width=width, use_fast_schedule=use_fast_schedule,
                            so_center_box=so_center_box, so_horizontal_center_only=so_horizontal_center_only,
                            align_with_overall_bboxes=align_with_overall_bboxes,
                            horizontal_shift_only=horizontal_shift_only, mask_refiner=refiner)
    return EasyDict(image=out["image"], so_img_list=out["so_images"])
