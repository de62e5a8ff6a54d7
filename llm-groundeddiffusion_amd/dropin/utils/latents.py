"""utils/latents.py of the reference — adapters over lgd_amd.hostprep keeping the reference signatures
(model_dict first, height/width in pixels)."""
import torch

from lgd_amd import hostprep as _hp
from utils import torch_device


def get_unscaled_latents(batch_size, in_channels, height, width, generator, dtype):
    x = torch.randn((batch_size, in_channels, height // 8, width // 8), generator=generator, dtype=dtype)
    return x.to(torch_device, dtype=dtype)


def get_scaled_latents(batch_size, in_channels, height, width, generator, dtype, scheduler):
    return get_unscaled_latents(batch_size, in_channels, height, width, generator, dtype) * scheduler.init_noise_sigma


def blend_latents(latents_bg, latents_fg, fg_mask, fg_blending_ratio=0.01):
    assert not torch.allclose(latents_bg, latents_fg), "latents_bg should be independent with latents_fg"
    r = fg_blending_ratio
    out = latents_bg * (1. - fg_mask) + (latents_bg * (1. - r) ** 0.5 + latents_fg * r ** 0.5) * fg_mask
    return out.to(latents_bg.dtype)


def get_input_latents_list(model_dict, bg_seed, fg_seed_start, fg_blending_ratio, height, width,
                           so_prompt_phrase_box_list=None, so_boxes=None, verbose=False):
    if so_boxes is None:
        so_boxes = [item[-1] for item in so_prompt_phrase_box_list]
    lst, bg = _hp.input_latents_list(bg_seed, fg_seed_start, so_boxes, fg_blending_ratio,
                                     model_dict.unet.config.in_channels, height // 8, width // 8,
                                     model_dict.scheduler.init_noise_sigma)
    dt = model_dict.dtype
    return [x.to(torch_device, dt) for x in lst], bg.to(torch_device, dt)


@torch.no_grad()
def compose_latents(model_dict, latents_all_list, mask_tensor_list, num_inference_steps, overall_batch_size, height,
                    width, latents_bg=None, bg_seed=None, compose_box_to_bg=True, use_fast_schedule=False,
                    fast_after_steps=None):
    if latents_bg is None:
        g = torch.manual_seed(bg_seed)
        latents_bg = get_scaled_latents(overall_batch_size, model_dict.unet.config.in_channels, height, width, g,
                                        model_dict.dtype, model_dict.scheduler)
    assert compose_box_to_bg, "compose_box_to_bg=False is not used by LMD / LMD+"
    steps = fast_after_steps if use_fast_schedule else num_inference_steps
    comp, fg = _hp.compose(latents_all_list, [m.cpu() for m in mask_tensor_list], steps, latents_bg.to(torch_device))
    return comp.to(torch_device), fg.to(torch_device)


def align_with_bboxes(latents_all_list, mask_tensor_list, bboxes, horizontal_shift_only=False):
    return _hp.align_with_bboxes(latents_all_list, mask_tensor_list, bboxes, horizontal_shift_only)


@torch.no_grad()
def compose_latents_with_alignment(model_dict, latents_all_list, mask_tensor_list, num_inference_steps,
                                   overall_batch_size, height, width, align_with_overall_bboxes=True,
                                   overall_bboxes=None, horizontal_shift_only=False, **kwargs):
    if align_with_overall_bboxes and len(latents_all_list):
        flat = _hp.expand_overall_bboxes(overall_bboxes)
        latents_all_list, mask_tensor_list, offset_list = align_with_bboxes(latents_all_list, mask_tensor_list, flat,
                                                                            horizontal_shift_only)
    else:
        offset_list = [(0., 0.) for _ in latents_all_list]
    comp, fg = compose_latents(model_dict, latents_all_list, mask_tensor_list, num_inference_steps,
                               overall_batch_size, height, width, **kwargs)
    return comp, fg, offset_list
