"""utils/utils.py of the reference — adapters over lgd_amd.hostprep (same names and argument meaning;
tensors go to `torch_device` where the reference put them there)."""
import gc

import torch

from lgd_amd import hostprep as _hp
from lgd_amd.hostprep import (binary_mask_to_box, binary_mask_to_center, expand_overall_bboxes,  # noqa: F401
                              get_centered_box, iou, scale_proportion, shift_tensor)

torch_device = "cuda"


def proportion_to_mask(obj_box, H, W, use_legacy=False, return_np=False):
    m = _hp.proportion_to_mask(obj_box, H, W, use_legacy, return_np)
    return m if return_np else m.to(torch_device)


def binary_mask_to_box_mask(mask, to_device=True):
    m = _hp.binary_mask_to_box_mask(mask.cpu() if isinstance(mask, torch.Tensor) else mask)
    return m.to(torch_device) if to_device else m


def free_memory():
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
