"""Host-side preparation between the two stages of LMD / LMD+: box <-> mask
rounding rules, seeded initial noise, foreground/background blending, latent composition, box
alignment shifts.  Tiny CPU tensors; the arithmetic follows the reference exactly because seeds and
box rounding must reproduce (utils/utils.py, utils/latents.py of the reference — cited per function).
Everything returns CPU tensors; callers move them where they need them.
"""
import numpy as np
import torch


def get_centered_box(box, horizontal_center_only=True, vertical_placement='centered', vertical_center=0.5,
                     floor_padding=None):
    """utils/utils.py:20-44."""
    x_min, y_min, x_max, y_max = box
    w = x_max - x_min
    x_min_new, x_max_new = 0.5 - w / 2, 0.5 + w / 2
    if horizontal_center_only:
        return [x_min_new, y_min, x_max_new, y_max]
    h = y_max - y_min
    if vertical_placement == 'centered':
        assert floor_padding is None, "Set vertical_placement to floor_padding to use floor padding"
        y_min_new, y_max_new = vertical_center - h / 2, vertical_center + h / 2
    elif vertical_placement == 'floor_padding':
        y_max_new = 1 - floor_padding
        y_min_new = y_max_new - h
    else:
        raise ValueError(f"Unknown vertical placement: {vertical_placement}")
    return [x_min_new, y_min_new, x_max_new, y_max_new]


def scale_proportion(obj_box, H, W, use_legacy=False):
    """utils/utils.py:57-70."""
    if use_legacy:
        x_min, y_min, x_max, y_max = int(obj_box[0] * W), int(obj_box[1] * H), int(obj_box[2] * W), int(obj_box[3] * H)
    else:
        x_min, y_min = round(obj_box[0] * W), round(obj_box[1] * H)
        box_w, box_h = round((obj_box[2] - obj_box[0]) * W), round((obj_box[3] - obj_box[1]) * H)
        x_max, y_max = x_min + box_w, y_min + box_h
        x_min, y_min = max(x_min, 0), max(y_min, 0)
        x_max, y_max = min(x_max, W), min(y_max, H)
    return x_min, y_min, x_max, y_max


def proportion_to_mask(obj_box, H, W, use_legacy=False, return_np=False):
    """utils/utils.py:46-55."""
    x_min, y_min, x_max, y_max = scale_proportion(obj_box, H, W, use_legacy)
    mask = np.zeros((H, W)) if return_np else torch.zeros(H, W)
    mask[y_min: y_max, x_min: x_max] = 1.
    return mask


def binary_mask_to_box(mask, enlarge_box_by_one=True, w_scale=1, h_scale=1):
    """utils/utils.py:72-88."""
    mask_loc = torch.where(mask) if isinstance(mask, torch.Tensor) else np.where(mask)
    height, width = mask.shape
    if len(mask_loc) == 0:
        raise ValueError('The mask is empty')
    if enlarge_box_by_one:
        ymin, ymax = max(min(mask_loc[0]) - 1, 0), min(max(mask_loc[0]) + 1, height)
        xmin, xmax = max(min(mask_loc[1]) - 1, 0), min(max(mask_loc[1]) + 1, width)
    else:
        ymin, ymax = min(mask_loc[0]), max(mask_loc[0])
        xmin, xmax = min(mask_loc[1]), max(mask_loc[1])
    return [xmin * w_scale, ymin * h_scale, xmax * w_scale, ymax * h_scale]


def binary_mask_to_box_mask(mask):
    """utils/utils.py:90-100."""
    x_min, y_min, x_max, y_max = binary_mask_to_box(mask)
    H, W = mask.shape
    mask = torch.zeros(H, W)
    mask[y_min: y_max + 1, x_min: x_max + 1] = 1.
    return mask


def binary_mask_to_center(mask, normalize=False):
    """utils/utils.py:102-123: mass centre of a mask."""
    h, w = mask.shape
    total = mask.sum()
    if isinstance(mask, torch.Tensor):
        cs, rs = mask.sum(dim=0), mask.sum(dim=1)          # bool masks (the reference's case) sum to int64
        x_coord = ((cs @ torch.arange(w, dtype=cs.dtype)) / total).item()
        y_coord = ((rs @ torch.arange(h, dtype=rs.dtype)) / total).item()
    else:
        x_coord = (mask.sum(axis=0) @ np.arange(w)) / total
        y_coord = (mask.sum(axis=1) @ np.arange(h)) / total
    if normalize:
        x_coord, y_coord = x_coord / w, y_coord / h
    return x_coord, y_coord


def iou(mask, masks, eps=1e-6):
    mask = mask[None].astype(bool)
    masks = masks.astype(bool)
    i = (mask & masks).sum(axis=(1, 2))
    u = (mask | masks).sum(axis=(1, 2))
    return i / (u + eps)


def expand_overall_bboxes(overall_bboxes):
    """[[box 1 for phrase 1, box 2 for phrase 1], ...] -> [box 1, box 2, ...] (utils/utils.py:136-143)."""
    return sum(overall_bboxes, start=[])


def shift_tensor(tensor, x_offset, y_offset, base_w=8, base_h=8, offset_normalized=False, ignore_last_dim=False):
    """utils/utils.py:145-180: integer shift with zero fill; normalised offsets are quantised on the
    8x8 base grid so latents and all cross-attention levels move consistently."""
    if ignore_last_dim:
        tensor_h, tensor_w = tensor.shape[-3:-1]
    else:
        tensor_h, tensor_w = tensor.shape[-2:]
    if offset_normalized:
        assert tensor_h % base_h == 0 and tensor_w % base_w == 0, f"{tensor_h, tensor_w} is not a multiple of {base_h, base_w}"
        sh, sw = tensor_h // base_h, tensor_w // base_w
        x_offset, y_offset = round(x_offset * base_w) * sw, round(y_offset * base_h) * sh
    new_tensor = torch.zeros_like(tensor)
    overlap_w, overlap_h = tensor_w - abs(x_offset), tensor_h - abs(y_offset)
    y_src, y_dst = (0, y_offset) if y_offset >= 0 else (-y_offset, 0)
    x_src, x_dst = (0, x_offset) if x_offset >= 0 else (-x_offset, 0)
    if ignore_last_dim:
        new_tensor[..., y_dst:y_dst + overlap_h, x_dst:x_dst + overlap_w, :] = \
            tensor[..., y_src:y_src + overlap_h, x_src:x_src + overlap_w, :]
    else:
        new_tensor[..., y_dst:y_dst + overlap_h, x_dst:x_dst + overlap_w] = \
            tensor[..., y_src:y_src + overlap_h, x_src:x_src + overlap_w]
    return new_tensor


# -------------------------------------------------------------------------------------------------
# latents (utils/latents.py of the reference)
# -------------------------------------------------------------------------------------------------
def seeded_noise(seed, in_channels, h, w, dtype=torch.float32):
    """latents.py:7-18: CPU generator; fp32 first (directly sampling fp16 gives different noise)."""
    return torch.randn((1, in_channels, h, w), generator=torch.manual_seed(seed), dtype=dtype)


def input_latents_list(bg_seed, fg_seed_start, so_boxes, fg_blending_ratio, in_channels=4, H=64, W=64,
                       init_noise_sigma=1.0):
    """latents.py:120-161 + blend_latents :25-36: per-box start noise = background noise with an
    independent foreground mixed in inside the box (variance preserving)."""
    bg = seeded_noise(bg_seed, in_channels, H, W)
    out = []
    r = fg_blending_ratio
    for idx, box in enumerate(so_boxes):
        m = proportion_to_mask(box, H, W)
        fg_seed = fg_seed_start + idx
        if fg_seed == bg_seed:
            fg_seed += 12345
        fg = seeded_noise(fg_seed, in_channels, H, W)
        out.append((bg * (1. - m) + (bg * np.sqrt(1. - r) + fg * np.sqrt(r)) * m) * init_noise_sigma)
    return out, bg * init_noise_sigma


def compose(latents_all_list, mask_list, steps, latents_bg):
    """latents.py:38-83 (compose_box_to_bg=True): histories are pasted largest mask first; step 0 also
    takes the whole bounding box of each mask.  Works on the device the histories live on."""
    dev = latents_bg.device
    composed = torch.zeros((steps + 1, *latents_bg.shape), device=dev, dtype=latents_bg.dtype)
    composed[0] = latents_bg
    fg_idx = torch.zeros(latents_bg.shape[-2:], dtype=torch.long)
    order = np.argsort(-np.array([float(m.sum()) for m in mask_list])) if len(mask_list) else []
    for i in order:
        bm = binary_mask_to_box_mask(mask_list[i]).to(dev)[None, None, None]
        composed[0] = composed[0] * (1. - bm) + latents_all_list[i][0].to(dev) * bm
    for i in order:
        m = mask_list[i].bool().cpu()
        fg_idx = fg_idx * (~m) + (i + 1) * m
        me = m.to(dev)[None, None, None].to(latents_bg.dtype)
        composed = composed * (1. - me) + latents_all_list[i][:steps + 1].to(dev) * me
    return composed, fg_idx


def align_with_bboxes(latents_all_list, mask_list, bboxes, horizontal_shift_only=False):
    """latents.py:85-106."""
    new_l, new_m, offsets = [], [], []
    for lat, m, bbox in zip(latents_all_list, mask_list, bboxes):
        xs, ys = binary_mask_to_center(m.cpu(), normalize=True)
        x_off = (bbox[0] + bbox[2]) / 2 - xs
        y_off = 0. if horizontal_shift_only else (bbox[1] + bbox[3]) / 2 - ys
        new_l.append(shift_tensor(lat, x_off, y_off, offset_normalized=True))
        new_m.append(shift_tensor(m, x_off, y_off, offset_normalized=True))
        offsets.append((x_off, y_off))
    return new_l, new_m, offsets
