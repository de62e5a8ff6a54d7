"""Host-side preparation between the two stages of LMD / LMD+: box <-> mask
rounding rules, seeded initial noise, foreground/background blending, latent composition, box
alignment shifts.  Tiny CPU tensors; the arithmetic follows the reference exactly because seeds and
box rounding must reproduce (utils/utils.py, utils/latents.py of the reference — cited per function).
Everything returns CPU tensors; callers move them where they need them.
"""
import threading

import numpy as np
import torch


# -------------------------------------------------------------------------------------------------
# box / mask geometry.  The RULES are the reference's (utils/utils.py, cited per function) because box
# rounding decides which latent pixels are guided, blended and frozen; the code is this repo's own and is
# pinned against known-answer vectors made by the reference functions (tests/test_hostgeom.py).
# -------------------------------------------------------------------------------------------------
def get_centered_box(box, horizontal_center_only=True, vertical_placement='centered', vertical_center=0.5,
                     floor_padding=None):
    """Rule of utils/utils.py:20-44: same size, horizontally centred; vertically untouched, centred on
    `vertical_center`, or standing `floor_padding` above the bottom edge."""
    x0, y0, x1, y1 = box
    half_w = (x1 - x0) / 2
    out = [0.5 - half_w, y0, 0.5 + half_w, y1]
    if horizontal_center_only:
        return out
    h = y1 - y0
    if vertical_placement == 'centered':
        if floor_padding is not None:
            raise AssertionError("Set vertical_placement to floor_padding to use floor padding")
        out[1], out[3] = vertical_center - h / 2, vertical_center + h / 2
    elif vertical_placement == 'floor_padding':
        bottom = 1 - floor_padding
        out[1], out[3] = bottom - h, bottom
    else:
        raise ValueError(f"Unknown vertical placement: {vertical_placement}")
    return out


def _snap_axis(lo, hi, n):
    """One axis of a [0,1] box on an n-pixel grid: start and EXTENT are rounded independently (a box keeps
    its pixel size wherever it sits; Python round = half-to-even), then clipped to the grid."""
    start = round(lo * n)
    end = start + round((hi - lo) * n)
    return max(start, 0), min(end, n)


def scale_proportion(obj_box, H, W, use_legacy=False):
    """[0,1] xyxy box -> integer pixel rectangle (x_min, y_min, x_max, y_max), rule of utils/utils.py:57-70
    (`use_legacy`: plain truncation of the four corners)."""
    bx0, by0, bx1, by1 = obj_box
    if use_legacy:
        return int(bx0 * W), int(by0 * H), int(bx1 * W), int(by1 * H)
    (x0, x1), (y0, y1) = _snap_axis(bx0, bx1, W), _snap_axis(by0, by1, H)
    return x0, y0, x1, y1


def _filled(H, W, x0, y0, x1, y1, as_numpy=False):
    m = np.zeros((H, W)) if as_numpy else torch.zeros(H, W)
    m[y0:y1, x0:x1] = 1.
    return m


def proportion_to_mask(obj_box, H, W, use_legacy=False, return_np=False):
    """Box mask on the H x W latent grid (utils/utils.py:46-55)."""
    return _filled(H, W, *scale_proportion(obj_box, H, W, use_legacy), as_numpy=return_np)


def _as_bool_array(mask):
    return (mask.detach().cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)).astype(bool)


def _span(flags):
    hit = np.flatnonzero(flags)
    if hit.size == 0:
        raise ValueError('The mask is empty')
    return int(hit[0]), int(hit[-1])


def binary_mask_to_box(mask, enlarge_box_by_one=True, w_scale=1, h_scale=1):
    """Bounding box [xmin, ymin, xmax, ymax] of the set pixels, optionally grown by one pixel per side and
    clipped to the mask size (utils/utils.py:72-88; the max side is clipped to `size`, not `size - 1`)."""
    m = _as_bool_array(mask)
    H, W = m.shape
    (y_lo, y_hi), (x_lo, x_hi) = _span(m.any(axis=1)), _span(m.any(axis=0))
    if enlarge_box_by_one:
        y_lo, y_hi, x_lo, x_hi = max(y_lo - 1, 0), min(y_hi + 1, H), max(x_lo - 1, 0), min(x_hi + 1, W)
    return [x_lo * w_scale, y_lo * h_scale, x_hi * w_scale, y_hi * h_scale]


def binary_mask_to_box_mask(mask):
    """Mask of the (grown, inclusive) bounding box of a mask (utils/utils.py:90-100)."""
    x0, y0, x1, y1 = binary_mask_to_box(mask)
    return _filled(mask.shape[0], mask.shape[1], x0, y0, x1 + 1, y1 + 1)


def binary_mask_to_center(mask, normalize=False):
    """Mass centre (x, y) of a mask in pixels, or as fractions of the width / height (utils/utils.py:102-123).
    The value feeds `round()` in the alignment shift, so the reference's number format is kept: for torch
    masks (bool / integer) the quotient of the two integer sums is a float32 division, for numpy masks a
    float64 one."""
    is_torch = isinstance(mask, torch.Tensor)
    if is_torch and mask.is_floating_point():
        m = mask.detach().cpu()
        h, w = m.shape
        total = m.sum()
        x = float((m.sum(dim=0) @ torch.arange(w, dtype=m.dtype)) / total)
        y = float((m.sum(dim=1) @ torch.arange(h, dtype=m.dtype)) / total)
    else:
        m = mask.detach().cpu().numpy() if is_torch else np.asarray(mask)
        h, w = m.shape
        cols, rows, total = m.sum(axis=0), m.sum(axis=1), m.sum()
        num_x, num_y = cols @ np.arange(w), rows @ np.arange(h)
        if is_torch:
            x = float(np.float32(num_x) / np.float32(total))
            y = float(np.float32(num_y) / np.float32(total))
        else:
            x, y = num_x / total, num_y / total
    if normalize:
        x, y = x / w, y / h
    return x, y


def iou(mask, masks, eps=1e-6):
    """IoU of one [h, w] mask against a stack [n, h, w] (utils/utils.py:125-131)."""
    a = np.asarray(mask).astype(bool)[None]
    b = np.asarray(masks).astype(bool)
    inter = np.logical_and(a, b).reshape(b.shape[0], -1).sum(axis=1)
    union = np.logical_or(a, b).reshape(b.shape[0], -1).sum(axis=1)
    return inter / (union + eps)


def expand_overall_bboxes(overall_bboxes):
    """[[box 1 for phrase 1, box 2 for phrase 1], ...] -> [box 1, box 2, ...] (utils/utils.py:136-143)."""
    return [box for group in overall_bboxes for box in group]


def shift_tensor(tensor, x_offset, y_offset, base_w=8, base_h=8, offset_normalized=False, ignore_last_dim=False):
    """Translate the 2-D image held in the last two dims (or, with `ignore_last_dim`, in dims -3/-2 of an
    attention map [..., h, w, tokens]) by an integer pixel offset with zero fill (utils/utils.py:145-180).
    Normalised offsets are quantised on the base_w x base_h grid first, so that 64x64 latents and the 8x8 /
    16x16 attention maps of one object move by the same fraction of the image."""
    ydim, xdim = (-3, -2) if ignore_last_dim else (-2, -1)
    H, W = tensor.shape[ydim], tensor.shape[xdim]
    if offset_normalized:
        if H % base_h or W % base_w:
            raise AssertionError(f"{H, W} is not a multiple of {base_h, base_w}")
        x_offset, y_offset = round(x_offset * base_w) * (W // base_w), round(y_offset * base_h) * (H // base_h)
    out = torch.zeros_like(tensor)
    src, dst = tensor, out
    for dim, off, size in ((ydim, int(y_offset), H), (xdim, int(x_offset), W)):
        keep = size - abs(off)
        if keep <= 0:
            return out                                                        # shifted out of the frame
        src, dst = src.narrow(dim, max(-off, 0), keep), dst.narrow(dim, max(off, 0), keep)
    dst.copy_(src)
    return out


# -------------------------------------------------------------------------------------------------
# latents (utils/latents.py of the reference)
# -------------------------------------------------------------------------------------------------
# torch.manual_seed() re-seeds the PROCESS-WIDE default generator (the reference's own idiom, latents.py:7-18): the
# seed + draw pair must not interleave with another host thread's (lanes.LanePool)
RNG_LOCK = threading.Lock()


def seeded_noise(seed, in_channels, h, w, dtype=torch.float32):
    """latents.py:7-18: CPU generator; fp32 first (directly sampling fp16 gives different noise)."""
    with RNG_LOCK:
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
