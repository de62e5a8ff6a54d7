"""ORACLE TEST INFRASTRUCTURE — CPU fp32 restatement of the reference's BoxDiff baseline:
utils/boxdiff.py (energy + its one-step latent update) and the `use_boxdiff=True` branch of
models/pipelines.py:129-247 that generation/boxdiff.py:114-126 drives.

Same rules as restate.py: plain functional torch, needs neither /root/reference nor diffusers, every function cites the
reference file:line it follows, pinned against the reference's OWN functions run through oracle/ref_harness.py
(oracle/make_golden_boxdiff.py -> tests/golden/boxdiff_energy.npz, run_boxdiff_tiny.npz; tests/test_oracle.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math
from collections.abc import Iterable

import torch
import torch.nn.functional as F

import restate as R

BOXDIFF_GUIDANCE_ATTN_KEYS = [("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]  # generation/boxdiff.py:33-39


def gaussian_kernel(kernel_size=3, sigma=0.5):
    """utils/attn.py:92-110 (GaussianSmoothing.__init__, dim = 2): the product over both axes of
    1/(std sqrt(2 pi)) exp(-((x - mean) / (2 std))^2) — note the (2 std) INSIDE the square, as written there —
    normalised to sum 1."""
    ax = torch.arange(kernel_size, dtype=torch.float32)
    grids = torch.meshgrid([ax, ax], indexing="ij")
    kernel = 1
    for mg in grids:
        mean = (kernel_size - 1) / 2
        kernel = kernel * (1 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-((mg - mean) / (2 * sigma)) ** 2))
    return kernel / kernel.sum()


def max_attention_per_index(attention_maps, object_positions, bboxes, smooth_attentions=True, sigma=0.5, kernel_size=3,
                            P=0.2, L=1):
    """utils/boxdiff.py:20-101 `_compute_max_attention_per_index`: attention_maps (H, W, 77), layer- and head-averaged.
    The first and last text tokens are dropped, the rest is multiplied by 100 and soft-maxed over the tokens (:34-36);
    per phrase token: inner-box / outer-box top-k means (:80-87) and the corner terms (:89-99)."""
    att = F.softmax(attention_maps[:, :, 1:-1] * 100, dim=-1)
    fg, bg, dist_x, dist_y = [], [], [], []
    weight = gaussian_kernel(kernel_size, sigma).view(1, 1, kernel_size, kernel_size)
    for obj_idx, positions in enumerate(object_positions):
        for pos in positions:
            image = att[:, :, pos - 1]                                   # indices shift: the first token was removed (:46)
            H, W = image.shape
            obj_mask = torch.zeros_like(image)
            cx, cy = torch.zeros(W), torch.zeros(H)
            obj_boxes = bboxes[obj_idx]
            if not isinstance(obj_boxes[0], Iterable):
                obj_boxes = [obj_boxes]
            for box in obj_boxes:
                x_min, y_min, x_max, y_max = R.scale_proportion(box, H, W)
                obj_mask[y_min:y_max, x_min:x_max] = 1
                cx[max(x_min - L, 0):min(x_min + L + 1, W)] = 1.        # :64-67
                cx[max(x_max - L, 0):min(x_max + L + 1, W)] = 1.
                cy[max(y_min - L, 0):min(y_min + L + 1, H)] = 1.
                cy[max(y_max - L, 0):min(y_max + L + 1, H)] = 1.
            bg_mask = 1 - obj_mask
            if smooth_attentions:                                        # :71-76 reflect pad 1 + depthwise 3x3
                image = F.conv2d(F.pad(image[None, None], (1, 1, 1, 1), mode="reflect"), weight)[0, 0]
            k = (obj_mask.sum() * P).long()
            fg.append((image * obj_mask).reshape(-1).topk(k)[0].mean())
            k = (bg_mask.sum() * P).long()
            bg.append((image * bg_mask).reshape(-1).topk(k)[0].mean())
            gt_x, gt_y = obj_mask.max(dim=0).values, obj_mask.max(dim=1).values
            dist_x.append(((image.max(dim=0)[0] - gt_x).abs() * cx).mean())
            dist_y.append(((image.max(dim=1)[0] - gt_y).abs() * cy).mean())
    return fg, bg, dist_x, dist_y


def compute_ca_loss_boxdiff(saved_attn, bboxes, object_positions, guidance_attn_keys, **kw):
    """utils/boxdiff.py:121-169 without the (never enabled: generation/boxdiff.py:104,108 pass None / weight 0)
    reference-attention term, + `_compute_loss` :104-118 and `add_ca_loss_per_attn_map_to_loss_boxdiff` :172-196:
    the maps of all keys are concatenated over heads and averaged (:152) -> (HW, 77)."""
    if len(bboxes) == 0:
        return torch.tensor(0.)
    attn_map = torch.cat([saved_attn[k].squeeze(dim=0) for k in guidance_attn_keys], dim=0).mean(dim=0)
    i, j = attn_map.shape
    side = int(math.sqrt(i))
    kw = {k: v for k, v in kw.items() if k in ("P", "L", "smooth_attentions", "sigma", "kernel_size")}
    fg, bg, dx, dy = max_attention_per_index(attn_map.view(side, side, j), object_positions, bboxes, **kw)
    # `max(0, 1. - curr_max)` is PYTHON's max of (int, tensor) (:107-109): the tensor only if `tensor > 0` holds, else the
    # int 0 — in particular a NaN (top-k of k = 0 elements: a box whose mask holds fewer than 1 / P pixels) drops out
    pymax0 = lambda v: v if bool(v > 0) else torch.tensor(0.)
    losses_fg = [pymax0(1. - v) for v in fg]
    losses_bg = [pymax0(v) for v in bg]
    return sum(losses_fg) + sum(losses_bg) + sum(dx) + sum(dy)


def boxdiff_step_scale(index, n_timesteps, latent_scale=20, scale_range=(1., 0.5)):
    """utils/boxdiff.py:233-238 (the branch that always runs): latent_scale * sqrt(linear ramp from 1 to 0.5)."""
    return latent_scale * (scale_range[0] + (scale_range[1] - scale_range[0]) * index / (n_timesteps - 1)) ** 0.5


def latent_backward_guidance_boxdiff(sd, cfg, sched, cond_emb, index, bboxes, object_positions, t, latents, loss,
                                     amp_loss_scale=10, latent_scale=20, scale_range=(1., 0.5), max_index_step=25,
                                     guidance_attn_keys=None, early_exit=True, trace=None, **kw):
    """utils/boxdiff.py:199-259: ONE gradient step per denoising step while index < max_index_step (no loss threshold,
    no inner loop); the loss is scaled by amp_loss_scale and the step de-scaled by it."""
    if index < max_index_step:
        saved = {}
        latents = latents.detach().requires_grad_(True)
        order = [("down", i, j, 0) for i in range(3) for j in range(2)] + [("mid", 0, 0, 0)] + \
                [("up", i, j, 0) for i in range(1, 4) for j in range(3)]
        stop = max(guidance_attn_keys, key=lambda k: order.index(tuple(k))) if early_exit else None
        R.unet_forward(sd, cfg, latents, t, cond_emb, saved=saved, save_keys=guidance_attn_keys, stop_after=stop)
        loss = compute_ca_loss_boxdiff(saved, bboxes, object_positions, guidance_attn_keys, **kw) * amp_loss_scale
        grad = torch.autograd.grad(loss.requires_grad_(True), [latents])[0]
        latents = latents.detach()
        scale = boxdiff_step_scale(index, len(sched.timesteps), latent_scale, scale_range)
        latents = latents - scale / amp_loss_scale * grad
        loss = loss.detach()
        if trace is not None:
            trace.append(dict(index=index, loss=float(loss), grad=grad.clone()))
    return latents, loss


def generate_boxdiff(sd, cfg, sched, latents, input_embeddings, steps, bboxes, object_positions, guidance_scale=7.5,
                     max_index_step=25, guidance_attn_keys=None, trace=None, per_step=None, starts=None, **kw):
    """pipelines.py:129-247 with use_boxdiff=True (:187-188), as generation/boxdiff.py:114-126 calls it."""
    text_emb, _, cond_emb = input_embeddings
    keys = [tuple(k) for k in (guidance_attn_keys or BOXDIFF_GUIDANCE_ATTN_KEYS)]
    latents = latents.clone()
    sched.set_timesteps(steps)
    loss = torch.tensor(10000.)
    for index, t in enumerate(sched.timesteps):
        if starts is not None:
            starts.append(latents.clone())
        if bboxes:
            latents, loss = latent_backward_guidance_boxdiff(sd, cfg, sched, cond_emb, index, bboxes, object_positions, t,
                                                             latents, loss, max_index_step=max_index_step,
                                                             guidance_attn_keys=keys, trace=trace, **kw)
        latents = R._cfg_step(sd, cfg, sched, latents, t, text_emb, guidance_scale)
        if per_step is not None:
            per_step.append(latents.clone())
    return latents
