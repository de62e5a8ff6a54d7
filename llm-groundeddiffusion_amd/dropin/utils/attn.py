"""`utils.attn` surface of the plugin boundary (reference: utils/attn.py:9-70), in this repo's own form.

Boundary glue with prescribed semantics, not kernels: the two functions exist because callers of the
reference's `utils.attn` expect these names.  On the HIP path the per-step maps already live on the device
as [T, B', heads, HW, T'] tensors; these adapters accept the reference's list-of-dicts form."""
import math

import torch

from lgd_amd.hostprep import shift_tensor


def get_token_attnv2(token_id, saved_attns, attn_key, attn_aggregation_step_start=10,
                     input_ca_has_condition_only=False, return_np=False):
    """Mean map of one token over heads and over the steps from `attn_aggregation_step_start` on (the SAM point
    prompt of training-free LMD).  Like the reference it fails when no step is left to aggregate
    (num_inference_steps <= start)."""
    steps = saved_attns[attn_aggregation_step_start:]
    if not steps:
        raise RuntimeError(f"no saved attention at or after step {attn_aggregation_step_start}")
    rows = 1 if input_ca_has_condition_only else 2
    total = None
    for step in steps:
        m = step[attn_key]
        if m.shape[0] != rows:
            raise AssertionError(f"expected {rows} batch item(s) in the saved map, found {m.shape[0]}")
        col = m[rows - 1, :, :, token_id].float().cpu()          # conditional item: [heads, HW]
        total = col if total is None else total + col
    mean = total.mean(dim=0) / len(steps)
    side = math.isqrt(mean.numel())
    mean = mean.reshape(side, side)
    return mean.numpy() if return_np else mean


def shift_saved_attns(saved_attns, offset, guidance_attn_keys, horizontal_shift_only=False):
    """Moves every saved map of every step by `offset` (normalised x, y; quantised on the 8x8 grid, zero fill)."""
    dx, dy = offset
    if horizontal_shift_only:
        dy = 0.

    def moved(m):
        side = math.isqrt(m.shape[-2])
        return shift_tensor(m.unflatten(2, (side, side)), dx, dy, offset_normalized=True,
                            ignore_last_dim=True).flatten(2, 3)
    return [{k: moved(step[k]) for k in guidance_attn_keys} for step in saved_attns]


def shift_saved_attns_item(saved_attns_item, offset, guidance_attn_keys, horizontal_shift_only=False):
    return shift_saved_attns([saved_attns_item], offset, guidance_attn_keys, horizontal_shift_only)[0]
