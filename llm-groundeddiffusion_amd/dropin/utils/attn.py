"""utils/attn.py of the reference: host-side handling of saved cross-attention maps."""
import math

import torch

import utils


def get_token_attnv2(token_id, saved_attns, attn_key, attn_aggregation_step_start=10,
                     input_ca_has_condition_only=False, return_np=False):
    """attn.py:9-38: mean over steps >= start and over heads of one token's map (SAM point prompt).
    Raises on an empty stack exactly like the reference (num_inference_steps <= start)."""
    saved_attns = saved_attns[attn_aggregation_step_start:]
    saved_attns = [saved_attn[attn_key].cpu() for saved_attn in saved_attns]
    attn = torch.stack(saved_attns, dim=0).mean(dim=0)
    if not input_ca_has_condition_only:
        assert attn.shape[0] == 2, f"Expect to have 2 items (uncond and cond), but found {attn.shape[0]} items"
        attn = attn[1]
    else:
        assert attn.shape[0] == 1, f"Expect to have 1 item (cond only), but found {attn.shape[0]} items"
        attn = attn[0]
    attn = attn.mean(dim=0)[:, token_id]
    H = W = int(math.sqrt(attn.shape[0]))
    attn = attn.reshape((H, W))
    return attn.numpy() if return_np else attn


def shift_saved_attns_item(saved_attns_item, offset, guidance_attn_keys, horizontal_shift_only=False):
    """attn.py:40-64."""
    x_offset, y_offset = offset
    if horizontal_shift_only:
        y_offset = 0.
    out = {}
    for k in guidance_attn_keys:
        attn_map = saved_attns_item[k]
        side = int(math.sqrt(attn_map.shape[-2]))
        attn_map = attn_map.unflatten(2, (side, side))
        attn_map = utils.shift_tensor(attn_map, x_offset, y_offset, offset_normalized=True, ignore_last_dim=True)
        out[k] = attn_map.flatten(2, 3)
    return out


def shift_saved_attns(saved_attns, offset, guidance_attn_keys, **kwargs):
    """attn.py:66-70: per timestep."""
    return [shift_saved_attns_item(item, offset, guidance_attn_keys, **kwargs) for item in saved_attns]
