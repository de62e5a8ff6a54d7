"""utils/guidance.py of the reference: phrase -> token-index lookup (host, tokenizer) and the
cross-attention energy.  `compute_ca_lossv3` keeps the reference signature and returns a scalar tensor
that is differentiable w.r.t. the maps in `saved_attn` — the value and the map gradients come from the
single-launch HIP energy kernel (lgd_amd.energy), not from a chain of torch.topk calls."""
import math

import torch

from lgd_amd.energy import EnergyTables


def get_token_map(tokenizer, prompt, verbose=False, padding="do_not_pad"):
    """guidance.py:10-30: token strings of a prompt (no padding, BOS at 0)."""
    ids = tokenizer([prompt], padding=padding, max_length=77, return_tensors="np")['input_ids'][0]
    return [tokenizer._convert_id_to_token(i) for i in ids.tolist()]


def get_phrase_indices(tokenizer, prompt, phrases, verbose=False, words=None, include_eos=False, token_map=None,
                       return_word_token_indices=False, add_suffix_if_not_found=False):
    """guidance.py:32-89: positions of each phrase's tokens inside the prompt's token sequence (found by
    substring search over the space-joined token strings); missing phrases are appended after "| "."""
    for obj in phrases:
        if obj not in prompt:
            prompt += "| " + obj
    if token_map is None:
        token_map = get_token_map(tokenizer, prompt=prompt, verbose=verbose)
    joined = " ".join(token_map)
    object_positions, word_token_indices = [], []
    for obj_ind, obj in enumerate(phrases):
        ptoks = get_token_map(tokenizer, prompt=obj, verbose=verbose)[1:-1]      # strip <bos>/<eos>
        needle = " ".join(ptoks)
        first = len(joined[:joined.index(needle) - 1].split(" "))
        pos = list(range(first, first + len(ptoks)))
        if include_eos:
            pos.append(token_map.index(tokenizer.eos_token))
        object_positions.append(pos)
        if return_word_token_indices:
            if words is None:
                idx = object_positions[0][-1]
            else:
                wtoks = get_token_map(tokenizer, prompt=words[obj_ind], verbose=verbose)
                idx = first + ptoks.index(wtoks[-2])                          # last token of the word
            word_token_indices.append(idx)
    out = [object_positions]
    if return_word_token_indices:
        out.append(word_token_indices)
    if add_suffix_if_not_found:
        out.append(prompt)
    return out[0] if len(out) == 1 else tuple(out)


class _EnergyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tables, index, *maps):
        dev = maps[0].device
        m32 = {k: m.detach().reshape(m.shape[-3:]).float().contiguous().unsqueeze(0) for k, m in zip(tables.keys, maps)}
        g32 = {k: torch.zeros_like(v) for k, v in m32.items()}
        tables.bind(m32, g32)
        dyn = torch.tensor([index or 0, 0, 0, 0], dtype=torch.int32, device=dev)
        loss = tables.run(dyn, grad_scale=1.0).clone()
        ctx.grads = [g32[k] for k in tables.keys]
        ctx.shapes = [(m.shape, m.dtype) for m in maps]
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        return (None, None) + tuple((g.reshape(s) * gout).to(dt) for g, (s, dt) in zip(ctx.grads, ctx.shapes))


def compute_ca_lossv3(saved_attn, bboxes, object_positions, guidance_attn_keys, ref_ca_saved_attns=None,
                      ref_ca_last_token_only=True, ref_ca_word_token_only=False, word_token_indices=None, index=None,
                      ref_ca_loss_weight=1.0, verbose=False, **kwargs):
    """guidance.py:244-286.  `use_ratio_based_loss` (in **kwargs, forwarded to add_ca_loss_per_attn_map_to_loss in
    the reference) defaults to True as at guidance.py:91: ratio-based branch :118-130; False: max-based :131-145."""
    keys = [tuple(k) for k in guidance_attn_keys]
    dev = saved_attn[keys[0]].device if keys else "cuda"
    if len(bboxes) == 0:
        return torch.tensor(0., device=dev)
    maps = [saved_attn[k] for k in keys]
    heads = maps[0].shape[-3]
    hw = {k: m.shape[-2] for k, m in zip(keys, maps)}
    ekw = {k: kwargs[k] for k in ("use_ratio_based_loss", "fg_top_p", "bg_top_p", "fg_weight", "bg_weight") if k in kwargs}
    tables = EnergyTables(dev, bboxes, object_positions, keys, hw, heads, maps[0].shape[-1], loss_scale=1.0,
                          ref_boxes=ref_ca_saved_attns is not None, ref_ca_loss_weight=ref_ca_loss_weight,
                          ref_ca_word_token_only=ref_ca_word_token_only,
                          ref_ca_last_token_only=ref_ca_last_token_only, word_token_indices=word_token_indices, **ekw)
    if ref_ca_saved_attns is not None and tables.n_refs:
        refs = torch.zeros((1, tables.n_refs, heads, tables.max_hw), device=dev)
        from collections.abc import Iterable
        for rid, (o, bi, ki) in enumerate(tables.ref_slots):
            per_box = ref_ca_saved_attns[o]
            if not isinstance(bboxes[o][0], Iterable):
                per_box = [per_box]
            r = per_box[bi][index][keys[ki]][0, :, :, 0].to(dev).float()
            refs[0, rid, :, :r.shape[1]] = r
        tables.set_refs(refs)
        index = 0
    return _EnergyFn.apply(tables, index, *maps)
