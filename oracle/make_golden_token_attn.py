"""ORACLE TEST INFRASTRUCTURE (build container only) — known answers for `utils.attn.get_token_attnv2`
(utils/attn.py:9-38): the reference's own function on seeded saved-attention lists
-> tests/golden/token_attn.npz (inputs and outputs; replayed by tests/test_align_host.py against the drop-in
`utils.attn.get_token_attnv2` and the pipeline's `_token_attn`).

    python oracle/make_golden_token_attn.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

attn_mod = H.ref_modules()["attn"]
KEY = ("down", 2, 1, 0)
T, HEADS, HW = 6, 4, 64


def seeded(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)).softmax(-1)


arrs = {}
pair = [{KEY: seeded((2, HEADS, HW, 9), 300 + t)} for t in range(T)]           # [uncond; cond], 9 text tokens
cond = [{KEY: seeded((1, HEADS, HW, 1), 400 + t)} for t in range(T)]           # condition only, one saved token
arrs["pair"] = torch.stack([s[KEY] for s in pair]).numpy()
arrs["cond"] = torch.stack([s[KEY] for s in cond]).numpy()
cases = []
for start in (0, 2, 5):
    for tok in (0, 3, 8):
        out = attn_mod.get_token_attnv2(tok, pair, KEY, attn_aggregation_step_start=start, return_np=True)
        arrs[f"pair_s{start}_t{tok}"] = out
    out = attn_mod.get_token_attnv2(0, cond, KEY, attn_aggregation_step_start=start, input_ca_has_condition_only=True)
    assert isinstance(out, torch.Tensor)
    arrs[f"cond_s{start}"] = out.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "token_attn.npz"), **arrs)
print("wrote tests/golden/token_attn.npz", {k: v.shape for k, v in arrs.items()})
