"""ORACLE TEST INFRASTRUCTURE — golden vectors for the centred-box + re-alignment variant of training-free
LMD (generation/lmd.py:314-324, 438-452, 489-497), produced by the reference's own
utils (get_centered_box, compose_latents_with_alignment -> align_with_bboxes/shift_tensor, shift_saved_attns)
through oracle/ref_harness.py.

    python oracle/make_golden_align.py        # build container only; writes tests/golden/align_host.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

mods = rh.ref_modules()
utils, latents_mod, attn_mod = mods["utils"], mods["latents"], mods["attn"]
from easydict import EasyDict  # noqa: E402
from diffusers import DDIMScheduler  # noqa: E402

KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]
BOXES_XYWH = [[74, 177, 183, 235], [314, 193, 189, 216], [20, 300, 120, 150]]
BBOXES = [[x / 512, y / 512, (x + w) / 512, (y + h) / 512] for x, y, w, h in BOXES_XYWH]
OVERALL = [[BBOXES[0]], [BBOXES[1], BBOXES[2]]]          # phrase 2 has two boxes
L, STEPS, HEADS = 64, 3, 2


def seeded(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


class U:
    class config:
        in_channels = 4


md = EasyDict(unet=U(), scheduler=DDIMScheduler(), dtype=torch.float32)
arrs = {}
for tag, kw in (("lmd", dict(horizontal_center_only=False, vertical_placement="floor_padding", floor_padding=0.2)),
                ("lmdplus", dict(horizontal_center_only=True))):
    so = [utils.get_centered_box(b, **kw) for b in BBOXES]
    arrs[f"so_boxes_{tag}"] = np.array(so, dtype=np.float64)
so = [utils.get_centered_box(b, horizontal_center_only=False, vertical_placement="floor_padding", floor_padding=0.2)
      for b in BBOXES]
bg = seeded((1, 4, L, L), 5)
lall = [seeded((STEPS + 1, 1, 4, L, L), 90 + i) for i in range(3)]
masks = [utils.proportion_to_mask(b, L, L).bool() for b in so]
saved = [[{k: seeded((1, HEADS, {"mid": 64}.get(k[0], 256), 1), 1000 + 10 * b + t + 100 * ki) for ki, k in enumerate(KEYS)}
          for t in range(STEPS)] for b in range(3)]
for hso in (False, True):
    comp, fg, offs = latents_mod.compose_latents_with_alignment(
        md, [x.clone() for x in lall], [m.clone() for m in masks], STEPS, 1, 512, 512, latents_bg=bg,
        align_with_overall_bboxes=True, overall_bboxes=OVERALL, horizontal_shift_only=hso)
    t = "h" if hso else "xy"
    arrs[f"composed_{t}"] = comp.numpy()
    arrs[f"fg_idx_{t}"] = fg.numpy()
    arrs[f"offsets_{t}"] = np.array(offs, dtype=np.float64)
    for b in range(3):
        sh = attn_mod.shift_saved_attns(saved[b], offs[b], guidance_attn_keys=KEYS, horizontal_shift_only=hso)
        for ki, k in enumerate(KEYS):
            arrs[f"shifted_{t}_{b}_{ki}"] = torch.stack([s[k] for s in sh]).numpy()   # [T,1,H,HW,1]
arrs["bg"] = bg.numpy()
for i in range(3):
    arrs[f"lall{i}"] = lall[i].numpy()
    for ki, k in enumerate(KEYS):
        arrs[f"saved_{i}_{ki}"] = torch.stack([s[k] for s in saved[i]]).numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "align_host.npz"), **arrs)
print("wrote tests/golden/align_host.npz", {k: v.shape for k, v in arrs.items() if k.startswith(("composed", "offsets"))})
