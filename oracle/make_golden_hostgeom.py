"""ORACLE TEST INFRASTRUCTURE — known-answer vectors for the host-side box / mask geometry rules
(SURVEY.md §8a rows G1 pixel rounding, H1-H3), produced by the reference's own utils/utils.py
functions on seeded random inputs, through oracle/ref_harness.py.

    python oracle/make_golden_hostgeom.py     # build container only; writes tests/golden/hostgeom.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

U = rh.ref_modules()["utils"]
rng = np.random.RandomState(0)
arrs = {}

# ---- boxes: random xyxy in [0,1] incl. boxes touching / crossing the border and .5-pixel edges
boxes = []
for _ in range(300):
    x0, y0 = rng.uniform(-0.05, 0.9, 2)
    w, h = rng.uniform(0.01, 0.7, 2)
    boxes.append([x0, y0, x0 + w, y0 + h])
for k in range(40):                       # coordinates that land exactly on .5 pixels at 64 / 16 / 8
    boxes.append([(2 * k + 1) / 128, (2 * k + 3) / 128, (2 * k + 1) / 128 + 0.2578125, (2 * k + 3) / 128 + 0.1171875])
boxes = np.array(boxes, dtype=np.float64)
arrs["boxes"] = boxes
for hw in (64, 32, 16, 8):
    arrs[f"rect_{hw}"] = np.array([U.scale_proportion(list(b), hw, hw) for b in boxes], dtype=np.int64)
    arrs[f"rect_legacy_{hw}"] = np.array([U.scale_proportion(list(b), hw, hw, use_legacy=True) for b in boxes], dtype=np.int64)
arrs["mask_64_first20"] = np.stack([U.proportion_to_mask(list(b), 64, 64).numpy() for b in boxes[:20]])
arrs["centered_h"] = np.array([U.get_centered_box(list(b), horizontal_center_only=True) for b in boxes])
arrs["centered_c"] = np.array([U.get_centered_box(list(b), horizontal_center_only=False) for b in boxes])
arrs["centered_c3"] = np.array([U.get_centered_box(list(b), horizontal_center_only=False, vertical_center=0.3) for b in boxes])
arrs["centered_f"] = np.array([U.get_centered_box(list(b), horizontal_center_only=False, vertical_placement="floor_padding",
                                                  floor_padding=0.2) for b in boxes])

# ---- masks: random blobs -> bounding box (enlarged or not), box mask, mass centre
masks, bb1, bb0, bmask, cen, cen_n = [], [], [], [], [], []
for i in range(60):
    H, W = [(64, 64), (32, 32), (16, 24)][i % 3]
    m = np.zeros((H, W), dtype=bool)
    for _ in range(rng.randint(1, 4)):
        y0, x0 = rng.randint(0, H - 1), rng.randint(0, W - 1)
        m[y0:y0 + rng.randint(1, H // 2), x0:x0 + rng.randint(1, W // 2)] = True
    mt = torch.from_numpy(m)
    pad = np.zeros((64, 64), dtype=bool)
    pad[:H, :W] = m
    masks.append(pad)
    bb1.append([int(v) for v in U.binary_mask_to_box(mt)])
    bb0.append([int(v) for v in U.binary_mask_to_box(mt, enlarge_box_by_one=False, w_scale=2, h_scale=3)])
    bm = np.zeros((64, 64), dtype=np.float32)
    bm[:H, :W] = U.binary_mask_to_box_mask(mt, to_device=False).numpy()
    bmask.append(bm)
    cen.append(U.binary_mask_to_center(mt))
    cen_n.append(U.binary_mask_to_center(mt, normalize=True))
arrs.update(masks=np.stack(masks), mask_hw=np.array([[(64, 64), (32, 32), (16, 24)][i % 3] for i in range(60)]),
            bbox_enlarged=np.array(bb1), bbox_plain_scaled=np.array(bb0), box_masks=np.stack(bmask),
            centers=np.array(cen, dtype=np.float64), centers_norm=np.array(cen_n, dtype=np.float64))
a, b = masks[0], np.stack(masks[1:10])
arrs["iou_0_vs_1to9"] = U.iou(a, b)

# ---- shifts: latents-like (.., H, W) and attention-like (.., h, w, T) tensors
g = torch.Generator().manual_seed(0)
lat = torch.randn((3, 1, 4, 64, 64), generator=g)
att = torch.randn((2, 1, 2, 16, 16, 3), generator=g)
msk = torch.from_numpy(masks[0])
arrs.update(shift_lat=lat.numpy(), shift_att=att.numpy())
offs = [(0.0, 0.0), (0.13, -0.07), (-0.31, 0.26), (0.5, 0.5), (-0.0624, 0.0626), (0.9, -0.9), (0.1875, 0.3125)]
arrs["shift_offsets"] = np.array(offs)
for i, (dx, dy) in enumerate(offs):
    arrs[f"shift_lat_{i}"] = U.shift_tensor(lat, dx, dy, offset_normalized=True).numpy()
    arrs[f"shift_att_{i}"] = U.shift_tensor(att, dx, dy, offset_normalized=True, ignore_last_dim=True).numpy()
    arrs[f"shift_msk_{i}"] = U.shift_tensor(msk, dx, dy, offset_normalized=True).numpy()
for i, (dx, dy) in enumerate([(3, -5), (-64, 0), (0, 63), (-7, 9)]):
    arrs[f"shift_px_{i}"] = U.shift_tensor(lat, dx, dy).numpy()
arrs["shift_px_offsets"] = np.array([(3, -5), (-64, 0), (0, 63), (-7, 9)])
arrs["expand"] = np.array(U.expand_overall_bboxes([[[0.1, 0.2, 0.3, 0.4]], [[0.5, 0.5, 0.6, 0.7], [0.0, 0.1, 0.2, 0.3]]]))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hostgeom.npz"), **arrs)
print("wrote tests/golden/hostgeom.npz with", len(arrs), "arrays")
