"""ORACLE TEST INFRASTRUCTURE (build container only) — goldens of the SAM mask-refinement rules.

Runs the reference's OWN, unmodified `models/sam.py` (sam_refine_box / sam_refine_boxes :174-213, sam_refine_attn
:125-172, through sam() :25-55 and select_mask() :67-111) on CPU through oracle/ref_harness.py with
  * the Hugging Face `SamModel` + `SamProcessor` it is written against ([ext] transformers; seeded random parameters of
    tests/sam_cases.py — no checkpoints in the sandbox — at sam-vit-base geometry with a short vision tower),
  * `cv2` replaced by oracle/stubs/cv2.py (same-size resize = identity, the only case the plugins produce),
and writes tests/golden/sam_refine.npz: for every call the selected 64x64 mask, its confidence, and the three
candidate masks with their predicted IoUs.

    python oracle/make_golden_sam.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness as H  # noqa: E402
import sam_cases  # noqa: E402

KW = dict(height=512, width=512, H=64, W=64, discourage_mask_below_confidence=0.85, discourage_mask_below_coarse_iou=0.2)
# generation/lmd.py:39-49,243-250 defaults (point input; with use_box_input=True the reference hands the processor a
# two-level box list, which transformers refuses: "Input boxes must be a list of list of list of floating points")
ATTN_KW = dict(use_box_input=False, gaussian_sigma=1.5, mask_th_for_box=0.05, n_erode_dilate_mask_for_box=1,
               mask_th_for_point=0.25, **KW)


def main():
    H.setup()
    import transformers
    from models import sam as ref_sam
    ref_sam.torch_device = "cpu"
    _autocast = torch.autocast
    torch.autocast = lambda *a, **k: _autocast("cpu", enabled=False)          # models/sam.py:38: stay fp32 on the CPU
    hf = sam_cases.build_refine_hf(transformers)
    md = dict(sam_model=hf, sam_processor=transformers.SamProcessor(transformers.SamImageProcessor()))
    images, boxes, attn = sam_cases.refine_inputs()
    out = {}
    # candidates straight from sam() for one image / one box
    masks, conf = ref_sam.sam_box_input(md, image=[images[0]], input_boxes=[[list(np.array(boxes[0][0]) * 512)]],
                                        target_mask_shape=(64, 64))
    out["cand_masks"], out["cand_conf"] = masks[0][0], conf
    n = 0
    for ii, per_image in enumerate(boxes):                       # the way generation/lmd_plus.py:122 calls it
        for box in per_image:
            m, c = ref_sam.sam_refine_box(sam_input_image=images[ii], box=box, model_dict=md, verbose=False, **KW)
            out[f"box{n}_mask"], out[f"box{n}_conf"] = m, np.float32(c)
            n += 1
    mm, cc = ref_sam.sam_refine_boxes(images, boxes, md, verbose=False, **KW)       # batched (shared-confidence quirk)
    out["batched_masks"] = np.array([[m for m in row] for row in mm])
    out["batched_conf"] = np.array(cc, dtype=np.float32)
    n = 0
    for ii, per_image in enumerate(boxes):                       # generation/lmd.py:141
        for _ in per_image:
            m, c = ref_sam.sam_refine_attn(sam_input_image=images[ii], token_attn_np=attn[n], model_dict=md, verbose=False,
                                           **ATTN_KW)
            out[f"attn{n}_mask"], out[f"attn{n}_conf"] = m, np.float32(c)
            n += 1
    path = os.path.join(ROOT, "tests", "golden", "sam_refine.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    for k, v in out.items():
        if k.endswith("_mask"):
            print(k, "area", int(v.sum()), "conf", float(out[k[:-5] + "_conf"]))
    print("candidate areas", out["cand_masks"].sum(axis=(1, 2)), "conf", out["cand_conf"])


if __name__ == "__main__":
    main()
