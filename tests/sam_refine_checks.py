"""Shared by tests/test_sam_refine_host.py (CPU, emulated kernels, fp32) and tests/test_sam_gpu.py (HIP kernels): replays
the calls oracle/make_golden_sam.py made on the reference's own models/sam.py and compares with tests/golden/sam_refine.npz."""
import os

import numpy as np

import sam_cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sam_refine.npz")
KW = dict(height=512, width=512, H=64, W=64, discourage_mask_below_confidence=0.85, discourage_mask_below_coarse_iou=0.2)
ATTN_KW = dict(use_box_input=False, gaussian_sigma=1.5, mask_th_for_box=0.05, n_erode_dilate_mask_for_box=1,
               mask_th_for_point=0.25, **KW)


def replay(dsam, sam_model_dict, min_agree, conf_tol):
    g = np.load(GOLD)
    images, boxes, attn = sam_cases.refine_inputs()
    worst = 1.0

    def same(name, mask, conf):
        nonlocal worst
        want = g[name + "_mask"]
        agree = float((np.asarray(mask).astype(bool) == want).mean())
        worst = min(worst, agree)
        assert agree >= min_agree, (name, agree, int(want.sum()), int(np.asarray(mask).sum()))
        assert abs(float(conf) - float(g[name + "_conf"])) <= conf_tol, (name, float(conf), float(g[name + "_conf"]))

    masks, conf = dsam.sam_box_input(sam_model_dict, image=[images[0]], input_boxes=[[list(np.array(boxes[0][0]) * 512)]],
                                     target_mask_shape=(64, 64))
    assert masks[0][0].shape == (3, 64, 64) and masks[0][0].dtype == bool
    assert float((masks[0][0] == g["cand_masks"]).mean()) >= min_agree
    assert np.abs(conf - g["cand_conf"]).max() <= conf_tol
    n = 0
    for ii, per_image in enumerate(boxes):
        for box in per_image:
            m, c = dsam.sam_refine_box(sam_input_image=images[ii], box=box, model_dict=sam_model_dict, verbose=False, **KW)
            same(f"box{n}", m, c)
            n += 1
    mm, cc = dsam.sam_refine_boxes(images, boxes, sam_model_dict, verbose=False, **KW)
    assert float((np.array(mm) == g["batched_masks"]).mean()) >= min_agree
    assert np.abs(np.array(cc, dtype=np.float32) - g["batched_conf"]).max() <= conf_tol
    n = 0
    for ii, per_image in enumerate(boxes):
        for _ in per_image:
            m, c = dsam.sam_refine_attn(sam_input_image=images[ii], token_attn_np=attn[n], model_dict=sam_model_dict,
                                        verbose=False, **ATTN_KW)
            same(f"attn{n}", m, c)
            n += 1
    return worst
