"""ORACLE TEST INFRASTRUCTURE (build container only) — the RATIO-based attention energy.

`add_ca_loss_per_attn_map_to_loss` (utils/guidance.py:91) defaults to `use_ratio_based_loss=True`; the reference's
`generation/backward_guidance.py:99-112` never sets the flag, so its layout-guidance baseline minimises the ratio
branch (:118-130).  This script calls the reference's OWN, unmodified `guidance.compute_ca_lossv3` WITHOUT the flag on
the maps already committed in tests/golden/energy.npz and records value + map gradients, for
  * the canonical two-box layout (one box per phrase),
  * a three-level layout (several boxes per phrase -> union mask, guidance.py:108-114),
  * an explicit `use_ratio_based_loss=True` call next to a reference-attention term (mixed, weight 0.5),
and checks oracle/restate.py against them.  -> tests/golden/energy_ratio.npz

    python oracle/make_golden_ratio.py
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402
import restate as R  # noqa: E402

KEYS = R.DEFAULT_GUIDANCE_ATTN_KEYS
BOXES_XYWH = [("a white deer", [74, 177, 183, 235]), ("a gray bear", [314, 193, 189, 216])]
BBOXES = [[x / 512, y / 512, (x + w) / 512, (y + h) / 512] for _, (x, y, w, h) in BOXES_XYWH]
BBOXES3 = [[BBOXES[0], [0.05, 0.05, 0.3, 0.35]], [BBOXES[1]]]
OBJ_POS = [[1, 2, 3], [5, 6, 7]]
WORD_TOK = [3, 7]


def ks(k):
    return "_".join(str(x) for x in k)


def main():
    H.setup()
    from utils import guidance
    g = np.load(os.path.join(ROOT, "tests", "golden", "energy.npz"))
    arrs, report = {}, {}
    refs = [[None, {k: torch.from_numpy(g[f"ref_{o}_{ks(k)}"]) for k in KEYS}] for o in range(2)]
    cases = (("two_level", BBOXES, {}),
             ("three_level", BBOXES3, {}),
             ("with_ref", BBOXES, dict(use_ratio_based_loss=True, ref_ca_saved_attns=refs, ref_ca_word_token_only=True,
                                       ref_ca_last_token_only=True, word_token_indices=WORD_TOK, index=1,
                                       ref_ca_loss_weight=0.5)))
    for tag, boxes, kw in cases:
        maps = {k: torch.from_numpy(g["map_" + ks(k)]).requires_grad_(True) for k in KEYS}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss = guidance.compute_ca_lossv3(saved_attn=maps, bboxes=boxes, object_positions=OBJ_POS,
                                              guidance_attn_keys=KEYS, **kw)
        grads = torch.autograd.grad(loss, [maps[k] for k in KEYS])
        maps2 = {k: torch.from_numpy(g["map_" + ks(k)]).requires_grad_(True) for k in KEYS}
        loss2 = R.compute_ca_lossv3(maps2, boxes, OBJ_POS, KEYS, **kw)
        grads2 = torch.autograd.grad(loss2, [maps2[k] for k in KEYS])
        report[f"{tag}/loss"] = float((loss2 - loss).abs() / loss.abs())
        report[f"{tag}/grad"] = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(grads2, grads))
        arrs[f"loss_{tag}"] = loss.detach().numpy()
        for k, gr in zip(KEYS, grads):
            arrs[f"grad_{tag}_{ks(k)}"] = gr.numpy()
        print(tag, "reference loss", float(loss), report[f"{tag}/loss"], report[f"{tag}/grad"])
    assert max(report.values()) < 1e-5, report
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "energy_ratio.npz"), **arrs)
    print("wrote energy_ratio.npz")


if __name__ == "__main__":
    main()
