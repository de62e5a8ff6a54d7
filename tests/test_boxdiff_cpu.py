"""CPU suite for the BoxDiff row: the oracle restatement (oracle/restate_boxdiff.py) against goldens
recorded from the reference's OWN utils/boxdiff.py and generation/boxdiff.run (oracle/make_golden_boxdiff.py), and the
host tables of the HIP kernel (lgd_amd.energy.BoxDiffTables) against the reference's counting rules."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lgd_amd  # noqa: E402,F401
from lgd_amd import weights  # noqa: E402
import restate as R  # noqa: E402
import restate_boxdiff as B  # noqa: E402
from boxdiff_maps import make_maps  # noqa: E402

KEYS = B.BOXDIFF_GUIDANCE_ATTN_KEYS
CASES = ("hw256", "hw64", "two_boxes", "edge", "tiny_box")


def test_oracle_boxdiff_energy_matches_the_reference_function():
    """compute_ca_loss_boxdiff (utils/boxdiff.py:121-196) incl. Python's max(0, nan) = 0 for a box too small for its
    top-k: value and the gradient on all five maps, to fp32 round-off."""
    g = np.load(os.path.join(GOLD, "boxdiff_energy.npz"))
    for name in CASES:
        spec = json.loads(str(g[f"{name}_spec"]))
        maps = make_maps(spec["side"], spec["heads"], spec["seed"])             # the golden's inputs, regenerated from the seed
        assert float(maps[tuple(KEYS[0])].double().sum()) == float(g[f"{name}_map0_checksum"]), name
        leaves = {tuple(k): maps[tuple(k)].clone().requires_grad_(True) for k in KEYS}
        loss = B.compute_ca_loss_boxdiff(leaves, spec["bboxes"], spec["pos"], KEYS)
        grads = torch.autograd.grad(loss, [leaves[tuple(k)] for k in KEYS])
        assert abs(float(loss.detach()) - float(g[f"{name}_loss"])) <= 1e-6 * abs(float(g[f"{name}_loss"])), name
        ref = torch.from_numpy(g[f"{name}_grad"])                                # [HW, 77]: the same for every key and head
        for i, gr in enumerate(grads):
            assert float((gr - ref[None, None]).abs().max()) <= 1e-6 * float(ref.abs().max()), (name, i)


def test_gaussian_kernel_is_the_reference_smoothing_kernel():
    """GaussianSmoothing(1, 3, 0.5) of utils/attn.py:92-110 — (x / (2 sigma))^2 in the exponent, as written there."""
    k = B.gaussian_kernel(3, 0.5)
    assert abs(float(k.sum()) - 1) < 1e-6 and torch.allclose(k, k.t())
    assert abs(float(k[1, 1] / k[0, 1]) - float(np.exp(1.0))) < 1e-5          # exp(-(1 / (2 * 0.5))^2) = e^-1 off-centre
    from lgd_amd.energy import gaussian_kernel
    assert torch.equal(gaussian_kernel(3, 0.5), k)


def test_oracle_boxdiff_run_matches_the_reference_run():
    """generation/boxdiff.run on the tiny network (CPU fp32): per-step latents entering / leaving the BoxDiff step, its
    loss, the final latents — the restatement of pipelines.py:129-247 with use_boxdiff=True."""
    g = np.load(os.path.join(GOLD, "run_boxdiff_tiny.npz"))
    cfg = weights.CONFIGS["tiny"]
    sd = weights.synth_state_dict(cfg, 0)
    cd = dict(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
              attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps,
              gligen_positive_len=cfg.gligen_positive_len)
    for tag in "ab":
        ehs = torch.from_numpy(g[f"{tag}_text_embeddings"])
        kw = json.loads(str(g[f"{tag}_kwargs"]))
        bboxes, pos = json.loads(str(g[f"{tag}_bboxes"])), json.loads(str(g[f"{tag}_object_positions"]))
        gk = json.loads(str(g[f"{tag}_guidance_kwargs"]))
        assert [tuple(k) for k in gk["guidance_attn_keys"]] == KEYS and gk["max_index_step"] == kw["overall_max_index_step"]
        tr, starts = [], []
        out = B.generate_boxdiff(sd, cd, R.DDIM(), torch.from_numpy(g[f"{tag}_latents_in"]), (ehs, None, ehs[1:2]), 8, bboxes,
                                 pos, max_index_step=kw["overall_max_index_step"], trace=tr, starts=starts)
        n = kw["overall_max_index_step"]
        assert len(tr) == n
        ref_l = g[f"{tag}_losses"][:n]
        assert np.abs(np.array([t["loss"] for t in tr]) - ref_l).max() <= 2e-4 * np.abs(ref_l).max(), tag
        for i in range(8):
            e = float((starts[i] - torch.from_numpy(g[f"{tag}_starts"][i])).abs().max() / np.abs(g[f"{tag}_starts"][i]).max())
            assert e < 2e-4, (tag, i, e)
        e = float((out - torch.from_numpy(g[f"{tag}_final_latents"])).abs().max() / np.abs(g[f"{tag}_final_latents"]).max())
        assert e < 2e-4, (tag, e)


def test_boxdiff_tables_follow_the_reference_counting_rules():
    """Masks, corner masks, projections and the top-k counts (`(mask.sum() * P).long()`, utils/boxdiff.py:80,85) of the
    kernel's host tables; a merged batch keeps every image's items adjacent; k = 0 for a box of fewer than 1 / P pixels."""
    from lgd_amd.energy import BoxDiffTables
    hw = {k: 256 for k in KEYS}
    boxes = [[0.5, 0.5, 0.62, 0.62], [[0.05, 0.5, 0.3, 0.9], [0.4, 0.45, 0.7, 0.85]]]
    t = BoxDiffTables("cpu", boxes, [[2, 3], [8]], KEYS, hw, heads=8)
    assert t.side == 16 and t.n_items == 3 and t.max_items == 3
    items = t.items.tolist()
    assert [r[0] for r in items] == [2, 3, 8] and items[0][1] == items[1][1] != items[2][1]
    m0 = R.box_mask(boxes[0], 16, 16)
    assert torch.equal(t.masks[0, 0].reshape(16, 16), m0) and items[0][2] == int((m0.sum() * 0.2).long()) == 0
    m1 = R.box_mask(boxes[1], 16, 16)
    assert items[2][2] == int((m1.sum() * 0.2).long()) and items[2][3] == int(((1 - m1).sum() * 0.2).long())
    assert torch.equal(t.masks[1, 2, :16], m1.max(dim=0).values) and torch.equal(t.masks[1, 2, 16:32], m1.max(dim=1).values)
    x0, y0, x1, y1 = R.scale_proportion(boxes[0], 16, 16)
    cx = torch.zeros(16)
    cx[max(x0 - 1, 0):x0 + 2] = 1
    cx[max(x1 - 1, 0):min(x1 + 2, 16)] = 1
    assert torch.equal(t.masks[0, 1, :16], cx)
    both = BoxDiffTables.merged([t, None, BoxDiffTables("cpu", [[0.1, 0.1, 0.6, 0.7]], [[5, 6]], KEYS, hw, heads=8)])
    assert both.groups.tolist() == [[0, 3], [3, 0], [3, 2]] and both.n_samples == 3 and both.max_items == 3
    assert both.items[3:, 1].tolist() == [2, 2]                       # mask ids shifted behind the first image's two masks
    import pytest
    with pytest.raises(RuntimeError):
        BoxDiffTables("cpu", boxes, [[2, 3], [8]], KEYS, {**hw, KEYS[0]: 64}, heads=8)      # mixed resolutions
    with pytest.raises(RuntimeError):
        BoxDiffTables("cpu", boxes, [[0], [8]], KEYS, hw, heads=8)                               # token 0 is not a phrase token
