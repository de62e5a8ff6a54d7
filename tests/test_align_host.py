"""Centred-box + re-alignment wiring of the LMD / LMD+ pipelines (lmd.py:314-324,438-452,489-497; SURVEY.md §8a
rows H2/H3) against goldens produced by the reference's own utils (oracle/make_golden_align.py): integer /
copy work, so the comparison is bit-exact."""
import os

import numpy as np
import pytest
import torch

import lgd_amd  # noqa: F401
from lgd_amd import pipeline
from lgd_amd.hostprep import compose, proportion_to_mask

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align_host.npz")
KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]
BOXES_XYWH = [[74, 177, 183, 235], [314, 193, 189, 216], [20, 300, 120, 150]]
BBOXES = [[x / 512, y / 512, (x + w) / 512, (y + h) / 512] for x, y, w, h in BOXES_XYWH]


class Lay:
    boxes = [tuple(b) for b in BBOXES]
    overall_groups = [[0], [1, 2]]


def test_centered_boxes_match_reference():
    g = np.load(GOLD)
    lmd = pipeline._centered_so_boxes(Lay, True, horizontal_center_only=False, vertical_placement="floor_padding",
                                      floor_padding=0.2)
    plus = pipeline._centered_so_boxes(Lay, True, horizontal_center_only=True)
    assert np.array_equal(np.array(lmd, dtype=np.float64), g["so_boxes_lmd"])
    assert np.array_equal(np.array(plus, dtype=np.float64), g["so_boxes_lmdplus"])
    assert pipeline._centered_so_boxes(Lay, False) == [list(b) for b in Lay.boxes]


@pytest.mark.parametrize("horizontal_shift_only", [False, True])
def test_alignment_composition_and_ref_map_shift_match_reference(horizontal_shift_only):
    g = np.load(GOLD)
    t = "h" if horizontal_shift_only else "xy"
    so = pipeline._centered_so_boxes(Lay, True, horizontal_center_only=False, vertical_placement="floor_padding",
                                     floor_padding=0.2)
    d = dict(latents_all=[torch.from_numpy(g[f"lall{i}"]) for i in range(3)],
             masks=[proportion_to_mask(b, 64, 64).bool() for b in so],
             saved=[{k: torch.from_numpy(g[f"saved_{i}_{ki}"]) for ki, k in enumerate(KEYS)} for i in range(3)])
    d["saved"][0][("down", 2, 1, 0)] = torch.ones(3, 1, 2, 256, 1)            # non-guidance key: left alone
    pipeline._align_stage_a(d, Lay, KEYS, True, horizontal_shift_only)
    comp, fg = compose(d["latents_all"], d["masks"], 3, torch.from_numpy(g["bg"]))
    assert np.array_equal(comp.numpy(), g[f"composed_{t}"])
    assert np.array_equal(fg.numpy(), g[f"fg_idx_{t}"])
    for b in range(3):
        for ki, k in enumerate(KEYS):
            assert np.array_equal(d["saved"][b][k].numpy(), g[f"shifted_{t}_{b}_{ki}"]), (b, k)
    assert torch.equal(d["saved"][0][("down", 2, 1, 0)], torch.ones(3, 1, 2, 256, 1))


def test_alignment_off_is_identity():
    g = np.load(GOLD)
    lat = [torch.from_numpy(g[f"lall{i}"]) for i in range(3)]
    d = dict(latents_all=list(lat), masks=[proportion_to_mask(b, 64, 64).bool() for b in BBOXES], saved=[{}, {}, {}])
    pipeline._align_stage_a(d, Lay, KEYS, False, False)
    assert all(a is b for a, b in zip(d["latents_all"], lat))


def _dropin_attn():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dropin_utils_attn", os.path.join(root, "llm-groundeddiffusion_amd", "dropin", "utils", "attn.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_get_token_attnv2_known_answers_from_the_reference_function():
    """utils/attn.py:9-38 (row H3): mean over the steps from `attn_aggregation_step_start` on and over heads of one
    token's map — the reference's own outputs on seeded inputs (oracle/make_golden_token_attn.py).  The drop-in sums in
    a different order (per step, then heads), hence 1e-6 instead of bit equality."""
    g = np.load(os.path.join(os.path.dirname(GOLD), "token_attn.npz"))
    attn = _dropin_attn()
    key = ("down", 2, 1, 0)
    pair = [{key: torch.from_numpy(x)} for x in g["pair"]]
    cond = [{key: torch.from_numpy(x)} for x in g["cond"]]
    for start in (0, 2, 5):
        for tok in (0, 3, 8):
            got = attn.get_token_attnv2(tok, pair, key, attn_aggregation_step_start=start, return_np=True)
            want = g[f"pair_s{start}_t{tok}"]
            assert isinstance(got, np.ndarray) and got.shape == want.shape == (8, 8)
            assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
        got = attn.get_token_attnv2(0, cond, key, attn_aggregation_step_start=start, input_ca_has_condition_only=True)
        assert isinstance(got, torch.Tensor)
        assert (got - torch.from_numpy(g[f"cond_s{start}"])).abs().max() <= 1e-6 * float(np.abs(g[f"cond_s{start}"]).max())
        # the pipeline's own form of the same rule (lgd_amd.pipeline._token_attn: device tensor [T, 1, heads, HW, 1])
        got2 = pipeline._token_attn({pipeline.OBJ_ATTN_KEY: torch.from_numpy(g["cond"])}, start)
        assert np.abs(got2 - g[f"cond_s{start}"]).max() <= 1e-6 * np.abs(g[f"cond_s{start}"]).max()
    # the reference asserts on the batch layout (attn.py:25-30) and cannot aggregate an empty tail
    with pytest.raises(AssertionError):
        attn.get_token_attnv2(0, pair, key, attn_aggregation_step_start=0, input_ca_has_condition_only=True)
    with pytest.raises(AssertionError):
        attn.get_token_attnv2(0, cond, key, attn_aggregation_step_start=0)
    with pytest.raises(RuntimeError):
        attn.get_token_attnv2(0, pair, key, attn_aggregation_step_start=6)
