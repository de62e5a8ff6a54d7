"""Seeded inputs of the BoxDiff energy goldens: shared by oracle/make_golden_boxdiff.py (which runs the reference's own
compute_ca_loss_boxdiff on them) and by the tests (which regenerate them instead of loading 13 MB of random maps from
tests/golden/boxdiff_energy.npz — VERDICT r5 hygiene).  torch's CPU generator is deterministic for a given torch build; the
golden records a checksum of every case's first map so that a drift would be noticed, not silently compared."""
import torch

KEYS = [("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]    # generation/boxdiff.py:33-39


def energy_cases():
    return dict(
        hw256=dict(side=16, heads=8, bboxes=[[0.15, 0.35, 0.5, 0.8], [0.6, 0.38, 0.98, 0.8]], pos=[[1, 2, 3], [5, 6, 7]], seed=0),
        hw64=dict(side=8, heads=4, bboxes=[[0.1, 0.3, 0.45, 0.85], [0.55, 0.3, 0.95, 0.8]], pos=[[1, 2, 3], [5, 6, 7]], seed=1),
        two_boxes=dict(side=16, heads=8, bboxes=[[[0.05, 0.5, 0.3, 0.9], [0.4, 0.45, 0.7, 0.85]], [[0.72, 0.1, 0.97, 0.4]]],
                       pos=[[2, 3], [9]], seed=2),
        edge=dict(side=16, heads=8, bboxes=[[0.0, 0.0, 1.0, 0.6], [0.3, 0.7, 0.62, 0.97]], pos=[[4], [6, 7]], seed=3),
        # a 2 x 2-pixel box: (mask.sum() * P).long() = 0 -> top-k of ZERO elements, mean = NaN, and Python's
        # max(0, 1 - nan) = 0 drops the inner-box term (utils/boxdiff.py:81-83,107)
        tiny_box=dict(side=16, heads=8, bboxes=[[0.5, 0.5, 0.62, 0.62], [0.1, 0.2, 0.4, 0.9]], pos=[[2, 3], [8]], seed=4),
    )


def make_maps(side, heads, seed):
    """Five maps [1, heads, HW, 77] of probabilities over the 77 text tokens with a spatial structure (so that the
    token soft-max at x100 is not one-hot everywhere and the top-k selections are not degenerate)."""
    g = torch.Generator().manual_seed(seed)
    hw = side * side
    out = {}
    yy, xx = torch.meshgrid(torch.linspace(0, 1, side), torch.linspace(0, 1, side), indexing="ij")
    for k in KEYS:
        logits = torch.randn((1, heads, hw, 77), generator=g) * 0.3
        for tok in range(1, 12):                                        # smooth bumps per token, different per head
            cx, cy = torch.rand(heads, generator=g), torch.rand(heads, generator=g)
            bump = torch.exp(-(((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) / 0.05))
            logits[0, :, :, tok] += 2.0 * bump.reshape(heads, hw)
        out[k] = logits.softmax(dim=-1)
    return out
