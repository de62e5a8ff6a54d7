"""Mask-refinement rules of the drop-in `models/sam.py` vs goldens recorded from the reference's
OWN models/sam.py (oracle/make_golden_sam.py: sam_refine_box / sam_refine_boxes / sam_refine_attn on the Hugging Face
SamModel).  On the CPU the network runs through the torch restatement of the kernels in fp32 (tests/ops_emul.py), so
the masks must match pixel for pixel; the HIP run of the same replay is in tests/test_sam_gpu.py."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lgd_amd  # noqa: E402,F401
import ops_emul  # noqa: E402
import sam_cases  # noqa: E402
import sam_refine_checks  # noqa: E402
from lgd_amd import sam as lsam  # noqa: E402

transformers = pytest.importorskip("transformers")


@pytest.fixture()
def dsam(monkeypatch):
    monkeypatch.syspath_prepend(os.path.join(ROOT, "llm-groundeddiffusion_amd", "dropin"))
    monkeypatch.setattr(lsam, "ops", ops_emul)
    monkeypatch.setattr(lsam, "F16", torch.float32)
    monkeypatch.setattr(ops_emul, "F16", torch.float32)
    from models import sam as mod
    return mod


def test_refinement_replay_matches_reference_golden(dsam):
    hf = sam_cases.build_refine_hf(transformers)
    md = dsam.wrap_sam(hf, device="cpu")
    assert isinstance(md["sam_model"], lsam.HipSamModel)
    worst = sam_refine_checks.replay(dsam, md, min_agree=0.999, conf_tol=2e-4)
    assert worst >= 0.999


def test_select_mask_rules(dsam):
    """models/sam.py:67-111 on hand-made candidates: largest admissible mask wins; low confidence and low coarse IoU
    each cost one `largest area`, so a doubly penalised large mask loses to a singly penalised smaller one."""
    m = np.zeros((3, 8, 8), dtype=bool)
    m[0, :2], m[1, :4], m[2, :] = True, True, True                    # areas 16, 32, 64
    conf = np.array([0.9, 0.9, 0.5])
    pick = lambda **k: int(dsam.select_mask(m, conf, **k)[0].sum())
    assert pick() == 32                                               # 64 is below the confidence bar: 64 - 64 = 0 < 32
    assert pick(discourage_mask_below_confidence=0.4) == 64
    ious = np.array([0.5, 0.1, 0.9])
    assert pick(coarse_ious=ious) == 16                               # 32 loses its IoU penalty: 32 - 64 < 16
    assert pick(coarse_ious=ious, discourage_mask_below_coarse_iou=0.05) == 32
    mask, c = dsam.select_mask(m, conf, coarse_ious=ious)
    assert c == 0.9 and mask.shape == (8, 8)
    with pytest.raises(ValueError):
        dsam.select_mask(m, conf, rule="smallest")


def test_coarse_mask_preprocessing(dsam):
    a = np.zeros((16, 16))
    a[4:12, 4:12] = 1.0
    a[0, 0] = 0.6                                                     # isolated speck: removed by the opening
    assert int(dsam.preprocess_mask(a, 0.5).sum()) == 65
    opened = dsam.preprocess_mask(a, 0.5, n_erode_dilate_mask=1)
    assert not opened[0, 0] and opened[5:11, 5:11].all()
    cand = np.stack([a > 0.5, np.zeros_like(a, dtype=bool)])
    iou = dsam.get_iou_with_resize(a > 0.5, cand, masks_shape=(16, 16))
    assert abs(iou[0] - 1.0) < 1e-5 and iou[1] == 0.0


def test_device_processor_vs_hugging_face_processor(dsam):
    """`DeviceSamProcessor` (torch ops, any device) against the Hugging Face `SamProcessor` the reference uses: same
    tensors up to the 8-bit rounding of the resized image, same prompt scaling, same mask post-processing."""
    hfp = transformers.SamProcessor(transformers.SamImageProcessor())
    mine = dsam.DeviceSamProcessor(device="cpu")
    images, boxes, _ = sam_cases.refine_inputs()
    odd = np.random.RandomState(0).randint(0, 256, size=(300, 480, 3), dtype=np.uint8)      # non-square: padding + crop
    for imgs, bx in (([images[0]], [[[40.0, 60.0, 300.0, 410.0]]]), ([odd], [[[10.0, 20.0, 200.0, 250.0], [5.0, 5.0, 470.0, 290.0]]])):
        want = hfp(imgs, input_boxes=bx, return_tensors="pt")
        got = mine(imgs, input_boxes=bx, return_tensors="pt")
        assert got["pixel_values"].shape == want["pixel_values"].shape
        assert float((got["pixel_values"] - want["pixel_values"]).abs().max()) < 0.02       # one 8-bit step = 0.017
        assert float((got["pixel_values"] - want["pixel_values"]).abs().mean()) < 8e-3      # PIL resizes in fixed point
        assert torch.equal(got["original_sizes"], want["original_sizes"].long())
        assert torch.equal(got["reshaped_input_sizes"], want["reshaped_input_sizes"].long())
        assert torch.allclose(got["input_boxes"], want["input_boxes"].double())
        logits = torch.randn(1, len(bx[0]), 3, 256, 256, generator=torch.Generator().manual_seed(1))
        a = hfp.image_processor.post_process_masks(logits, want["original_sizes"], want["reshaped_input_sizes"])
        b = mine.post_process_masks(logits, got["original_sizes"], got["reshaped_input_sizes"])
        assert torch.equal(a[0], b[0])
    want = hfp([images[0]], input_points=[[[100.0, 200.0]]], return_tensors="pt")
    got = mine([images[0]], input_points=[[[100.0, 200.0]]], return_tensors="pt")
    assert got["input_points"].shape == want["input_points"].shape and torch.allclose(got["input_points"], want["input_points"].double())


def test_refinement_replay_with_the_device_processor(dsam):
    hf = sam_cases.build_refine_hf(transformers)
    md = dsam.wrap_sam(hf, "device", device="cpu")
    assert isinstance(md["sam_processor"], dsam.DeviceSamProcessor)
    # the image differs from the PIL path by one 8-bit step in a quarter of the pixels: masks move at the threshold only
    worst = sam_refine_checks.replay(dsam, md, min_agree=0.95, conf_tol=3e-2)       # measured worst 0.972
    print("worst agreement with the PIL-path golden", worst)
