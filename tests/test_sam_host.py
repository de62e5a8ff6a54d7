"""Host algebra of the SAM port (lgd_amd/sam.py) vs the Hugging Face `SamModel` the reference calls
(models/sam.py:39-40), on the CPU: `lgd_amd.sam`'s kernel calls are redirected to the torch restatement of
tests/ops_emul.py, so this checks the weight re-packing, the relative-position bias folded into extra head columns,
window partition with padding, the transposed convolutions as GEMMs and the nested pixel order of the mask head — not
the HIP kernels (tests/test_sam_gpu.py does that on the MI355X)."""
import pytest
import torch

import lgd_amd  # noqa: F401
import ops_emul
import sam_cases
from lgd_amd import sam as lsam

transformers = pytest.importorskip("transformers")


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.fixture()
def emulated_ops(monkeypatch):
    """Kernel calls -> torch restatement; storage dtype fp32 instead of fp16 so that the comparison isolates the algebra
    (with fp16 storage the same run differs from the fp32 Hugging Face module by the rounding noise of ~60 chained ops,
    which the GPU test bounds separately)."""
    monkeypatch.setattr(lsam, "ops", ops_emul)
    monkeypatch.setattr(lsam, "F16", torch.float32)
    monkeypatch.setattr(ops_emul, "F16", torch.float32)


@pytest.mark.parametrize("points", [False, True])
def test_sam_host_algebra_vs_transformers(emulated_ops, points):
    cfg = sam_cases.small_config(transformers)
    hf = sam_cases.build_hf(transformers, cfg)
    inp = sam_cases.inputs(cfg, B=2, P=2, points=points)
    with torch.no_grad():
        want = hf(**inp)
        want_emb = hf.get_image_embeddings(inp["pixel_values"])
    mine = lsam.HipSamModel(lsam.SamConfig.from_hf(cfg), hf.state_dict(), device="cpu")
    got = mine(**inp, original_sizes=torch.tensor([[128, 128]] * 2), reshaped_input_sizes=torch.tensor([[128, 128]] * 2))
    assert got.pred_masks.shape == want.pred_masks.shape and got.iou_scores.shape == want.iou_scores.shape
    assert relerr(mine.get_image_embeddings(inp["pixel_values"]), want_emb) < 1e-4
    assert relerr(got.pred_masks, want.pred_masks) < 1e-3
    assert relerr(got.iou_scores, want.iou_scores) < 1e-3
    agree = ((got.pred_masks > 0) == (want.pred_masks > 0)).float().mean()
    assert agree > 0.999


def test_sam_rejects_what_the_reference_never_passes(emulated_ops):
    cfg = sam_cases.small_config(transformers)
    hf = sam_cases.build_hf(transformers, cfg)
    mine = lsam.HipSamModel(lsam.SamConfig.from_hf(cfg), hf.state_dict(), device="cpu")
    inp = sam_cases.inputs(cfg)
    with pytest.raises(NotImplementedError):
        mine(pixel_values=inp["pixel_values"])
    with pytest.raises(NotImplementedError):
        mine(**inp, input_masks=torch.zeros(1, 1, 32, 32))
    with pytest.raises(ValueError):
        mine(pixel_values=inp["pixel_values"][:, :, :64, :64], input_boxes=inp["input_boxes"])
